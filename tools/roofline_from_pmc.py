#!/usr/bin/env python3
"""profiles/<tag>_roofline.json from the PMC summaries written by tools/profile_round.sh: one self-describing record of what the
bench line's `roofline` rests on -- the commit and the hash of the kernel sources it was measured on, the dominant kernel, its
VALU instruction count and VALU-busy fraction, and its HBM traffic per launch.
usage: roofline_from_pmc.py <tag> <pairs> <out.json>   (reads gpurun_out/<tag>_{kernel_trace_stats,pmc_fetch,pmc_write,pmc_sq}.md)
HBM bytes per launch = 2 x FETCH_SIZE (gfx950: 128-byte requests tallied at 64 bytes, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KiB."""
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_hash import kernel_hash

tag, pairs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KPAT = r"ba::k_(?:multi|align)<8, 1, true, true"


def counter(path, name):
    for line in open(path):
        m = re.match(r"\| `(.*)` \| %s \| (\d+) \| (\d+) \| (\d+) \|" % name, line)
        if m and re.search(KPAT, m.group(1)):
            return float(m.group(4)), m.group(1)
    return None, None


def kernel_time(path):
    for line in open(path):
        m = re.match(r"\| `(.*)` \| (\d+) \| (\d+) \| (\d+) \|", line)
        if m and re.search(KPAT, m.group(1)):
            return float(m.group(4)) / 1e6, int(m.group(2))
    return None, None


g = os.path.join(ROOT, "gpurun_out")
f, kname = counter(f"{g}/{tag}_pmc_fetch.md", "FETCH_SIZE")
w, _ = counter(f"{g}/{tag}_pmc_write.md", "WRITE_SIZE")
sq = {c: counter(f"{g}/{tag}_pmc_sq.md", c)[0] for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES")}
kt_ms, kt_calls = kernel_time(f"{g}/{tag}_kernel_trace_stats.md")
sq_ms, _ = kernel_time(f"{g}/{tag}_pmc_sq.md")
busy = None
if sq["SQ_ACTIVE_INST_VALU"] and sq_ms:
    busy = sq["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sq_ms * 1e-3 * 2.4e9)   # quad-cycles -> cycles, over 1024 SIMDs at 2.4 GHz
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"]).decode().strip()
except Exception:
    commit = None   # (the GPU box holds a snapshot without .git: filled in when the record is copied to profiles/)
rec = {"tag": tag, "commit": commit, "kernel_sources_sha16": kernel_hash(), "pairs": pairs, "kind": "xdrop+trace", "kernel": kname,
       "kernel_ms_avg_kernel_trace": kt_ms, "kernel_launches_in_trace": kt_calls, "kernel_ms_under_pmc_sq": sq_ms,
       "sq": sq, "valu_busy_per_simd": busy,
       "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
       "hbm_bytes_per_launch": None if f is None or w is None else 2 * f * 1024 + w * 1024,
       "correction": "FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 bytes), WRITE_SIZE as reported; both KiB -> bytes",
       "source": "rocprofv3 --kernel-trace [--stats | --pmc ...], separate passes (tools/profile_round.sh)"}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
