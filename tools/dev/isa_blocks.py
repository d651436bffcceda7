#!/usr/bin/env python3
"""Per-basic-block instruction mix of one k_align instantiation, to check a change before spending GPU time on it.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBA_KIND=1 -DBA_PMAX=8 -S --cuda-device-only -o /tmp/k.s block_aligner_amd/csrc/ba_kernels.hip
  python3 tools/dev/isa_blocks.py /tmp/k.s [kernel index, default 0 = <trace, xdrop>] [min VALU per block, default 8]

Columns: label, first line, VALU, v_max_i32_dpp (6 = one 128-cell scan), v_readlane/v_writelane (cross-lane carries, parked
state and SGPR spills), scratch ops (VGPR spills), SALU, DS ops, loop depth. How to read it for PMAX = 8: the fast step
(8 unrolled columns of one 128-cell chunk) is ONE block with 48 DPP ops and ~330 VALU; runs of 8 / 4 / 2 blocks with 6 DPP
each are one column of the 1024 / 512 / 256-cell paths (a peeled first column at depth 2, the column loop at depth 3).
Also printed: the kernel's .vgpr_spill_count / .sgpr_spill_count and whether any s_swappc (a call) exists."""
import re
import sys

path = sys.argv[1]
kidx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
min_v = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
import os
KPAT = os.environ.get("KPAT", "_ZN2ba7k_align")
starts = [i for i, l in enumerate(lines) if l.startswith(KPAT) and "@" in l]
s = starts[kidx]
e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
print(lines[s].split(":")[0])
blocks, cur = [], ["entry", s + 1, 0, 0, 0, 0, 0, 0, ""]
for n in range(s + 1, e):
    t = lines[n].strip()
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", t)
    if m:
        blocks.append(cur)
        d = re.search(r"Depth=(\d)", m.group(2))
        cur = [m.group(1), n + 1, 0, 0, 0, 0, 0, 0, d.group(1) if d else "-"]
    elif t.startswith("v_"):
        cur[2] += 1
        cur[3] += "v_max_i32_dpp" in t
        cur[4] += t.startswith(("v_readlane", "v_writelane"))
    elif t.startswith("scratch_"):
        cur[5] += 1
    elif t.startswith("s_") and not t.startswith(("s_nop", "s_waitcnt")):
        cur[6] += 1
    elif t.startswith("ds_"):
        cur[7] += 1
    if "s_swappc" in t:
        print("CALL at line", n + 1)
blocks.append(cur)
print(f"{'label':12s} {'line':>6s} {'VALU':>5s} {'dpp':>4s} {'lane':>4s} {'scr':>4s} {'SALU':>5s} {'DS':>3s} depth")
for b in blocks:
    if b[2] >= min_v or b[5]:
        print(f"{b[0]:12s} {b[1] - s:6d} {b[2]:5d} {b[3]:4d} {b[4]:4d} {b[5]:4d} {b[6]:5d} {b[7]:3d} {b[8]}")
tot = [sum(b[k] for b in blocks) for k in range(2, 8)]
print("total: VALU %d, dpp %d, lane ops %d, scratch %d, SALU %d, DS %d, blocks %d" % (*tot, len(blocks)))
meta = [l.strip() for l in lines if re.match(r"\s+\.(sgpr_spill_count|vgpr_spill_count|vgpr_count):", l)]
print("kernel metadata:", " ".join(meta[3 * kidx: 3 * kidx + 3]))
