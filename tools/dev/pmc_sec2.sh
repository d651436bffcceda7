#!/bin/bash
# SQ counters per kernel of a secondary configuration, two passes (run on the GPU box): tools/dev/pmc_sec2.sh c4t 400000
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp BA_GEN_WORKERS=1
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc2_$1
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc2_$1 -o p -- python3 tools/dev/sec.py $1 $2 > gpurun_out/pmc2_$1.log 2>&1
  db=$(find /tmp/pmc2_$1 -name '*.db' | head -1)
  python3 - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
kcol = next(x for x in ccols if x in ("kernel_name", "name", "kernel"))
rows = c.execute(f"select {kcol}, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by {kcol}, counter_name").fetchall()
by = {}
for k, cn, nd, v in rows:
    if k.startswith("__amd") or "pack_seq" in k: continue
    by.setdefault(k[:60], {})[cn] = v / max(nd, 1)
for k, m in by.items():
    print(k, " ".join(f"{a}={b:.4g}" for a, b in sorted(m.items())))
PY
done
tail -1 gpurun_out/pmc2_$1.log
