#!/bin/bash
# SQ counters per dispatch of a secondary configuration (run on the GPU box): tools/dev/pmc_sec.sh c5 80000
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp BA_GEN_WORKERS=1
rm -rf /tmp/pmc_$1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/pmc_$1 -o p -- python3 tools/dev/sec.py $1 $2 > gpurun_out/pmc_$1.log 2>&1
db=$(find /tmp/pmc_$1 -name '*.db' | head -1)
python3 - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
kcol = next(x for x in ccols if x in ("kernel_name", "name", "kernel"))
rows = c.execute(f"select dispatch_id, {kcol}, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name order by dispatch_id").fetchall()
by = {}
for d, k, cn, v in rows:
    by.setdefault((d, k[:40]), {})[cn] = v
for (d, k), m in by.items():
    if k.startswith("__amd"): continue
    print(d, k, " ".join(f"{a}={b:.3g}" for a, b in sorted(m.items())))
PY
