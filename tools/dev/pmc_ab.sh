#!/bin/bash
# SQ counters of k_multi for several builds of the library (GPU box): tools/dev/pmc_ab.sh lib1.so lib2.so ...   -> stdout
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp BA_GEN_WORKERS=1
N=${PMC_N:-100000}
for lib in "$@"; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
              "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    rm -rf /tmp/pmcab
    BA_LIB=$lib timeout 900 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmcab -o p -- python3 tools/dev/c3.py $N > /tmp/pmcab.log 2>&1
    db=$(find /tmp/pmcab -name '*.db' | head -1)
    python3 - "$db" "$lib" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = next(x for x in tabs if x.startswith("counters_collection"))
ccols = [r[1] for r in c.execute(f"pragma table_info({t})")]
kcol = next(x for x in ccols if x in ("kernel_name", "name", "kernel"))
rows = c.execute(f"select {kcol}, counter_name, count(distinct dispatch_id), sum(value) from {t} group by {kcol}, counter_name").fetchall()
by = {}
for k, cn, nd, v in rows:
    if "k_multi" not in k: continue
    by.setdefault(k[:40], {})[cn] = v / max(nd, 1)
for k, m in by.items():
    print(sys.argv[2], " ".join(f"{a}={b:.5g}" for a, b in sorted(m.items())))
PY
    tail -1 /tmp/pmcab.log
  done
done
