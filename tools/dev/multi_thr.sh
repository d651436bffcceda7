#!/bin/bash
# round 6: where k_multi takes over from the per-pair kernel with the new lane walk (same box)
cd "$(dirname "$0")/../.."
python tools/dev/multi_from.py 20000 200000 2>&1 | grep -v Warn
for L in 1000 2000; do MF_LEN=$L python tools/dev/multi_from.py 30000 2>&1 | grep "trace=True"; done
run() { echo -n "[$1 $2] "; env $2 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-120; }
for n in 6000 8000 10000 12000 12500 25000 100000; do run $n "BA_FORCE_MULTI=1"; [ $n -le 12500 ] && run $n "BA_NO_MULTI=1"; done
