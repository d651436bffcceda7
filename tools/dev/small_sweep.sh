#!/bin/bash
# k_small against the round-3 pipeline and the per-pair kernel over the batch size (run on the GPU box): tools/dev/small_sweep.sh
cd "$(dirname "$0")/../.."
for c in c2 c2t; do for n in 10000 20000 40000 80000; do
  a=$(BA_FORCE_SMALL=1 timeout -s KILL 200 python tools/dev/sec.py $c $n 2>&1 | tail -1 | awk '{print $5, $7}')
  b=$(BA_NO_SMALL=1 timeout -s KILL 200 python tools/dev/sec.py $c $n 2>&1 | tail -1 | awk '{print $5, $7}')
  d=$(BA_NO_SMALL=1 BA_NO_QUAD=1 timeout -s KILL 200 python tools/dev/sec.py $c $n 2>&1 | tail -1 | awk '{print $5, $7}')
  echo "$c n=$n  k_small: $a | r3 pipeline: $b | per-pair: $d"
done; done
for c in c4 c4t; do for n in 30000 70000 150000; do
  a=$(BA_FORCE_SMALL=1 timeout -s KILL 200 python tools/dev/sec.py $c $n 2>&1 | tail -1 | awk '{print $5, $7}')
  b=$(BA_NO_SMALL=1 BA_FORCE_QUAD=1 timeout -s KILL 200 python tools/dev/sec.py $c $n 2>&1 | tail -1 | awk '{print $5, $7}')
  d=$(BA_NO_SMALL=1 BA_NO_QUAD=1 timeout -s KILL 200 python tools/dev/sec.py $c $n 2>&1 | tail -1 | awk '{print $5, $7}')
  echo "$c n=$n  k_small: $a | r3 pipeline: $b | per-pair: $d"
done; done
