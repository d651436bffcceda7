#!/bin/bash
# round 6: traceback waves per workgroup (one per BA_TB_STRIDE workgroups) with the cheaper lane walk, config 3 at several sizes, same box
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; env $2 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150; }
for n in 100000 50000 12500; do
  for s in 2 3 4 6; do run $n "BA_TB_STRIDE=$s"; done
  run $n "BA_TB_STRIDE=3 BA_TB_RESERVE=3800"
  run $n "BA_TB_STRIDE=4 BA_TB_RESERVE=3800"
  run $n "BA_TB_STRIDE=3 BA_SLOTS_PER_WAVE=10"
done
