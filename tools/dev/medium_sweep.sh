#!/bin/bash
# round 6: one- and two-round batches of config 3 (12 500 / 25 000 pairs: what 8 / 4 GPUs get of the north-star's 100 000) against the launch knobs, same box
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; env $2 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150; }
for n in 12500 25000; do
  run $n "X=0"
  run $n "BA_WORK_CHUNK=4"
  run $n "BA_WORK_CHUNK=4 BA_MQ_DRAIN=0"
  run $n "BA_MQ_DRAIN=0"
  run $n "BA_MQ_DRAIN=3840"
  run $n "BA_GRID=448"
  run $n "BA_GRID=416 BA_WORK_CHUNK=4"
  run $n "BA_GRID=384 BA_WORK_CHUNK=4"
  run $n "BA_TB_RESERVE=1000"
  run $n "BA_TB_RESERVE=3800"
  run $n "BA_TB_RESERVE=6000"
  run $n "BA_NO_DONATE=1"
  run $n "BA_TB_STRIDE=1"
  run $n "BA_TB_STRIDE=3"
  run $n "BA_NO_MULTI=1"
done
