#!/bin/bash
# round 6: end-of-batch knobs under the three-wave geometry (12.5 k config-3 pairs) and the two-wave one (8 k)
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; o=$(env $2 timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-110); echo "$o"; }
for r in 1 2; do run 12500 X=0; done
for e in BA_MQ_DRAIN=0 BA_MQ_DRAIN=300 BA_MQ_DRAIN=1500 BA_MQ_DRAIN=2944 BA_TB_RESERVE=1500 BA_TB_RESERVE=4500 BA_TB_RESERVE=6000 BA_SLOTS_PER_WAVE=8 BA_SLOTS_PER_WAVE=11 BA_TB_STRIDE=5 BA_TB_STRIDE=7; do run 12500 $e; done
run 8000 X=0
for e in BA_MQ_DRAIN=0 BA_MQ_DRAIN=1500 BA_TB_RESERVE=1000 BA_TB_RESERVE=3500 BA_TB_STRIDE=8 BA_TB_STRIDE=12; do run 8000 $e; done
