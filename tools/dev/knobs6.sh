#!/bin/bash
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; env $2 timeout 120 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-130; }
for r in 1 2; do
run 100000 "X=0"
run 100000 "BA_LIB=libblock_aligner_hip_fc2.so"
run 100000 "BA_LIB=libblock_aligner_hip_nd2.so"
run 100000 "BA_LIB=libblock_aligner_hip_fd2.so"
done
run 100000 "BA_TB_RESERVE=2000"
run 100000 "BA_TB_RESERVE=6000"
run 100000 "BA_TB_RESERVE=8000"
run 100000 "BA_SLOTS_PER_WAVE=8"
run 100000 "BA_SLOTS_PER_WAVE=10"
run 100000 "BA_MQ_DRAIN=0"
run 100000 "BA_MQ_DRAIN=2000"
