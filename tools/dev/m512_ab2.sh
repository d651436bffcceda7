#!/bin/bash
# round 6: the 512-cell slots with the library's own rule, trace slots sized for the expected stack (long pairs), and the hand-off knobs; then the per-pair-ranges set
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; o=$(env $2 C3_LEN=${LEN:-32000} C3_EDITS=$(( ${LEN:-32000} / 10 )) timeout 200 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; }
for n in 600 1200 2500; do run $n "C3_SIZE=512,4096"; done
run 2500 "C3_SIZE=512,4096 BA_NO_MULTI=1"
run 2500 "C3_SIZE=512,4096 BA_TB_STRIDE=2"
run 2500 "C3_SIZE=512,4096 BA_TB_STRIDE=4"
run 2500 "C3_SIZE=512,4096 BA_SLOTS_PER_WAVE=3"
run 2500 "C3_SIZE=512,4096 BA_SLOTS_PER_WAVE=4"
timeout 200 python tools/dev/sized_line.py 20000 2>&1 | head -12
