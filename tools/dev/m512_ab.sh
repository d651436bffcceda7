#!/bin/bash
# round 6: one 512-cell slot per wave (k_multi<.., 512>) against the per-pair kernel on 32 kbp pairs at 512..4096 / 512..1024, several batch sizes, same box
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; o=$(env $2 C3_LEN=${LEN:-32000} C3_EDITS=$(( ${LEN:-32000} / 10 )) timeout 200 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; }
for n in 300 600 1200 2500 5000; do
run $n "C3_SIZE=512,4096 BA_FORCE_MULTI=1"
run $n "C3_SIZE=512,4096 BA_NO_MULTI=1"
done
run 2500 "C3_SIZE=512,4096 BA_FORCE_MULTI=1 C3_TRACE=0"
run 2500 "C3_SIZE=512,4096 BA_NO_MULTI=1 C3_TRACE=0"
run 600 "C3_SIZE=512,4096 BA_FORCE_MULTI=1 C3_TRACE=0"
run 600 "C3_SIZE=512,4096 BA_NO_MULTI=1 C3_TRACE=0"
run 2500 "C3_SIZE=512,1024 BA_FORCE_MULTI=1"
run 2500 "C3_SIZE=512,1024 BA_NO_MULTI=1"
