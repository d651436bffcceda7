#!/bin/bash
# round 6: batches that start at 256 cells (13 kbp reads, block 256..2048: percent_len 1 % .. 10 %), k_multi's 256-cell slots against the per-pair kernel, same box
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; env C3_LEN=13000 C3_EDITS=1300 C3_SIZE=256,2048 $2 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-140; }
for n in 70000 20000 8000; do run $n "BA_FORCE_MULTI=1"; run $n "BA_NO_MULTI=1"; run $n "X=1"; done
run 70000 "C3_TRACE=0 BA_FORCE_MULTI=1"; run 70000 "C3_TRACE=0 BA_NO_MULTI=1"
