#!/bin/bash
# round 6: k_multi at batch sizes below one pair per wave slot -- the block ranges of a per-pair-ranges batch (tools/dev/sized_line.py) each hold 1 .. 2.5 k pairs
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 len=$LEN $2] "; o=$(env $2 C3_LEN=$LEN C3_EDITS=$(( LEN / 10 )) timeout 200 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; }
LEN=22000; for n in 600 1262; do run $n "C3_SIZE=256,4096 BA_FORCE_MULTI=1"; run $n "C3_SIZE=256,4096 BA_NO_MULTI=1"; done
LEN=16000; for n in 600 1298 2600; do run $n "C3_SIZE=128,2048 BA_FORCE_MULTI=1"; run $n "C3_SIZE=128,2048 BA_NO_MULTI=1"; done
LEN=9000; for n in 1200 2527; do run $n "C3_SIZE=128,1024 BA_FORCE_MULTI=1"; run $n "C3_SIZE=128,1024 BA_NO_MULTI=1"; done
