#!/bin/bash
# round 6: a wave of 256-cell slots that cannot refill its empty slot goes on stepping with the live one (variant pm256) instead of finishing the pair solo
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 len=$LEN $2] "; o=$(env $2 C3_LEN=$LEN C3_EDITS=$(( LEN / 10 )) timeout 200 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; }
LEN=22000; for n in 600 1262 2500; do run $n "C3_SIZE=256,4096 BA_FORCE_MULTI=1 BA_LIB=libblock_aligner_hip_pm256.so"; run $n "C3_SIZE=256,4096 BA_FORCE_MULTI=1"; run $n "C3_SIZE=256,4096 BA_NO_MULTI=1"; done
LEN=13000; for n in 1500 4000 30000; do run $n "C3_SIZE=256,2048 BA_FORCE_MULTI=1 BA_LIB=libblock_aligner_hip_pm256.so"; run $n "C3_SIZE=256,2048 BA_FORCE_MULTI=1"; [ $n -le 4000 ] && run $n "C3_SIZE=256,2048 BA_NO_MULTI=1"; done
