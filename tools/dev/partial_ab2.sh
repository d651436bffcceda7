#!/bin/bash
# round 6: ... and the 128-cell slots (variant pm128): config 3 at several sizes, and the per-pair-ranges' 128-cell ranges at small n
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 len=$LEN $2] "; o=$(env $2 C3_LEN=$LEN C3_EDITS=$(( LEN / 10 )) timeout 200 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; }
LEN=10000; for n in 100000 25000 12500 8000; do run $n "BA_LIB=libblock_aligner_hip_pm128.so"; run $n "X=0"; done
LEN=9000; for n in 1200 2527; do run $n "C3_SIZE=128,1024 BA_FORCE_MULTI=1 BA_LIB=libblock_aligner_hip_pm128.so"; run $n "C3_SIZE=128,1024 BA_NO_MULTI=1"; done
