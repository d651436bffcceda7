#!/bin/bash
# round 6: k_multi at three waves per SIMD (tools/dev/w6_build.sh) against the release geometry, config 3 at three sizes, same box; every run bounded
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 ${2:-main} $3] "; o=$(env $3 BA_LIB=$2 timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-170); echo "$o"; [[ "$o" == *GCUPS* ]]; }
run 12500 libblock_aligner_hip_w6.so BA_TB_STRIDE=6 || exit 1
run 12500 "" X=0
for r in 1 2; do run 100000 libblock_aligner_hip_w6.so BA_TB_STRIDE=6; run 100000 "" X=0; done
run 100000 libblock_aligner_hip_w6.so BA_TB_STRIDE=4
run 100000 libblock_aligner_hip_w6.so BA_TB_STRIDE=8
run 25000 libblock_aligner_hip_w6.so BA_TB_STRIDE=6
run 25000 "" X=0
