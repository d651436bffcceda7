#!/bin/bash
# round 6: k_multi at four / three / two waves per SIMD (tools/dev/w6_build.sh: w6 = three, g2 = two) over batch sizes of one to two rounds, and
# the per-pair kernel below 12 k pairs; same box, every run bounded
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 ${2:-main} $3] "; o=$(env $3 BA_LIB=$2 timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; [[ "$o" == *GCUPS* ]]; }
run 8000 libblock_aligner_hip_g2.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=6" || exit 1
for n in ${GEOM_SIZES:-6000 8000 10000 11000 12000 12500 13000 14000 16000 20000 24000 25000}; do
  run $n "" BA_FORCE_MULTI=1
  run $n libblock_aligner_hip_w6.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=6"
  [ $n -le 16000 ] && run $n libblock_aligner_hip_g2.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=6"
  [ $n -le 12000 ] && run $n "" BA_NO_MULTI=1
done
run 8000 libblock_aligner_hip_g2.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=4"
run 8000 libblock_aligner_hip_g2.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=3"
run 12000 libblock_aligner_hip_w6.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=4"
run 12000 libblock_aligner_hip_w6.so "BA_FORCE_MULTI=1 BA_TB_STRIDE=8"
