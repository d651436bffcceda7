#!/bin/bash
# round 6: the geometries of tools/dev/geom_sweep.sh on score-only X-drop batches (C3_TRACE=0)
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 ${2:-main} $3] "; o=$(env $3 C3_TRACE=0 BA_LIB=$2 timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; [[ "$o" == *GCUPS* ]]; }
run 8000 libblock_aligner_hip_g2.so "BA_FORCE_MULTI=1" || exit 1
for n in 4000 6000 8000 10000 12000 14000 16000 20000; do
  run $n "" BA_FORCE_MULTI=1
  run $n libblock_aligner_hip_w6.so "BA_FORCE_MULTI=1"
  [ $n -le 12000 ] && run $n libblock_aligner_hip_g2.so "BA_FORCE_MULTI=1"
  [ $n -le 12000 ] && run $n "" BA_NO_MULTI=1
done
