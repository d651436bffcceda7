#!/bin/bash
# round 6: what the library's own rule (ba_host.cpp batch_build: k_multi's geometry by batch size) gives against each forced geometry and the per-pair kernel
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 trace=${3:-1} $2] "; o=$(env $2 C3_TRACE=${3:-1} timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-110); echo "$o"; }
for t in 1 0; do
for n in ${GEOM_SIZES:-5000 6200 7000 8000 8400 9000 10000 11000 12500 14000 16000 17500 20000 25000}; do
  run $n X=0 $t
  [ -n "$GEOM_ALL" ] && for g in 0 2 3; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" $t; done
  [ -n "$GEOM_ALL" ] && run $n BA_NO_MULTI=1 $t
done; done
