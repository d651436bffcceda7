#!/bin/bash
# round 6: the batch sizes at which k_multi's geometry rule changes over (see tools/dev/geom_rule.sh)
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 trace=${3:-1} $2] "; o=$(env $2 C3_TRACE=${3:-1} timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-110); echo "$o"; }
for n in 8400 9000 9500; do for g in 2 3; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" 1; done; done
for n in 17500 19000; do for g in 3 0; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" 1; done; done
for n in 5500 6200; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=2" 1; run $n BA_NO_MULTI=1 1; done
for n in 8400 9000; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=2" 0; run $n BA_NO_MULTI=1 0; done
for n in 10500 11000 11500; do for g in 2 3; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" 0; done; done
for n in 15000 16000 17000; do for g in 3 0; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" 0; done; done
