#!/bin/bash
# round 6: score-only batches between the two- and the three-wave geometry (workgroups cut to the batch)
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 trace=${3:-1} $2] "; o=$(env $2 C3_TRACE=${3:-1} timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-110); echo "$o"; }
for n in 9200 9500 10000 10500; do for g in 2 3; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" 0; done; done
for n in 15500 16000; do for g in 3 0; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=$g" 0; done; done
for n in 7000 7600; do run $n "BA_FORCE_MULTI=1 BA_MQ_GEOM=2" 0; done
