#!/bin/bash
# round 6: batches a little below a geometry's slot count -- fewer workgroups, so that every wave finds its four pairs (BA_GRID)?
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 trace=${3:-1} $2] "; o=$(env $2 C3_TRACE=${3:-1} timeout 100 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-110); echo "$o"; }
for t in 0 1; do
run 8000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=2" $t
run 8000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=2 BA_GRID=500" $t
run 7000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=2" $t
run 7000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=2 BA_GRID=438" $t
run 7000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=2 BA_GRID=400" $t
run 12000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=3" $t
run 12000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=3 BA_GRID=750" $t
run 11000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=3" $t
run 11000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=3 BA_GRID=688" $t
run 11000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=3 BA_GRID=640" $t
run 15000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=0" $t
run 15000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=0 BA_GRID=469" $t
run 15000 "BA_FORCE_MULTI=1 BA_MQ_GEOM=0 BA_GRID=440" $t
done
