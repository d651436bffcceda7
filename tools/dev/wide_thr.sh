#!/bin/bash
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; env C3_LEN=13000 C3_EDITS=1300 C3_SIZE=256,2048 $2 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-140; }
for n in 1500 2500 4000 6000; do run $n "BA_FORCE_MULTI=1"; run $n "BA_NO_MULTI=1"; done
echo -n "[c3 100k] "; python tools/dev/c3.py 100000 2>&1 | tail -1 | cut -c1-140
echo -n "[c3 100k] "; python tools/dev/c3.py 100000 2>&1 | tail -1 | cut -c1-140
