#!/bin/bash
# round 6: what the row-tiled class (blocks up to 4096 cells: borders in global memory) costs pairs whose block is mostly small -- the same 32 kbp pairs at 512..4096 and at 512..2048
cd "$(dirname "$0")/../.."
run() { echo -n "[$1 $2] "; o=$(env $2 C3_LEN=32000 C3_EDITS=3200 timeout 200 python tools/dev/c3.py $1 2>&1 | tail -1 | cut -c1-150); echo "$o"; }
for n in 2500 600; do
run $n C3_SIZE=512,4096
run $n C3_SIZE=512,2048
run $n "C3_SIZE=512,4096 C3_TRACE=0"
run $n "C3_SIZE=512,2048 C3_TRACE=0"
done
