# k_walk: workgroups per CU (BA_WALK_WGS_PER_CU; default 8; four waves each)
for c in "c2t 200000" "c4t 400000" "c5 80000"; do for v in 8 4 6 12 16; do echo -n "[$c wgs/cu $v] "; BA_WALK_WGS_PER_CU=$v python tools/dev/sec.py $c 2>&1 | tail -1; done; done
