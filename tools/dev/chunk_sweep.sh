# k_small: pairs a wave takes per atomic on the work counter (BA_WORK_CHUNK; default 4)
for c in "c2 200000" "c4 400000" "c2t 200000" "c5 80000"; do for v in 4 2 8 16; do echo -n "[$c chunk $v] "; BA_WORK_CHUNK=$v python tools/dev/sec.py $c 2>&1 | tail -1; done; done
