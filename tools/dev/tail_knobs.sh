# k_multi at its smallest batches: hand-offs reserved for the emptied waves' whole-wave walks (BA_TB_RESERVE; fill waves = 4096), same box
for n in 16384 25000 50000; do for v in 2048 4096 6144 8192; do echo -n "[$n reserve $v] "; BA_TB_RESERVE=$v python tools/dev/c3.py $n 2>&1 | tail -1; done; done
