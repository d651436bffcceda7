# cells per call of k_walk's lanes (BA_WALK_CELLS: 8; variants 4 / 6 / 12)
# (build first: for c in 4 6 12; do tools/dev/variant.sh wc$c "-DBA_WALK_CELLS=$c -Wno-inline-asm" 0 1; done)
for c in "c4t 400000" "c2t 200000" "c5 80000"; do for lib in "" libblock_aligner_hip_wc4.so libblock_aligner_hip_wc6.so libblock_aligner_hip_wc12.so; do echo -n "[$c $lib] "; BA_LIB=$lib python tools/dev/sec.py $c 2>&1 | tail -1; done; done
