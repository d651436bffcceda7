# k_multi compiled for two waves per SIMD (256 registers, libblock_aligner_hip_eu2.so) against four; and the new hand-off reserve
# (build first: tools/dev/variant.sh eu2 "-DMQ_WAVES_EU=2 -Wno-inline-asm" 1 8)
for r in 1 2; do
for lib in "" libblock_aligner_hip_eu2.so; do echo -n "[c3 100000 $lib] "; BA_LIB=$lib python tools/dev/c3.py 100000 2>&1 | tail -1; done
done
for lib in "" libblock_aligner_hip_eu2.so; do echo -n "[c3 25000 $lib] "; BA_LIB=$lib python tools/dev/c3.py 25000 2>&1 | tail -1; done
