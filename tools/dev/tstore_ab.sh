# k_small's trace stores: four 8-byte stores per step against two 16-byte ones (-DSM_TSTORE4 variants)
for r in 1 2; do
for c in "c4t 400000:0" "c2t 200000:1"; do
  cfg=${c%%:*}; k=${c##*:}
  for lib in "" libblock_aligner_hip_ts4$k.so; do
    echo -n "[$cfg $lib] "; BA_LIB=$lib python tools/dev/sec.py $cfg 2>&1 | tail -1
  done
done
done
