# what the traced fill of k_small pays for (walks skipped): baseline / prefetch at the top of the step / no trace stores
export BA_NO_TRACEBACK=1
for c in "c4t 400000:0" "c2t 200000:1"; do
  cfg=${c%%:*}; k=${c##*:}
  for lib in "" libblock_aligner_hip_pfe$k.so libblock_aligner_hip_nts$k.so; do
    echo -n "[$cfg $lib] "; BA_LIB=$lib python tools/dev/sec.py $cfg 2>&1 | tail -1
  done
done
unset BA_NO_TRACEBACK
echo -n "[c4 untraced] "; python tools/dev/sec.py c4 400000 | tail -1
echo -n "[c2 untraced] "; python tools/dev/sec.py c2 200000 | tail -1
