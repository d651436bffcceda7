# whole-wave walk out of LDS copies (libblock_aligner_hip_ldsw.so: the previous commit's kernels) against register windows
for r in 1 2 3; do
  for lib in "" libblock_aligner_hip_ldsw.so; do
    echo -n "[c4t $lib] "; BA_LIB=$lib python tools/dev/sec.py c4t 400000 2>&1 | tail -1
    echo -n "[c3 2000 $lib] "; BA_LIB=$lib python tools/dev/c3.py 2000 2>&1 | tail -1
  done
done
for lib in "" libblock_aligner_hip_ldsw.so; do echo -n "[c3 100000 $lib] "; BA_LIB=$lib python tools/dev/c3.py 100000 2>&1 | tail -1; echo -n "[c3 6000 $lib] "; BA_LIB=$lib python tools/dev/c3.py 6000 2>&1 | tail -1; done
