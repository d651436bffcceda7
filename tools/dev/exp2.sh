python -m pytest tests -m gpu -x -q > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
for lib in libblock_aligner_hip_dev.so libblock_aligner_hip_s3.so libblock_aligner_hip_s3h.so; do
  for n in 12500 25000 50000 100000; do
    BA_LIB=$lib BA_FORCE_MULTI=1 python tools/dev/c3.py $n 2>&1 | tail -1
  done
done
BA_LIB=libblock_aligner_hip_dev.so python tools/dev/c3.py 12500 2>&1 | tail -1
python tools/dev/ragged_end.py 100000 2>&1 | tail -12
