export BA_NO_YIELD=1
for lib in libblock_aligner_hip_old.so libblock_aligner_hip_lay1.so libblock_aligner_hip_dev.so; do
  echo -n "$lib: "; BA_LIB=$lib python tools/dev/c3.py 100000 2>&1 | tail -1
done
PMC_N=50000 tools/dev/pmc_ab.sh libblock_aligner_hip_old.so libblock_aligner_hip_lay1.so libblock_aligner_hip_dev.so 2>&1 | grep "^lib"
unset BA_NO_YIELD
for c in "BA_WORK_CHUNK=4 BA_MQ_DRAIN=0" "BA_WORK_CHUNK=4" "BA_MQ_DRAIN=0" "BA_WORK_CHUNK=1"; do
  for n in 8000 12500 25000; do
    echo -n "[$c] "; env $c BA_NO_YIELD=1 BA_FORCE_MULTI=1 BA_LIB=libblock_aligner_hip_lay1.so python tools/dev/c3.py $n 2>&1 | tail -1
  done
done
for n in 8000 12500; do echo -n "[per-pair kernel] "; BA_NO_MULTI=1 python tools/dev/c3.py $n 2>&1 | tail -1; done
