python -m pytest tests/test_gpu_multi.py tests/test_gpu_sized.py tests/test_gpu_pipelines.py -m gpu -x -q > gpurun_out/t6.log 2>&1; tail -3 gpurun_out/t6.log
echo -n "no yield: "; BA_NO_YIELD=1 python tools/dev/c3.py 100000 2>&1 | tail -1
for c in "BA_STEAL_LANES=4" "BA_STEAL_LANES=2" "BA_STEAL_LANES=8" "BA_STEAL_LANES=16" "BA_STEAL_LANES=4 BA_YIELD_WMASK=0x10" "BA_STEAL_LANES=4 BA_YIELD_WMASK=0x0" "BA_STEAL_LANES=8 BA_YIELD_WMASK=0x0"; do
  for n in 25000 100000; do
    echo -n "[$c] "; env $c BA_FORCE_MULTI=1 python tools/dev/c3.py $n 2>&1 | tail -1
  done
done
echo -n "no yield 25k: "; BA_NO_YIELD=1 python tools/dev/c3.py 25000 2>&1 | tail -1
python tools/dev/ragged_end.py 100000 2>&1 | tail -2
