python -m pytest tests/test_gpu_multi.py tests/test_gpu_small.py tests/test_gpu_kats.py -m gpu -x -q > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
for m in 0x11 0x01 0x55 0x0; do
  for n in 25000 100000; do
    echo -n "wmask $m: "; BA_YIELD_WMASK=$m BA_FORCE_MULTI=1 python tools/dev/c3.py $n 2>&1 | tail -1
  done
done
echo -n "no yield: "; BA_NO_YIELD=1 python tools/dev/c3.py 100000 2>&1 | tail -1
python tools/dev/ragged_end.py 100000 2>&1 | tail -2
python bench.py --no-e2e --steps 5 --warmup 2 > gpurun_out/b3.json 2> gpurun_out/b3.err; python tools/bench_line.py b3 < gpurun_out/b3.json
