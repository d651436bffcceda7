python -m pytest tests/test_gpu_multi.py tests/test_gpu_pipelines.py tests/test_gpu_independent.py -m gpu -x -q > gpurun_out/t4.log 2>&1; tail -3 gpurun_out/t4.log
for n in 12500 25000 50000 100000; do
  BA_FORCE_MULTI=1 python tools/dev/c3.py $n 2>&1 | tail -1
  BA_NO_YIELD=1 BA_FORCE_MULTI=1 python tools/dev/c3.py $n 2>&1 | tail -1
done
python tools/dev/ragged_end.py 100000 2>&1 | tail -4
python bench.py --no-e2e --no-secondary --steps 5 --warmup 2 > gpurun_out/b2.json 2> gpurun_out/b2.err; python tools/bench_line.py b2 < gpurun_out/b2.json
