python -m pytest tests/test_gpu_small.py -m gpu -x -q -k "profile" > gpurun_out/t8.log 2>&1; tail -15 gpurun_out/t8.log
python tools/dev/sec.py c5 80000 2>&1 | tail -2
BA_NO_SMALL=1 python tools/dev/sec.py c5 80000 2>&1 | tail -2
python tools/dev/sec.py c5 20000 2>&1 | tail -2
