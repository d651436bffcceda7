python -m pytest tests -m gpu -x -q > gpurun_out/t11.log 2>&1; tail -3 gpurun_out/t11.log
for n in 5000 11000 20000 40000; do
  for e in "BA_FORCE_SMALL=1" "BA_NO_SMALL=1"; do echo -n "[$e] "; env $e python tools/dev/sec.py c5 $n 2>&1 | tail -1; done
done
