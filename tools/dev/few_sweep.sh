# pairs per resident wave up to which the per-pair kernel walks with whole waves (few_pairs, ba_host.cpp batch_plan): config 3
for n in 8000 10000 12500 16000; do
  for x in 1 2 3 4; do echo -n "[c3 $n x$x] "; BA_NO_MULTI=1 BA_FEW_PAIRS_X=$x python tools/dev/c3.py $n 2>&1 | tail -1; done
done
