# LOCAL_START / FREE_QUERY_START_GAPS batches that start at 128 cells: k_multi's special instantiations against the per-pair kernel (config 3's pairs)
for n in ${MS_N:-16000 30000 60000}; do
  for m in ${MS_M:-local_start free_query_start_gaps}; do
    for e in "BA_FORCE_MULTI=1" "BA_NO_MULTI=1"; do
      echo -n "[c3 $n $m $e] "; env $e C3_MODE=$m python tools/dev/c3.py $n 2>&1 | tail -1
      [ -n "$MS_SCORE" ] && { echo -n "[c3 $n $m $e score-only] "; env $e C3_MODE=$m C3_TRACE=0 python tools/dev/c3.py $n 2>&1 | tail -1; }
    done
  done
done
