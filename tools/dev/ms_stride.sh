# traced special-mode k_multi: are the traceback waves the limit? (one per two workgroups as for the plain modes, or one per workgroup)
for m in local_start free_query_start_gaps; do
  for e in "X=1" "BA_TB_STRIDE=1" "BA_TB_STRIDE=1 BA_SLOTS_PER_WAVE=10" "BA_NO_TRACEBACK=1"; do
    echo -n "[c3 30000 $m $e] "; env $e BA_FORCE_MULTI=1 C3_MODE=$m python tools/dev/c3.py 30000 2>&1 | tail -1
  done
done
