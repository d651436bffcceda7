for n in 6000 12500 16000 20000; do
  for c in "X=1" "BA_FORCE_MULTI=1" "BA_NO_MULTI=1" "BA_FORCE_MULTI=1 BA_WGS_PER_CU=1" "BA_FORCE_MULTI=1 BA_WORK_CHUNK=1"; do
    echo -n "[$n $c] "; env $c python tools/dev/c3.py $n 2>&1 | tail -1
  done
done
