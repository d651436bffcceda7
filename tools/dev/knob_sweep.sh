for c in "BA_TB_STRIDE=2" "BA_TB_STRIDE=1" "BA_TB_STRIDE=3" "BA_SLOTS_PER_WAVE=10" "BA_TB_STRIDE=1 BA_SLOTS_PER_WAVE=6" "BA_MQ_DRAIN=0" "BA_NO_TRACEBACK=1" "BA_NO_SPEC=1"; do
  echo -n "[$c] "; env $c python tools/dev/c3.py 100000 2>&1 | tail -1
done
