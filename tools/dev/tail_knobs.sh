# the per-pair kernel at a medium batch (12.5 k config-3 pairs): trace slots per wave and their size, traceback-wave density (same box)
for e in "X=1" "BA_SLOTS_PER_WAVE=3" "BA_SLOTS_PER_WAVE=6" "BA_SLOTS_PER_WAVE=8" "BA_TB_STRIDE=2" "BA_TB_STRIDE=8" "X=2"; do echo -n "[$e] "; env $e python tools/dev/c3.py 12500 2>&1 | tail -1; done
