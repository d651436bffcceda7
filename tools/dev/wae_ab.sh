# whole-wave walks in k_walk counted from the first non-exclusive pair (BA_NO_WALK_AFTER_EXCL=1: as before)
for r in 1 2; do for c in "c4t 400000" "c4t 150000" "c5 80000"; do for e in "X=1" "BA_NO_WALK_AFTER_EXCL=1"; do echo -n "[$c $e] "; env $e python tools/dev/sec.py $c 2>&1 | tail -1; done; done; done
