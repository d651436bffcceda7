# whole-wave walks: config 3 at several sizes + the secondary configurations with traceback
for n in 2000 6000 12500 25000; do echo -n "[c3 $n] "; python tools/dev/c3.py $n 2>&1 | tail -1; done
echo -n "[c3 100000] "; python tools/dev/c3.py 100000 2>&1 | tail -1
for c in "c4t 400000" "c2t 200000" "c5 80000"; do echo -n "[$c] "; python tools/dev/sec.py $c 2>&1 | tail -1; done
