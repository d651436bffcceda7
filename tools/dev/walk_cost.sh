# what one whole-wave walk costs: config 3, one pair per wave, with and without the walk
for n in 1000 2000; do
  echo -n "[c3 $n walk] "; python tools/dev/c3.py $n 2>&1 | tail -1
  echo -n "[c3 $n no walk] "; BA_NO_TRACEBACK=1 python tools/dev/c3.py $n 2>&1 | tail -1
done
echo -n "[c4t] "; python tools/dev/sec.py c4t 400000 2>&1 | tail -1
echo -n "[c4t no walk] "; BA_NO_TRACEBACK=1 python tools/dev/sec.py c4t 400000 2>&1 | tail -1
echo -n "[c4t no excl] "; BA_NO_EXCL=1 python tools/dev/sec.py c4t 400000 2>&1 | tail -1
