# config 3 with other tail lengths (the unrelated bases behind the related part: what the X-drop search runs into at the end of every pair)
for t in 0 200 500 800 1200; do echo -n "[tail $t] "; C3_TAIL=$t python tools/dev/c3.py 30000 2>&1 | tail -1; done
echo -n "[tail 800 no walk] "; BA_NO_TRACEBACK=1 C3_TAIL=800 python tools/dev/c3.py 30000 2>&1 | tail -1
echo -n "[tail 800 score-only] "; C3_TRACE=0 C3_TAIL=800 python tools/dev/c3.py 30000 2>&1 | tail -1
echo -n "[tail 800 no spec] "; BA_NO_SPEC=1 C3_TAIL=800 python tools/dev/c3.py 30000 2>&1 | tail -1
