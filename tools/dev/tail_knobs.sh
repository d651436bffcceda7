# config 3: trace slots per wave against their size (margin over the expected stack), same box
for e in "X=1" "BA_TRACE_MARGIN_PCT=150 BA_SLOTS_PER_WAVE=9" "BA_TRACE_MARGIN_PCT=140 BA_SLOTS_PER_WAVE=10" "BA_TRACE_MARGIN_PCT=125 BA_SLOTS_PER_WAVE=10" "BA_TRACE_MARGIN_PCT=125 BA_SLOTS_PER_WAVE=9" "X=2"; do echo -n "[$e] "; env $e python tools/dev/c3.py 100000 2>&1 | tail -1; done
