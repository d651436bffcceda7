# LOCAL_START secondary line: how much of it is the re-run of pairs whose trace region overflowed (tools/dev/local_sweep.py under other margins)
for m in 175 300 500; do
  echo "== BA_TRACE_MARGIN_PCT=$m"; BA_TRACE_MARGIN_PCT=$m python tools/dev/local_sweep.py 50000 2>&1 | grep "trace"
done
