# the LOCAL_START secondary line (bench.py: config_local 50 k pairs) and what its parts cost
python tools/dev/local_sweep.py 50000 2>&1 | grep "trace', 'x_drop', 'local_start'"
BA_NO_TRACEBACK=1 python tools/dev/local_sweep.py 50000 2>&1 | grep "trace', 'x_drop', 'local_start'" | sed 's/^/[no walk] /'
