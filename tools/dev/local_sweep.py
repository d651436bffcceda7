"""LOCAL_START / FREE_QUERY_START_GAPS batches: k_small's special instantiations against the per-pair kernel.
python tools/dev/local_sweep.py [pairs ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
ns = [int(a) for a in sys.argv[1:]] or [50000, 100000]
for n in ns:
    w = W.config_local(n)
    for mode in (("trace", "x_drop", "local_start"), ("x_drop", "local_start"), ("trace", "x_drop", "free_query_start_gaps"), ("x_drop", "free_query_start_gaps")):
        for env in ({"BA_FORCE_SMALL": "1"}, {"BA_NO_SMALL": "1"}):
            for k, v in env.items(): os.environ[k] = v
            w.mode = mode
            b = W.make_batch(H, w)
            b.run(); b.run()
            ms = min(b.run() for _ in range(6))
            r = b.results(); cells = int(r["cells"].sum())
            print(f"local n={n} mode={mode} kernel={b.info()['kernel']} {ms:.3f} ms {cells/ms/1e6:.1f} GCUPS bad {int((r['status']!=0).sum())} retried {b.retried()}", flush=True)
            b.close()
            for k in env: os.environ.pop(k)
