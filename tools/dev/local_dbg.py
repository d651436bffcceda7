"""LOCAL_START / FREE_QUERY_START_GAPS through the per-pair kernel: ring walkers against the inline lane-0 walk (development)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
w = W.config_local(n)
for mode in (("trace", "x_drop", "free_query_start_gaps"), ("trace", "x_drop", "local_start")):
    for env in ({"BA_NO_SMALL": "1", "BA_INLINE_TRACEBACK": "1"}, {"BA_NO_SMALL": "1", "BA_FORCE_TB": "1"}, {"BA_FORCE_SMALL": "1"}):
        for k, v in env.items(): os.environ[k] = v
        w.mode = mode
        b = W.make_batch(H, w)
        b.run()
        r = b.results()
        bad = np.nonzero(r["status"])[0]
        print(mode, env, b.info()["kernel"], "bad", bad.size, bad[:8], r["status"][bad[:8]], "retried", b.retried(), flush=True)
        b.close()
        for k in env: os.environ.pop(k)
