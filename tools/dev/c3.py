"""Config 3 kernel time without the bench's checks (for development switches that skip work): python tools/dev/c3.py [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)   # the build that reads the BA_* development switches
if os.environ.get("BA_LIB"):   # another build of the library (same-box A/B)
    H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), os.environ["BA_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
trace = os.environ.get("C3_TRACE", "1") == "1"
w = W.config3(n, length=int(os.environ.get("C3_LEN", "10000")), edits=int(os.environ.get("C3_EDITS", "1000")), tail=int(os.environ.get("C3_TAIL", "500")),
              workers=int(os.environ.get("BA_GEN_WORKERS", "8")), size=tuple(int(x) for x in os.environ.get("C3_SIZE", "128,1024").split(",")), trace=trace)
if os.environ.get("C3_MODE"):   # e.g. local_start / free_query_start_gaps on top of the configuration's own modes
    w.mode = tuple(w.mode) + tuple(os.environ["C3_MODE"].split(","))
b = W.make_batch(H, w)
ms = min(b.run() for _ in range(3))
r = b.results(); cells = int(r["cells"].sum())
print(f"{os.environ.get('BA_LIB', '')} c3 n={n} trace={trace} {b.info()['kernel']} grid {b.info().get('grid')} kernel {ms:.2f} ms {cells/ms/1e6:.1f} GCUPS bad {int((r['status']!=0).sum())} cells/pair {cells/n/1e6:.2f} M retried {b.retried()} arena {b.info().get('trace_arena_bytes', 0) / 1e9:.1f} GB")
