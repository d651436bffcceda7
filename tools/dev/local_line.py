"""The bench's LOCAL_START / FREE_QUERY_END_GAPS lines without the oracle (kernel choice and knobs): python tools/dev/local_line.py [local|fqe] [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
which = sys.argv[1] if len(sys.argv) > 1 else "local"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
w = W.config_local(n) if which == "local" else W.config_free_end(n)
b = W.make_batch(H, w)
b.run(); b.run()
ms = min(b.run() for _ in range(6))
r = b.results(); cells = int(r["cells"].sum())
print(f"{which} n={n} {b.info()['kernel']} {ms:.3f} ms {cells/ms/1e6:.1f} GCUPS retried {b.retried()} bad {int((r['status']!=0).sum())} env {dict((k, os.environ[k]) for k in os.environ if k.startswith('BA_'))}")
