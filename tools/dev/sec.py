"""Secondary configs quickly: python tools/dev/sec.py [c2|c2t|c4|c4t|c5] [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
WK = int(os.environ.get("BA_GEN_WORKERS", "8"))
H.use_library(H.DEV_LIB_PATH)   # the build that reads the BA_* development switches
if os.environ.get("BA_LIB"):   # another build of the library (same-box A/B)
    H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), os.environ["BA_LIB"])
which = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
w = {"c2": lambda: W.config2(n, workers=WK), "c2t": lambda: W.config2(n, trace=True, workers=WK), "c4": lambda: W.config4(n), "c4t": lambda: W.config4(n, trace=True), "c5": lambda: W.config5(n)}[which]()
b = W.make_batch(H, w)
b.run(); b.run()
ms = min(b.run() for _ in range(8))   # (two untimed launches first: clocks and caches settle, see bench.py secondary_line)
r = b.results(); cells = int(r["cells"].sum())
print(f"{os.environ.get('BA_LIB','')} {which} n={n} env={os.environ.get('BA_INLINE_TRACEBACK','')}{os.environ.get('BA_TB_STRIDE','')} kernel {ms:.3f} ms {cells/ms/1e6:.1f} GCUPS  cells/pair {cells/n:.0f} bad {int((r['status']!=0).sum())}")
