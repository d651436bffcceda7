"""What does the X-drop end game (the grows into the random tails) cost config 3? The same pairs without tails: python tools/dev/endgame.py [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for tail in (500, 0):
    w = W.config3(n, tail=tail, workers=8, size=(128, 1024))
    b = W.make_batch(H, w)
    b.run(); ms = min(b.run() for _ in range(2))
    r = b.results(); cells = int(r["cells"].sum())
    print(f"tail {tail}: kernel {ms:.2f} ms, cells/pair {cells/n:.0f}, {cells/ms/1e6:.1f} GCUPS, speculative cells/pair {b.spec_cells()/n:.0f}", flush=True)
    b.close()
