"""Launch-to-launch kernel times of the side configurations (clock / cache warm-up, spread): python tools/dev/ramp.py [c2 c4 c2t c4t ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
for which in (sys.argv[1:] or ["c2", "c4", "c2t"]):
    n = 200000 if which.startswith("c2") else 400000
    w = W.config2(n, workers=8, trace=which.endswith("t")) if which.startswith("c2") else W.config4(n, trace=which.endswith("t"))
    b = W.make_batch(H, w)
    ts = [b.run() for _ in range(16)]
    print(which, " ".join(f"{t:.2f}" for t in ts), flush=True)
    b.close()
