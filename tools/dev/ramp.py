import sys, os
sys.path.insert(0, "/root/repo")
from block_aligner_amd import hip as H, workloads as W
for which, n in (("c2", 200000), ("c4", 400000), ("c2t", 200000)):
    w = W.config2(n, workers=8, trace=which.endswith("t")) if which.startswith("c2") else W.config4(n)
    b = W.make_batch(H, w)
    ts = [b.run() for _ in range(16)]
    print(which, " ".join(f"{t:.2f}" for t in ts), flush=True)
    b.close()
