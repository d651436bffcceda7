"""C5 (sequence-to-PSSM) variants: python tools/dev/c5x.py [pairs] -- with / without traceback, k_small against the round-2 pipeline"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
w = W.config5(n)
for mode in (("trace",), ()):
    for env in ({}, {"BA_NO_SMALL": "1"}):
        for k, v in env.items(): os.environ[k] = v
        w.mode = mode
        b = W.make_batch(H, w)
        b.run(); b.run()
        ms = min(b.run() for _ in range(8))
        r = b.results(); cells = int(r["cells"].sum())
        print(f"c5 n={n} mode={mode} env={env} kernel={b.info()['kernel']} {ms:.3f} ms {cells/ms/1e6:.1f} GCUPS bad {int((r['status']!=0).sum())}", flush=True)
        b.close()
        for k in env: os.environ.pop(k)
