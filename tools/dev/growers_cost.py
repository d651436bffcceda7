"""C4 with and without the pairs that grow, score only and with traceback: what do the solo episodes cost k_small? python tools/dev/growers_cost.py [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
os.environ["BA_FORCE_SMALL"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
w = W.config4(n, trace=False)
b = W.make_batch(H, w); b.run(); cells = b.results()["cells"].astype(np.float64); b.close()
L = w.pairs.q_len.astype(np.float64) + w.pairs.r_len.astype(np.float64)
grow = cells > (32.0 * L + 32 * 32) * 1.05 + 2048
print(f"pairs that grow: {grow.mean()*100:.1f} %, their cells {cells[grow].sum()/cells.sum()*100:.1f} %")
for name, idx in (("all", np.arange(n)), ("without growers", np.nonzero(~grow)[0]), ("growers only", np.nonzero(grow)[0])):
    ps = w.pairs.subset(idx)
    for trace in (0, 1):
        bb = H.BatchAligner(w.matrix, w.gaps, w.size, 0, H.TRACE if trace else 0, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len)
        ms = min(bb.run() for _ in range(3)); c = int(bb.results()["cells"].sum())
        print(f"  {name:16s} trace={trace} {bb.info()['kernel']}: {len(idx)} pairs, {ms:.3f} ms, {c/ms/1e6:.1f} GCUPS")
        bb.close()
