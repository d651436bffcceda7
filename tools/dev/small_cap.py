"""C4 (score only / with traceback), pairs above a length cap removed: is the launch bound by its longest pairs' serial chains?
python tools/dev/small_cap.py 400000 [trace]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, workloads as W
H.use_library(H.DEV_LIB_PATH)
n = int(sys.argv[1]); trace = len(sys.argv) > 2
w = W.config4(n, trace=trace)
L = w.pairs.q_len.astype(np.int64) + w.pairs.r_len.astype(np.int64)
for cap in (10**9, 5000, 2500, 1200, 800):
    ps = w.pairs.subset(np.nonzero(L < cap)[0])
    b = H.BatchAligner(w.matrix, w.gaps, w.size, 0, H.TRACE if trace else 0, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len)
    ms = min(b.run() for _ in range(3)); cells = int(b.results()["cells"].sum())
    print(f"{b.info()['kernel']} trace={trace} cap {cap}: {len(ps)} pairs, kernel {ms:.3f} ms, {cells/ms/1e6:.1f} GCUPS")
    b.close()
