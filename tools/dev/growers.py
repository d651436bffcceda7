"""Which pairs of a 32..256 workload grow (leave k_quad) and how many of the cells are theirs: python tools/dev/growers.py c4t 400000"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, workloads as W
which = sys.argv[1]; n = int(sys.argv[2])
w = {"c2t": lambda: W.config2(n, trace=True, workers=8), "c4t": lambda: W.config4(n, trace=True), "c5": lambda: W.config5(n)}[which]()
b = W.make_batch(H, w)
b.run(); r = b.results()
cells = r["cells"].astype(np.float64)
ql = w.pairs.q_len.astype(np.float64)
rl = np.array([p.str_len for p in w.profiles], np.float64) if w.profiles else w.pairs.r_len.astype(np.float64)
base = 32.0 * (ql + rl) + 32 * 32      # a pair that never grows: its first block and 8-wide strips of 32 cells
grow = cells > base * 1.05 + 2048
print(f"{which} n={n}: pairs that grow {grow.mean()*100:.1f} %, their share of the cells {cells[grow].sum()/cells.sum()*100:.1f} %, cells/base of growers {cells[grow].sum()/base[grow].sum():.2f}")
L = ql + rl
for lo, hi in ((0, 300), (300, 600), (600, 1200), (1200, 2500), (2500, 5000), (5000, 1e9)):
    m = (L >= lo) & (L < hi)
    if m.any(): print(f"  |q|+|r| in [{lo},{hi}): {m.sum()} pairs, grow {grow[m].mean()*100:.0f} %, cells share {cells[m].sum()/cells.sum()*100:.1f} %")
