#!/usr/bin/env python3
"""Which rectangles does config 3 fill? Histogram of Trace::blocks() (the surviving rectangles) over a few pairs: shift steps
(8 x B or B x 8) by block size B, and grow rectangles, with their share of the cells."""
import os, sys
from collections import Counter
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, scores as S, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pairs = synth.make_pairs(n, 10000, 1000, 500, synth.DNA, seed=1234)
m = S.NucMatrix.new_simple(2, -3)
cnt, cells = Counter(), Counter()
for p in range(n):
    q = bytes(pairs.pool[int(pairs.q_off[p]): int(pairs.q_off[p]) + int(pairs.q_len[p])])
    r = bytes(pairs.pool[int(pairs.r_off[p]): int(pairs.r_off[p]) + int(pairs.r_len[p])])
    b = H.Block(len(q), len(r), 1024, trace=True, x_drop=True)
    b.align(H.PaddedBytes.from_bytes(q, 1024, m), H.PaddedBytes.from_bytes(r, 1024, m), m, (-5, -1), (128, 1024), 100)
    for (_, _, w, h) in b.trace().blocks():
        key = ("shift", max(w, h)) if min(w, h) == 8 else ("grow", f"{w}x{h}")
        cnt[key] += 1; cells[key] += w * h
tot_c, tot_n = sum(cells.values()), sum(cnt.values())
for key in sorted(cnt, key=lambda k: -cells[k]):
    print(f"{key[0]:5s} {str(key[1]):>10s}: {cnt[key] / n:8.1f} rectangles/pair ({100 * cnt[key] / tot_n:5.1f} %)  {100 * cells[key] / tot_c:5.1f} % of the cells")
