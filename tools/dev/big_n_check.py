#!/usr/bin/env python3
"""Sanity at large n (GPU box): 2 M tiny protein pairs with traceback through one batch (grid sizing, ticket ring, device
packing, longest-first order), with a spot check of 30 pairs against the per-pair API."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, scores as S, synth
n = 2000000
t0 = time.time()
p = synth.make_pairs(n, (22, 60), (0, 10), 0, synth.AMINO, seed=5, workers=16)
print("gen", time.time() - t0)
t0 = time.time()
b = H.BatchAligner(S.BLOSUM62, (-11, -1), (32, 64), 0, H.TRACE, p.pool, p.q_off, p.q_len, p.r_off, p.r_len)
print("create", time.time() - t0)
ms = b.run(); ms = b.run()
r = b.results()
print("kernel ms", ms, "pairs/s", n / ms * 1e3, "bad", int((r["status"] != 0).sum()))
runs, off = b.cigars(r["cigar_len"])
print("runs", runs.size, "first", H.runs_to_string(runs[off[0]:off[1]]))
# spot check vs per-pair API on 50 pairs
from block_aligner_amd.hip import Block, PaddedBytes
bad = 0
for k in np.random.default_rng(1).integers(0, n, 30):
    q = bytes(p.pool[int(p.q_off[k]): int(p.q_off[k]) + int(p.q_len[k])]); rr = bytes(p.pool[int(p.r_off[k]): int(p.r_off[k]) + int(p.r_len[k])])
    blk = Block(len(q), len(rr), 64, trace=True)
    blk.align(PaddedBytes.from_bytes(q, 64, S.BLOSUM62), PaddedBytes.from_bytes(rr, 64, S.BLOSUM62), S.BLOSUM62, (-11, -1), (32, 64), 0)
    res = blk.res()
    if res.score != int(r["score"][k]): bad += 1
print("spot-check mismatches", bad)
