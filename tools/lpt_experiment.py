#!/usr/bin/env python3
"""Does longest-first order matter? Protein pairs with the length distribution SURVEY 8(d) gives for config 4 (lognormal,
clipped to [22, 8881], mean ~ 320), batch order as generated vs sorted by decreasing length."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_aligner_amd import hip as H, scores as S, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(3)
lens = np.clip(np.exp(rng.normal(5.45, 0.75, n)), 22, 8881).astype(np.int64)
print("length mean %.0f median %.0f max %d" % (lens.mean(), np.median(lens), lens.max()))
qs, rs = [], []
for L in lens:
    a = synth.AMINO[rng.integers(0, 20, L)]
    qs.append(a)
    rs.append(synth.mutate(rng, a, int(L * rng.uniform(0.0, 0.7)), synth.AMINO).astype(np.uint8))
def build(order):
    seqs, qo, ql, ro, rl, pos = [], [], [], [], [], 0
    for k in order:
        qo.append(pos); ql.append(len(qs[k])); seqs.append(qs[k]); pos += len(qs[k])
        ro.append(pos); rl.append(len(rs[k])); seqs.append(rs[k]); pos += len(rs[k])
    return np.concatenate(seqs + [np.zeros(8, np.uint8)]), np.array(qo, np.uint64), np.array(ql, np.uint32), np.array(ro, np.uint64), np.array(rl, np.uint32)
for name, order in (("as generated", np.arange(n)), ("longest first", np.argsort(-lens, kind="stable"))):
    for mode, mname in ((0, "score"), (H.TRACE, "trace")):
        pool, qo, ql, ro, rl = build(order)
        b = H.BatchAligner(S.BLOSUM62, (-11, -1), (32, 256), 0, mode, pool, qo, ql, ro, rl)
        ms = min(b.run() for _ in range(3))
        cells = int(b.results()["cells"].sum())
        print(f"{name:14s} {mname}: kernel {ms:.2f} ms  {cells / ms / 1e6:.1f} GCUPS")
        b.close()
