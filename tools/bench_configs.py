#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json on one GPU (information for DESIGN.md; bench.py stays on config 3):
  C2  10 000 x 1 kbp DNA, ~90 % identity, X-drop 100, block 32..256, score only
  C4  protein pairs, BLOSUM62, gaps (-11,-1), global, block 32..256, score only and with traceback
  C5  sequence-to-profile, block 32..256, traceback
usage: bench_configs.py [pairs_scale]   (1.0 = 200k / 200k / 100k pairs; inputs resident, best of 3 launches)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_aligner_amd import hip as H, scores as S, synth   # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0


def run(name, b, n):
    ms = min(b.run() for _ in range(3))
    r = b.results()
    assert not r["status"].any()
    cells = int(r["cells"].sum())
    print(f"{name}: pairs={n} kernel={ms:.2f} ms  {cells / ms / 1e6:.1f} GCUPS  {n / ms * 1e3 / 1e6:.3f} M pairs/s", flush=True)
    b.close()


n2 = int(200000 * scale)
p = synth.make_pairs(n2, 1000, 100, 50, synth.DNA, seed=1234, workers=16)
run("C2 dna 1kbp xdrop 32..256", H.BatchAligner(S.NucMatrix.new_simple(2, -3), (-5, -1), (32, 256), 100, H.X_DROP, p.pool, p.q_off, p.q_len, p.r_off, p.r_len), n2)
run("C2 dna 1kbp xdrop+trace 32..256", H.BatchAligner(S.NucMatrix.new_simple(2, -3), (-5, -1), (32, 256), 100, H.X_DROP | H.TRACE | H.CIGAR_EQ, p.pool, p.q_off, p.q_len, p.r_off, p.r_len), n2)

n4 = int(200000 * scale)
p = synth.make_pairs(n4, (22, 900), (5, 250), 0, synth.AMINO, seed=77, workers=16)
run("C4 protein blosum62 global 32..256", H.BatchAligner(S.BLOSUM62, (-11, -1), (32, 256), 0, 0, p.pool, p.q_off, p.q_len, p.r_off, p.r_len), n4)
run("C4 protein blosum62 global+trace 32..256", H.BatchAligner(S.BLOSUM62, (-11, -1), (32, 256), 0, H.TRACE, p.pool, p.q_off, p.q_len, p.r_off, p.r_len), n4)

n5 = int(20000 * scale)
rng = np.random.default_rng(5)
AA20 = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
rows = {int(c): np.array([S.BLOSUM62.get(int(c), int(a)) for a in AA20], np.int8) for c in AA20}
t0 = time.time()
profiles, qs = [], []
for _ in range(n5):
    length = int(rng.integers(50, 501))
    cons = AA20[rng.integers(0, 20, length)]
    pr = S.AAProfile(length, 256, -1)
    pr.pos_aa[1: length + 1, AA20 - 65] = np.stack([rows[int(c)] for c in cons])
    pr.pos_gap_open_C[: length + 1] = -10; pr.pos_gap_open_R[: length + 1] = -10; pr.pos_gap_close_C[1: length + 1] = 0
    profiles.append(pr)
    qs.append(synth.mutate(rng, cons, int(0.3 * length), AA20).astype(np.uint8).tobytes())
pool = np.frombuffer(b"".join(qs) + b"\0" * 8, np.uint8)
ql = np.array([len(q) for q in qs], np.uint32)
qo = np.concatenate([[0], np.cumsum(ql[:-1])]).astype(np.uint64)
print(f"(C5 inputs built in {time.time() - t0:.1f} s)", flush=True)
run("C5 seq-to-profile trace 32..256", H.ProfileBatchAligner(profiles, (32, 256), 0, H.TRACE, pool, qo, ql), n5)
