"""Time of ONE traceback through the handle API (k_traceback: a whole wave on one path): python tools/dev/walk_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
H.use_library(H.DEV_LIB_PATH)
if os.environ.get("BA_LIB"):
    H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), os.environ["BA_LIB"])
rng = np.random.default_rng(1)
for name, alpha, cls, matrix, gaps, L, edits, size, eq in (("protein 8881", synth.AMINO, S.AAMatrix, S.BLOSUM62, (-11, -1), 8881, 2600, (32, 256), False),
                                                          ("dna 10k", synth.DNA, S.NucMatrix, S.NucMatrix.new_simple(2, -3), (-5, -1), 10000, 1000, (128, 1024), True),
                                                          ("dna 1k", synth.DNA, S.NucMatrix, S.NucMatrix.new_simple(2, -3), (-5, -1), 1000, 100, (32, 256), True)):
    r = synth.rand_str(rng, L, alpha); q = synth.mutate(rng, r, edits, alpha)
    pq = H.PaddedBytes.from_bytes(q.tobytes(), size[1], cls); pr = H.PaddedBytes.from_bytes(r.tobytes(), size[1], cls)
    a = H.Block(len(q), len(r), size[1], trace=True)
    a.align(pq, pr, matrix, S.Gaps(*gaps), size, 0)
    res = a.res()
    cg = H.Cigar(res.query_idx, res.reference_idx)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        if eq: a.trace().cigar_eq(pq, pr, res.query_idx, res.reference_idx, cg)
        else: a.trace().cigar(res.query_idx, res.reference_idx, cg)
        ts.append(time.perf_counter() - t0)
    n = res.query_idx + res.reference_idx
    print(f"{os.environ.get('BA_LIB','')} {name}: cigar call {min(ts)*1e3:.3f} ms, path <= {n} cells: {min(ts)*1e6/n*1e3:.1f} ns per cell (incl. launch + copy), runs {cg.len()}")
