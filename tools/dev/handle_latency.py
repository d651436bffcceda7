"""Latency of the per-pair handle API (block_align_* + block_cigar_*): python tools/dev/handle_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
rng = np.random.default_rng(2)
for name, L in (("20 residues", 20), ("300 residues", 300), ("900 residues", 900), ("3000 residues", 3000)):
    r = synth.rand_str(rng, L, synth.AMINO); q = synth.mutate(rng, r, L // 10, synth.AMINO)
    pq = H.PaddedBytes.from_bytes(q.tobytes(), 256, S.AAMatrix); pr = H.PaddedBytes.from_bytes(r.tobytes(), 256, S.AAMatrix)
    a = H.Block(len(q), len(r), 256, trace=True)
    cg = H.Cigar(len(q), len(r))
    ta, tc = [], []
    for _ in range(30):
        t0 = time.perf_counter(); a.align(pq, pr, S.BLOSUM62, S.Gaps(-11, -1), (32, 256), 0); t1 = time.perf_counter()
        res = a.res(); a.trace().cigar(res.query_idx, res.reference_idx, cg); t2 = time.perf_counter()
        ta.append(t1 - t0); tc.append(t2 - t1)
    print(f"{name}: align {min(ta)*1e6:.0f} us, cigar {min(tc)*1e6:.0f} us (best of 30)")
