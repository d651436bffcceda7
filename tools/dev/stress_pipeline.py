"""Repeated runs of the small-block pipeline on mixed batches (GPU box): every run of every batch must equal the first one, which is
checked against the oracle. python tools/dev/stress_pipeline.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
from oracle.oracle_py import Oracle
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
o = Oracle("avx2")
os.environ["BA_FORCE_QUAD"] = "1"
cases = []
for k, (alpha, matrix, gaps, mode, xd) in enumerate([(synth.AMINO, S.BLOSUM62, (-11, -1), H.TRACE, 0), (synth.AMINO, S.BLOSUM62, (-11, -1), 0, 0),
                                                     (synth.DNA, S.NucMatrix.new_simple(2, -3), (-5, -1), H.TRACE | H.X_DROP | H.CIGAR_EQ, 60),
                                                     (synth.DNA, S.NucMatrix.new_simple(2, -3), (-5, -1), H.X_DROP, 60)]):
    ps = synth.make_pairs(6000, (0, 1200), (0, 200), 20, alpha, seed=4000 + k, indels=1, indel_len=(5, 80))
    b = H.BatchAligner(matrix, gaps, (32, 256), xd, mode, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len)
    names = tuple(n for n, f in (("trace", H.TRACE), ("x_drop", H.X_DROP)) if mode & f)
    ref = o.batch_align(matrix, ps.pool, ps.q_off, ps.q_len, ps.r_off, ps.r_len, gaps, (32, 256), xd, names, cigar_eq=bool(mode & H.CIGAR_EQ), threads=8)
    cases.append((b, ps, ref, mode))
bad = 0
for rnd in range(rounds):
    for b, _, _, _ in cases:
        b.launch()
    for b, ps, ref, mode in cases:
        b.wait()
        res = b.results()
        ok = not res["status"].any() and np.array_equal(res["score"], ref["scores"]) and int(res["cells"].sum()) == ref["cells"]
        if mode & H.TRACE:
            ok = ok and np.array_equal(res["cigar_len"], ref["cig_len"])
            runs, off = b.cigars(res["cigar_len"])
            want = np.concatenate([ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])] for p in range(len(ps))])
            ok = ok and np.array_equal(runs, want)
        if not ok:
            bad += 1
            print("MISMATCH round", rnd, "mode", mode, flush=True)
print("done:", rounds, "rounds x", len(cases), "batches in flight together, mismatches:", bad)
