"""A mixed-length read set, every pair with its own block range (percent_len 1 % .. 10 % of its length, examples/nanopore_bench_global.rs:144-171):
python tools/dev/sized_line.py [pairs] -- what a bench line for ba_sized_batch_* would say."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
H.use_library(H.DEV_LIB_PATH)   # the build that reads the BA_* development switches
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
rng = np.random.default_rng(7)
lens = np.exp(rng.uniform(np.log(1000), np.log(40000), n)).astype(int)
lists = []
for L in lens:
    r = synth.rand_str(rng, int(L), synth.DNA)
    q = synth.mutate(rng, r, int(L) // 10, synth.DNA)
    t = int(rng.integers(0, 500))
    lists.append((np.concatenate([q, synth.rand_str(rng, t, synth.DNA)]).astype(np.uint8).tobytes(), np.concatenate([r, synth.rand_str(rng, t, synth.DNA)]).astype(np.uint8).tobytes()))
pairs = synth.PairSet.from_lists(lists)
for mode in (H.TRACE | H.X_DROP | H.CIGAR_EQ, H.X_DROP):
    b = H.SizedBatchAligner(S.NucMatrix.new_simple(2, -3), (-5, -1), 100, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, percent=(0.01, 0.1))
    b.run()
    ms = min(b.run() for _ in range(3))
    res = b.results(); cells = int(res["cells"].sum())
    print(f"sized n={n} trace={bool(mode & H.TRACE)} {ms:.2f} ms {cells / ms / 1e6:.1f} GCUPS bad {int((res['status'] != 0).sum())}")
    for c in b.classes(): print("   ", c)
    b.close()
