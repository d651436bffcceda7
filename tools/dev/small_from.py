"""k_small against the round-2 / round-3 pipelines on pairs WITH indels (growers): 1 kbp-ish DNA, block 32..256, X-drop + traceback / score only.
python tools/dev/small_from.py [pairs ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, scores as S, synth
H.use_library(H.DEV_LIB_PATH)
NUC = S.NucMatrix.new_simple(2, -3)
for n in [int(a) for a in sys.argv[1:]] or [60000, 200000]:
    for name, kw in (("1 indel 5..60", dict(indels=1, indel_len=(5, 60))), ("3 indels 20..150", dict(indels=3, indel_len=(20, 150)))):
        pairs = synth.make_pairs(n, (200, 1500), (10, 150), 40, synth.DNA, seed=812, workers=8, **kw)
        for env in ({"BA_FORCE_SMALL": "1"}, {"BA_NO_SMALL": "1"}):
            for k, v in env.items(): os.environ[k] = v
            for mode in (H.TRACE | H.X_DROP | H.CIGAR_EQ, H.X_DROP):
                b = H.BatchAligner(NUC, (-5, -1), (32, 256), 100, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
                b.run(); b.run()
                ms = min(b.run() for _ in range(5))
                r = b.results(); cells = int(r["cells"].sum())
                print(f"n={n} {name} trace={bool(mode & H.TRACE)} {b.info()['kernel']} {ms:.3f} ms {cells / ms / 1e6:.1f} GCUPS retried {b.retried()}", flush=True)
                b.close()
            for k in env: os.environ.pop(k)
