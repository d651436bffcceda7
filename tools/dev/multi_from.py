"""From how many pairs k_multi? Short pairs that start at 128 cells (300..1500 bases, block 128..512, X-drop + traceback), k_multi forced against the
per-pair kernel: python tools/dev/multi_from.py [pairs ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, scores as S, synth
H.use_library(H.DEV_LIB_PATH)
if os.environ.get("BA_LIB"):   # another build of the library (same-box A/B)
    H.use_library(os.path.join(os.path.dirname(H.DEV_LIB_PATH), os.environ["BA_LIB"]))
NUC = S.NucMatrix.new_simple(2, -3)
for n in [int(a) for a in sys.argv[1:]] or [10000, 13000, 16384]:
    L = int(os.environ.get("MF_LEN", "0"))
    pairs = synth.make_pairs(n, (L, L + 1), (L // 12, L // 10), 100, synth.DNA, seed=2025, workers=8) if L else synth.make_pairs(n, (300, 1500), (20, 150), 80, synth.DNA, seed=2025, indels=1, indel_len=(10, 120), workers=8)
    for env in ({"BA_FORCE_MULTI": "1"}, {"BA_NO_MULTI": "1"}):
        for k, v in env.items(): os.environ[k] = v
        for mode in (H.TRACE | H.X_DROP | H.CIGAR_EQ, H.X_DROP):
            b = H.BatchAligner(NUC, (-5, -1), (128, 512), 80, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
            b.run(); b.run()
            ms = min(b.run() for _ in range(5))
            r = b.results(); cells = int(r["cells"].sum())
            print(f"n={n} trace={bool(mode & H.TRACE)} {b.info()['kernel']} {ms:.3f} ms {cells / ms / 1e6:.1f} GCUPS retried {b.retried()} env {dict((k, os.environ[k]) for k in os.environ if k.startswith('BA_'))}", flush=True)
            b.close()
        for k in env: os.environ.pop(k)
