"""Config 3 against the number of pairs per slot: is the launch a whole number of rounds + a ragged one?
python tools/dev/rounds.py  (GPU box, development library; one workload of the largest size, sub-batches by slicing)
  n sweep at the default grid; then n = 100000 at fewer workgroups (BA_GRID) and other drain counts (BA_MQ_DRAIN)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W, synth
H.use_library(H.DEV_LIB_PATH)
NMAX = int(os.environ.get("R_NMAX", "122880"))
w = W.config3(NMAX, workers=8, size=(128, 1024))
P = w.pairs


def run(n, env=None, reps=2):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = str(v)
    try:
        b = H.BatchAligner(w.matrix, w.gaps, w.size, w.x_drop, H.TRACE | H.X_DROP | H.CIGAR_EQ, P.pool, P.q_off[:n], P.q_len[:n], P.r_off[:n], P.r_len[:n])
        b.run()
        ms = min(b.run() for _ in range(reps))
        r = b.results(); cells = int(r["cells"].sum()); bad = int((r["status"] != 0).sum())
        b.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    print(f"n={n:6d} {env or ''} kernel {ms:8.2f} ms {cells/ms/1e6:7.1f} GCUPS  ms per 15360 pairs {ms*15360/n:6.2f} bad {bad}", flush=True)


sel = os.environ.get("R_SEL", "n,grid,drain")
if "n" in sel:
    for n in (61440, 76800, 84480, 92160, 96000, 100000, 104000, 107520, 115200, 122880):
        if n <= NMAX: run(n)
if "grid" in sel:
    for g in (512, 496, 480, 476, 464, 448, 416, 384):
        run(100000, {"BA_GRID": g})
if "drain" in sel:
    for d in (0, 480, 960, 1920, 3840, 7680):
        run(100000, {"BA_MQ_DRAIN": d})
