"""k_multi's trace slots at 125 % of the expected stack (round 5): how many pairs of an indel-heavy batch outgrow them and are run again?
python tools/dev/margin_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
H.use_library(H.DEV_LIB_PATH)
NUC = S.NucMatrix.new_simple(2, -3)
for name, kw in (("config-3 shape", dict(length=(3000, 10000), k_edits=(300, 1000), tail=500, indels=0)),
                 ("+ 3 indels of 20..200", dict(length=(3000, 10000), k_edits=(300, 1000), tail=500, indels=3, indel_len=(20, 200))),
                 ("+ 6 indels of 100..600", dict(length=(3000, 10000), k_edits=(300, 1000), tail=500, indels=6, indel_len=(100, 600)))):
    pairs = synth.make_pairs(20000, kw["length"], kw["k_edits"], kw["tail"], synth.DNA, seed=5, indels=kw["indels"], indel_len=kw.get("indel_len", (20, 100)), workers=8)
    for env in ({}, {"BA_TRACE_MARGIN_PCT": "175", "BA_SLOTS_PER_WAVE": "8"}):
        for k, v in env.items(): os.environ[k] = v
        b = H.BatchAligner(NUC, (-5, -1), (128, 1024), 100, H.TRACE | H.X_DROP | H.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
        b.run()
        ms = min(b.run() for _ in range(3))
        r = b.results(); cells = int(r["cells"].sum())
        print(f"{name}: env {env} {b.info()['kernel']} {ms:.2f} ms {cells / ms / 1e6:.1f} GCUPS retried {b.retried()} bad {int((r['status'] != 0).sum())}", flush=True)
        b.close()
        for k in env: os.environ.pop(k)
