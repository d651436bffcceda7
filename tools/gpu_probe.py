import sys, os, time
t0 = time.time()
def log(*a): print(f"[{time.time()-t0:7.2f}s]", *a, flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
log("start")
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
log("imports done")
L = H.lib(); log("lib loaded")
log("device_count", H.device_count())
pairs = synth.make_pairs(2, 40, 2, 5, synth.DNA, seed=1)
m = S.NucMatrix.new_simple(2, -3)
b = H.BatchAligner(m, (-5, -1), (16, 16), 0, 0, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
log("batch created", b.info())
ms = b.run(); log("run done ms", ms)
res = b.results(); log("results", res)
