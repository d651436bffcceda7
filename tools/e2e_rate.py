#!/usr/bin/env python3
"""PCIe-inclusive rate of config 3: per round, reload the pairs from host memory (raw bytes up, images
built on the device), one launch, scores / end positions / compacted CIGARs back to host arrays. Nothing is overlapped.
usage: e2e_rate.py [pairs]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_aligner_amd import hip as H, scores as S, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pairs = synth.make_pairs(n, 10000, 1000, 500, synth.DNA, seed=1234)
mode = H.X_DROP | H.TRACE | H.CIGAR_EQ
t0 = time.time()
b = H.BatchAligner(S.NucMatrix.new_simple(2, -3), (-5, -1), (128, 1024), 100, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
print(f"batch creation (arena allocation included): {time.time() - t0:.2f} s")
b.run()
for rnd in range(3):
    t0 = time.time()
    b.reload(pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    t1 = time.time()
    ms = b.run()
    t2 = time.time()
    res = b.results()
    runs, off = b.cigars(res["cigar_len"])
    t3 = time.time()
    cells = int(res["cells"].sum())
    print(f"round {rnd}: reload {t1 - t0:.3f} s, launch {t2 - t1:.3f} s (kernel {ms:.1f} ms), results + {runs.size * 4 / 1e6:.0f} MB of CIGAR runs {t3 - t2:.3f} s"
          f" -> {cells / (t3 - t0) / 1e9:.0f} GCUPS end to end, {cells / (ms / 1e3) / 1e9:.0f} GCUPS kernel")
