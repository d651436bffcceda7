#!/usr/bin/env python3
"""One batch run for profiling: python3 tools/gpu_perf_one.py <pairs> <trace 0|1> [runs] [min_block] [max_block]
Prints kernel time, GCUPS and the total number of driver steps is not known here; use cells / (8 * block) as a guide."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_aligner_amd import hip as H, scores as S, synth
n, trace = int(sys.argv[1]), int(sys.argv[2])
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
mn = int(sys.argv[4]) if len(sys.argv) > 4 else 128
mx = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
pairs = synth.make_pairs(n, 10000, 1000, 500, synth.DNA, seed=1234)
mode = H.X_DROP | ((H.TRACE | H.CIGAR_EQ) if trace else 0)
b = H.BatchAligner(S.NucMatrix.new_simple(2, -3), (-5, -1), (mn, mx), 100, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
for _ in range(runs):
    ms = b.run()
res = b.results()
print(f"pairs={n} trace={trace} block={mn}..{mx} kernel_ms={ms:.2f} GCUPS={res['cells'].sum()/ms/1e6:.1f} cells={int(res['cells'].sum())} bad_status={int((res['status']!=0).sum())}")
