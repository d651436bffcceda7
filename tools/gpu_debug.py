#!/usr/bin/env python3
"""Staged GPU-vs-oracle comparison with verbose mismatch output (development aid; run under `timeout`)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from block_aligner_amd import hip as H, scores as S, synth
from oracle.oracle_py import Oracle, cigar_runs_to_string

def stage(name, pairs, m, gaps, size, xd, mode_names, eq=True):
    o = Oracle("avx2")
    mode = 0
    for x in mode_names: mode |= {"trace": H.TRACE, "x_drop": H.X_DROP}[x]
    if eq and "trace" in mode_names: mode |= H.CIGAR_EQ
    t0 = time.time()
    b = H.BatchAligner(m, gaps, size, xd, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    ms = min(b.run() for _ in range(int(os.environ.get('BA_DEBUG_RUNS', '1')))); res = b.results()
    ref = o.batch_align(m, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, gaps, size, xd, mode_names, cigar_eq=eq, threads=8)
    bad = np.nonzero((res["score"] != ref["scores"]) | (res["query_idx"] != ref["query_idx"]) | (res["reference_idx"] != ref["reference_idx"]))[0]
    st = np.nonzero(res["status"])[0]
    cig_bad = 0
    if "trace" in mode_names and not len(st):
        runs, off = b.cigars(res["cigar_len"])
        for p in range(len(pairs)):
            want = ref["cig_ops"][int(ref["cig_off"][p]): int(ref["cig_off"][p]) + int(ref["cig_len"][p])]
            got = runs[int(off[p]): int(off[p + 1])]
            if not np.array_equal(got, want):
                cig_bad += 1
                if cig_bad <= 2: print("   cigar mismatch pair", p, cigar_runs_to_string(got)[:120], "|", cigar_runs_to_string(want)[:120])
    cells_ok = int(res["cells"].sum()) == ref["cells"]
    print(f"[{name}] n={len(pairs)} size={size} mode={mode_names} kernel={ms:.2f}ms  mismatches={len(bad)} status!=0:{len(st)} cigar_bad={cig_bad} cells_ok={cells_ok} "
          f"GCUPS={res['cells'].sum()/ms/1e6:.2f} info={b.info()}", flush=True)
    for p in bad[:4]:
        print("   pair", p, "qlen", pairs.q_len[p], "rlen", pairs.r_len[p], "gpu", res["score"][p], res["query_idx"][p], res["reference_idx"][p],
              "ref", ref["scores"][p], ref["query_idx"][p], ref["reference_idx"][p], "cells", res["cells"][p])
    for p in st[:4]: print("   status pair", p, hex(res["status"][p]))
    b.close()
    return len(bad) == 0 and len(st) == 0 and cig_bad == 0

which = sys.argv[1] if len(sys.argv) > 1 else "a"
NUC = S.NucMatrix.new_simple(2, -3)
if which == "a":
    pairs = synth.make_pairs(8, 100, 5, 10, synth.DNA, seed=1)
    stage("tiny-global-16", pairs, NUC, (-5, -1), (16, 16), 0, ())
    stage("tiny-xdrop-16", pairs, NUC, (-5, -1), (16, 16), 50, ("x_drop",))
    pairs = synth.make_pairs(64, 600, 60, 30, synth.DNA, seed=7)
    stage("global-32", pairs, NUC, (-5, -1), (32, 32), 0, ())
    stage("xdrop-32", pairs, NUC, (-5, -1), (32, 32), 100, ("x_drop",))
    stage("xdrop-32-128", pairs, NUC, (-5, -1), (32, 128), 100, ("x_drop",))
    stage("global-128", pairs, NUC, (-5, -1), (128, 128), 0, ())
elif which == "b":
    pairs = synth.make_pairs(64, 600, 60, 30, synth.DNA, seed=7)
    stage("trace-global-32", pairs, NUC, (-5, -1), (32, 32), 0, ("trace",))
    stage("trace-xdrop-32-128", pairs, NUC, (-5, -1), (32, 128), 100, ("trace", "x_drop"))
elif which == "c":
    pairs = synth.make_pairs(64, 3000, 300, 100, synth.DNA, seed=9, indels=3, indel_len=(20, 200))
    stage("grow-xdrop-32-256", pairs, NUC, (-5, -1), (32, 256), 100, ("x_drop",))
    stage("grow-xdrop-128-1024", pairs, NUC, (-5, -1), (128, 1024), 100, ("x_drop",))
    stage("grow-trace-128-1024", pairs, NUC, (-5, -1), (128, 1024), 100, ("trace", "x_drop"))
    stage("grow-trace-32-2048", pairs, NUC, (-5, -1), (32, 2048), 100, ("trace",))
    pairs = synth.make_pairs(200, (22, 900), (0, 200), 0, synth.AMINO, seed=31)
    stage("aa-global", pairs, S.BLOSUM62, (-11, -1), (32, 256), 0, ())
    stage("aa-trace-xdrop", pairs, S.BLOSUM62, (-11, -1), (32, 256), 50, ("trace", "x_drop"))
elif which == "perf":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    pairs = synth.make_pairs(n, 10000, 1000, 500, synth.DNA, seed=1234)
    for mode in [("x_drop",), ("trace", "x_drop")]:
        stage("perf-10k", pairs, NUC, (-5, -1), (128, 1024), 100, mode)
