"""Phase timers (-DBA_TIMING build of the class-4 DNA kernels: tools/dev/variant.sh timing "-DBA_TIMING" 1 4) of the short-pair indel set of
tools/dev/multi_from.py through k_multi and through the per-pair kernel: python tools/dev/mf_timing.py [pairs]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, scores as S, synth
H.use_library(os.path.join(os.path.dirname(H.DEV_LIB_PATH), os.environ.get("BA_TIMING_LIB", "libblock_aligner_hip_timing.so")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
NUC = S.NucMatrix.new_simple(2, -3)
pairs = synth.make_pairs(n, (300, 1500), (20, 150), 80, synth.DNA, seed=2025, indels=1, indel_len=(10, 120), workers=8)
names = {4: "generic<=128 prologue", 5: "generic<=128 columns", 6: "generic<=128 epilogue", 8: "tall prologue", 9: "tall columns", 10: "tall epilogue",
         0: "fast prologue", 1: "fast columns", 2: "fast epilogue", 12: "driver: rect setup", 13: "driver: place (all)", 14: "driver: post-step", 15: "pair total (run)",
         45: "run() total", 47: "run(): before the loop", 32: "setup: direction / pointers", 33: "setup: prefetch consumption", 34: "setup: add_block, rz",
         35: "post: border maxima", 36: "post: offset, best cell, checkpoint", 37: "post: x-drop test"}
for env in ({"BA_FORCE_MULTI": "1"}, {"BA_NO_MULTI": "1"}):
    for k, v in env.items(): os.environ[k] = v
    b = H.BatchAligner(NUC, (-5, -1), (128, 512), 80, H.TRACE | H.X_DROP | H.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
    b.run(); ms = b.run()
    prof = np.zeros(128, np.uint64)
    H.lib().ba_batch_prof.argtypes = [C.c_void_p, C.c_void_p]
    H.lib().ba_batch_prof(b._h, prof.ctypes.data)
    r = b.results(); cells = int(r["cells"].sum())
    print(f"== {b.info()['kernel']} n={n} {ms:.2f} ms {cells/ms/1e6:.0f} GCUPS retried {b.retried()} steps/pair={float(prof[16])/n:.1f}")
    for k, nm in names.items():
        if prof[k]: print(f"   {nm:36s} {float(prof[k])/n:10.0f} ticks/pair")
    if prof[55]:
        w = float(prof[55])
        print(f"   k_multi fill waves={w:.0f}: solo ticks/wave={prof[50]/w:.0f} (episodes {prof[53]/w:.1f}, waiting for a trace slot {prof[52]/w:.0f}) step-loop ticks/wave={prof[51]/w:.0f} (steps {prof[54]/w:.0f})")
        print(f"   run() ticks/wave: new pairs {prof[56]/w:.0f}, back from slots {prof[57]/w:.0f}, to their end {prof[58]/w:.0f}")
    b.close()
    for k in env: os.environ.pop(k)
