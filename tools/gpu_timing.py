#!/usr/bin/env python3
"""Phase timers from a -DBA_TIMING build (make -C block_aligner_amd/csrc EXTRA=-DBA_TIMING OBJ=_build_t LIB=../lib/libblock_aligner_hip_timing.so DEVLIB=../lib/libblock_aligner_hip_timing_dev.so -- without DEVLIB= the development library is overwritten with a timing build).
usage: gpu_timing.py <pairs> <trace 0|1> [max_block]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_aligner_amd import hip as H, scores as S, synth
H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), os.environ.get("BA_TIMING_LIB", "libblock_aligner_hip_timing.so"))
n, trace = int(sys.argv[1]), int(sys.argv[2])
maxb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
pairs = synth.make_pairs(n, 10000, 1000, 500, synth.DNA, seed=1234)
mode = H.X_DROP | ((H.TRACE | H.CIGAR_EQ) if trace else 0)
b = H.BatchAligner(S.NucMatrix.new_simple(2, -3), (-5, -1), (128, maxb), 100, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
b.run(); ms = b.run()
prof = np.zeros(128, np.uint64)
H.lib().ba_batch_prof.argtypes = [C.c_void_p, C.c_void_p]
H.lib().ba_batch_prof(b._h, prof.ctypes.data)
names = {0: "fast prologue", 1: "fast columns", 2: "fast epilogue", 4: "generic<=128 prologue", 5: "generic<=128 columns", 6: "generic<=128 epilogue",
         8: "tall prologue", 9: "tall columns", 10: "tall epilogue", 12: "driver: rect setup", 13: "driver: place (all)", 14: "driver: post-step", 15: "pair total", 17: "waiting for a free trace slot", 18: "fast step: wait for outstanding memory ops", 32: "setup: direction / pointers", 33: "setup: prefetch consumption", 34: "setup: add_block, rz", 35: "post: border maxima (+ generic border moves)", 36: "post: offset, best cell, checkpoint", 37: "post: x-drop test"}
tot = max(float(prof[15]), 1.0); steps = float(prof[16])
print(f"pairs={n} trace={trace} kernel_ms={ms:.2f} steps/pair={steps/n:.0f} cycles/pair={tot/n:.0f} (s_memtime ticks)")
print(f"  prefetch hits per step: {float(prof[19])/max(steps,1):.3f}")
for k, v in names.items():
    print(f"  {v:28s} {float(prof[k])/n:12.0f} ticks/pair  {100*float(prof[k])/tot:5.1f}%   {float(prof[k])/max(steps,1):8.1f} ticks/step")
if prof[26]:
    w = float(prof[26])
    print(f"traceback waves={w:.0f}: iterations/wave={prof[20]/w:.0f} walking iterations/wave={prof[22]/w:.0f} polling iterations/wave={prof[23]/w:.0f}")
    print(f"  mean walking lanes per walking iteration={float(prof[21])/max(float(prof[22]),1):.1f}  ticks per walking iteration={float(prof[24])/max(float(prof[22]),1):.0f}  wave lifetime ticks={prof[25]/w:.0f}")
    print(f"  lane-iterations per walk={float(prof[21])/n:.0f}")
    wi = max(float(prof[22]), 1)
    print(f"  per walking iteration: rect entry + load issue={float(prof[27])/wi:.0f}  wait for loads={float(prof[28])/wi:.0f}  whole tb_step={float(prof[29])/wi:.0f} ticks")
if prof[43]:
    t0, tf, tw = float(prof[43]), float(prof[40]), float(prof[41])
    print(f"wall clock (100 MHz counter): last fill wave done at {(tf-t0)/1e5:.2f} ms, last traceback wave done at {(tw-t0)/1e5:.2f} ms after launch start")
    if prof[42]: print(f"  (k_multi: the last dedicated traceback wave done at {(float(prof[42])-t0)/1e5:.2f} ms, the last helper at {(tw-t0)/1e5:.2f} ms)")
if prof[55]:
    w = float(prof[55])
    print(f"k_multi fill waves={w:.0f}: solo ticks/wave={prof[50]/w:.0f} (episodes {prof[53]/w:.1f}, of which waiting for a trace slot {prof[52]/w:.0f}) step-loop ticks/wave={prof[51]/w:.0f} (steps {prof[54]/w:.0f}, {float(prof[51])/max(float(prof[54]),1):.1f} ticks/step)")
    print(f"  run() ticks/wave: new pairs {prof[56]/w:.0f}, pairs back from their slots {prof[57]/w:.0f}, pairs taken to their end {prof[58]/w:.0f}")
