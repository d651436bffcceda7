"""Per-pair phase times of a secondary configuration from a -DBA_TIMING build (libblock_aligner_hip_timing.so):
python tools/dev/sec_timing.py [c2|c2t|c4|c4t|c5] [pairs]. (The per-pair launches of a small-block batch add into one set of timers.)"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), "libblock_aligner_hip_timing.so")
which = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
w = {"c2": lambda: W.config2(n), "c2t": lambda: W.config2(n, trace=True), "c4": lambda: W.config4(n), "c4t": lambda: W.config4(n, trace=True), "c5": lambda: W.config5(n)}[which]()
b = W.make_batch(H, w)
b.run(); ms = b.run()
prof = np.zeros(128, np.uint64)
H.lib().ba_batch_prof.argtypes = [C.c_void_p, C.c_void_p]
H.lib().ba_batch_prof(b._h, prof.ctypes.data)
names = {44: "pair taken -> run()", 45: "run()", 47: "run(): before the step loop", 15: "run(): step loop", 12: "  rect setup", 13: "  place", 14: "  post-step", 17: "slot wait"}
print(f"{which} n={n} kernel {ms:.3f} ms (timing build); s_memtime ticks = 10 ns")
for ps, base in (("per-pair kernel", 0),):
    pairs = float(prof[base + 46])
    if not pairs:
        continue
    print(f"{ps}: {pairs:.0f} pairs, {float(prof[base + 16]) / pairs:.1f} driver steps per pair")
    for k, v in names.items():
        print(f"  {v:30s} {float(prof[base + k]) / pairs / 100:9.2f} us/pair")
