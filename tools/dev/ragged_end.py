"""When do config 3's fill waves run out of pairs? (GPU box, development library): python tools/dev/ragged_end.py [pairs]
k_multi's fill waves count themselves into 4 ms buckets after the launch's start when they leave the fill (ba_multi.hpp, -DBA_ENDHIST)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from block_aligner_amd import hip as H, workloads as W
H.use_library(os.path.join(os.path.dirname(H.LIB_PATH), "libblock_aligner_hip_endhist.so"))   # tools/dev/variant.sh endhist -DBA_ENDHIST
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = W.config3(n, workers=8, size=(128, 1024))
b = W.make_batch(H, w)
b.run(); ms = b.run()
prof = np.zeros(128, np.uint64)
H.lib().ba_batch_prof.argtypes = [C.c_void_p, C.c_void_p]
H.lib().ba_batch_prof(b._h, prof.ctypes.data)
h = prof[64:128].astype(np.int64)
tot = int(h.sum()); last = int(np.nonzero(h)[0].max())
print(f"pairs={n} kernel {ms:.2f} ms; fill waves {tot}; last fill wave leaves in bucket {last} ({4*last}..{4*last+4} ms)")
acc = 0; idle = 0.0
for k in range(last + 1):
    if h[k]:
        print(f"  {4*k:4d}..{4*k+4:4d} ms: {h[k]:5d} waves")
    idle += h[k] * (4 * (last - k))
print(f"idle wave-time before the last fill wave leaves: {idle / (tot * 4.0 * (last + 1)) * 100:.1f} % of fill waves x launch")
t0 = float(prof[43])
print(f"last fill wave leaves at {(float(prof[40])-t0)/1e5:.2f} ms, last dedicated traceback wave at {(float(prof[42])-t0)/1e5:.2f} ms, last helper walk done at {(float(prof[41])-t0)/1e5:.2f} ms")
