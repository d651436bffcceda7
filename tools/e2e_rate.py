#!/usr/bin/env python3
"""PCIe-inclusive rate of config 3: sets of pairs stream from host memory through the device and their
scores, end positions and compacted CIGAR runs come back to host arrays.
  serial:     reload -> launch -> wait -> results, nothing overlapped (one batch object)
  overlapped: two batch objects on their own streams; while one set is being aligned the host reloads the other batch with
              the next set and collects the previous set's results (ba_batch_launch / ba_batch_wait)
usage: e2e_rate.py [pairs per set] [sets]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from block_aligner_amd import hip as H, workloads as W   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sets = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = W.config3(n, workers=16)
w.size = (128, 1024)
p = w.pairs
args = (p.pool, p.q_off, p.q_len, p.r_off, p.r_len)


def collect(b):
    res = b.results()
    runs, off = b.cigars(res["cigar_len"])
    assert not res["status"].any()
    return int(res["cells"].sum()), runs.size


t0 = time.time()
a = W.make_batch(H, w)
print(f"batch creation (arena allocation included): {time.time() - t0:.2f} s, trace arena {a.info()['trace_arena_bytes'] / 1e9:.1f} GB")
a.run()
# ---- serial
t0 = time.time()
cells = 0
for _ in range(sets):
    a.reload(*args)
    a.run()
    c, nruns = collect(a)
    cells += c
dt = time.time() - t0
print(f"serial:     {sets} sets of {n} pairs in {dt:.3f} s -> {cells / dt / 1e9:.0f} GCUPS end to end ({nruns * 4 / 1e6:.0f} MB of CIGAR runs per set)")
# ---- overlapped: two batches alive at once
b = W.make_batch(H, w)
b.run()
t0 = time.time()
cells = 0
cur, nxt = a, b
cur.reload(*args)
cur.launch()
for s in range(sets):
    if s + 1 < sets:
        nxt.reload(*args)          # host packing + upload of the next set while `cur` is being aligned
    cur.wait()
    if s + 1 < sets:
        nxt.launch()
    c, _ = collect(cur)            # results of the finished set while `nxt` runs
    cells += c
    cur, nxt = nxt, cur
dt = time.time() - t0
print(f"overlapped: {sets} sets of {n} pairs in {dt:.3f} s -> {cells / dt / 1e9:.0f} GCUPS end to end")
a.close(); b.close()
