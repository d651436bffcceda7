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


import numpy as np   # noqa: E402
host_bufs = {}   # per batch: a page-locked output buffer (ba_host_alloc) that the gather kernel writes directly


tc = [0.0, 0.0]


def collect(b):
    t_a = time.time()
    res = b.results()
    t_b = time.time()
    runs, off = b.cigars(res["cigar_len"], out=host_bufs.get(id(b)))
    tc[0] += t_b - t_a; tc[1] += time.time() - t_b
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
for x in (a, b):
    host_bufs[id(x)] = H.pinned_array(200_000_000 * max(1, n // 100000))
t0 = time.time()
cells = 0
cur, nxt = a, b
cur.reload(*args)
cur.launch()
cur.compact_cigars(host_bufs[id(cur)])
ph = {"reload": 0.0, "wait": 0.0, "launch": 0.0, "collect": 0.0}
for s in range(sets):
    t1 = time.time()
    if s + 1 < sets:
        nxt.reload(*args)          # host packing + upload of the next set while `cur` is being aligned
    t2 = time.time()
    cur.wait()
    t3 = time.time()
    if s + 1 < sets:
        nxt.launch()
        nxt.compact_cigars(host_bufs[id(nxt)])       # gather of its CIGAR runs on the device, right behind its kernels
    t4 = time.time()
    c, _ = collect(cur)            # results of the finished set while `nxt` runs: device-to-host copies only
    t5 = time.time()
    ph["reload"] += t2 - t1; ph["wait"] += t3 - t2; ph["launch"] += t4 - t3; ph["collect"] += t5 - t4
    cells += c
    cur, nxt = nxt, cur
dt = time.time() - t0
print(f"collect split (s, all loops): results {tc[0]:.3f} cigars {tc[1]:.3f}")
print(f"overlapped: {sets} sets of {n} pairs in {dt:.3f} s -> {cells / dt / 1e9:.0f} GCUPS end to end; host phases per set (ms): " + ", ".join(f"{k} {v / sets * 1e3:.0f}" for k, v in ph.items()))
a.close(); b.close()
