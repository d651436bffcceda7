#!/usr/bin/env python3
"""bench.py — block-aligner hot path on MI355X: GCUPS on the 10 kbp DNA X-drop + traceback batch (BASELINE.json config 3).

A "step" is one pass of the batch launcher (one persistent kernel launch, one wavefront per pair) over the whole
per-GPU batch; inputs (PaddedBytes images, matrix, scratch) are resident in HBM before the timed region. Pairs shard
embarrassingly: with --gpus N every rank aligns its own batch of --pairs pairs (weak scaling, no collective on the
data path; torch.distributed only provides the barrier and the max-over-ranks of the elapsed time).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0. GCUPS counts computed DP cells: the sum over every block-fill call of columns
iterated x block height, counted on the device and checked against the oracle.

Besides the headline the line carries (N = 1 only): `roofline` for the binding resource of the dominant kernel (packed-int16
VALU issue; the HBM view sits in `roofline.hbm`), `cpu_baseline` (the AVX2 oracle on the host cores, which doubles as a
bit-exact check of scores, end positions, cell counts and CIGAR runs), `secondary` (BASELINE.json configs 2, 4, 5 at reduced
counts on the same kernels, each oracle-checked) and the secondary metrics SURVEY.md 8(d) asks for.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# roofline constants. A wave64 packed-int16 VALU instruction (v_pk_add_i16 / v_pk_max_i16 / any VOP3, DPP or SDWA form)
# occupies its SIMD for 4 cycles -- measured, profiles/r02_valu_rate.md -- i.e. 64 lanes x 2 halves per CU per clock:
# 256 CUs x 64 x 2 x 2.4e9 = 78.6 T int16-op/s. HBM3E: 8 TB/s (MI355X_MICROARCH.md).
HBM_PEAK_GBS = 8000.0
VALU_PEAK_INT16_TOPS = 256 * 64 * 2 * 2.4e9 / 1e12
BYTES_PER_CELL = {False: 0.008, True: 0.52}        # algorithmic HBM bytes per cell without / with trace


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=100000, help="pairs per GPU (config 3: 100k)")
    ap.add_argument("--len", type=int, default=10000)
    ap.add_argument("--edits", type=int, default=1000)
    ap.add_argument("--tail", type=int, default=500)
    ap.add_argument("--strong", action="store_true", help="strong scaling: ONE list of --pairs pairs, cut into cost-balanced slices, one per rank "
                                                        "(default: weak scaling, --pairs pairs per rank)")
    ap.add_argument("--multibatch", action="store_true", help="ONE process, the library's own multi-GPU launcher (ba_multibatch_*: one list of pairs cut into "
                                                            "cost-balanced slices, one per device) instead of one process per GPU under torch.distributed.run")
    ap.add_argument("--no-trace", action="store_true", help="score-only variant (not the headline workload)")
    ap.add_argument("--cpu-baseline-pairs", type=int, default=0, help="0 = size the sample for ~15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config 2 / 4 / 5 lines")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive pass (e2e_gcups)")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes for input generation (0 = auto; use 1 under rocprofv3)")
    return ap.parse_args()


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota if there is one (a pod
    that sees 256 logical CPUs but is throttled to a few would otherwise report a meaningless core count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())            # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def oracle_compare(np, o, w, res, runs, off, n_s, threads):
    """Bit-exact comparison of the first n_s pairs of a sequence-sequence workload with the oracle: score, both end
    indices, computed cells and -- with traceback -- every CIGAR run. Returns (oracle result dict, runs compared)."""
    sub = w.pairs.subset(np.arange(n_s))
    trace = "trace" in w.mode
    ref = o.batch_align(w.matrix, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, w.gaps, w.size, w.x_drop, w.mode,
                        cigar_eq=w.cigar_eq and trace, threads=threads)
    ok = (np.array_equal(ref["scores"], res["score"][:n_s]) and np.array_equal(ref["query_idx"], res["query_idx"][:n_s])
          and np.array_equal(ref["reference_idx"], res["reference_idx"][:n_s]) and int(res["cells"][:n_s].sum()) == ref["cells"])
    n_runs = 0
    if trace and ok:
        ok = np.array_equal(ref["cig_len"], res["cigar_len"][:n_s])
        if ok:
            ln = ref["cig_len"].astype(np.int64)
            n_runs = int(ln.sum())
            # index of every oracle run: pair p's runs sit at cig_off[p] .. cig_off[p] + cig_len[p]
            start = np.repeat(ref["cig_off"].astype(np.int64), ln)
            within = np.arange(n_runs, dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
            ok = np.array_equal(ref["cig_ops"][start + within], runs[: int(off[n_s])])
    if not ok:
        raise RuntimeError(f"bench.py: GPU results differ from the oracle on {w.name}; number is invalid")
    return ref, n_runs


def self_check(np, w, res, runs, off, idx):
    """Oracle-independent: the HIP CIGARs of pairs `idx` are paths that end at the HIP end positions and re-score to the HIP scores."""
    from block_aligner_amd.verify import check_cigar
    for p in idx:
        check_cigar(runs[int(off[p]): int(off[p + 1])], w.pairs.query(p), w.pairs.reference(p), w.matrix, w.gaps, int(res["score"][p]),
                    int(res["query_idx"][p]), int(res["reference_idx"][p]), w.mode, what=(w.name, int(p)))
    return len(idx)


def secondary_line(np, H, W, o, w, cores):
    """One BASELINE.json side configuration on the same kernels: two untimed launches (these launches take 1 .. 30 ms: the first ones of a
    batch run at clocks and caches that have not settled -- C4: 4.5, 4.5, 4.2, 4.0, 4.0 ... ms), then eight timed ones; `gcups` is the best
    of them (as in rounds 1 - 3: best of 3), `gcups_median` their median (launch-to-launch spread: C2 8.3 .. 9.1 ms). EVERY pair is
    compared with the oracle (score, end positions, computed cells, CIGAR runs)."""
    b = W.make_batch(H, w)
    for _ in range(2):
        b.run()
    times = sorted(b.run() for _ in range(8))
    ms, ms_median = times[0], 0.5 * (times[3] + times[4])
    res = b.results()
    kernel = b.info()["kernel"]
    if res["status"].any():
        raise RuntimeError(f"bench.py: {w.name}: pairs failed on the device")
    cells = int(res["cells"].sum())
    n = len(w.pairs)
    trace = "trace" in w.mode
    runs, off = b.cigars(res["cigar_len"]) if trace else (None, None)
    retried = b.retried()
    checked = 0
    if o is not None:
        if w.profiles:
            ref = o.batch_align_profile(w.pairs.pool, w.pairs.q_off, w.pairs.q_len, w.profiles, w.size, w.x_drop, w.mode, threads=cores)
            ok = (np.array_equal(ref["scores"], res["score"]) and np.array_equal(ref["query_idx"], res["query_idx"])
                  and np.array_equal(ref["reference_idx"], res["reference_idx"]) and np.array_equal(ref["cells"], res["cells"].astype(np.uint64)))
            if ok and trace:
                ok = np.array_equal(ref["cig_len"], res["cigar_len"])
                if ok:
                    ln = ref["cig_len"].astype(np.int64)
                    start = np.repeat(ref["cig_off"].astype(np.int64), ln)
                    within = np.arange(int(ln.sum()), dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
                    ok = np.array_equal(ref["cig_ops"][start + within], runs[: int(off[n])])
            if not ok:
                raise RuntimeError(f"bench.py: GPU results differ from the oracle on {w.name}; number is invalid")
            checked = n
        else:
            checked = n
            oracle_compare(np, o, w, res, runs, off, checked, cores)
            if trace:
                self_check(np, w, res, runs, off, range(0, checked, max(1, checked // 256)))
    b.close()
    gc = cells / (ms * 1e-3) / 1e9
    return {"config": w.name, "kernel": kernel, "gcups": round(gc, 1), "valu_frac": round(gc * 1e9 * w.ops_per_cell / 1e12 / VALU_PEAK_INT16_TOPS, 4),
            "ops_per_cell": w.ops_per_cell, "kernel_ms": round(ms, 3), "kernel_ms_median": round(ms_median, 3),
            "gcups_median": round(cells / (ms_median * 1e-3) / 1e9, 1), "launches": 8, "pairs": n, "m_pairs_per_s": round(n / (ms * 1e-3) / 1e6, 3),
            "full_matrix_equiv_gcups": round(w.full_matrix_cells() / (ms * 1e-3) / 1e9, 1), "parity_checked_pairs": checked, "retried": retried}


def secondary_sized(np, H, W, o, cores, n=20000):
    """Every pair with its own block range (ba_sized_batch_*: percent_len 1 % .. 10 % of the pair's length, examples/nanopore_bench_global.rs:144-171)
    on a mixed-length read set: the pairs are binned by range, the bins are launched together (the reported time is the host's wall clock from the first launch to the last completion, not a sum of HIP-event times); `gcups` counts all of them. EVERY pair is compared
    with the oracle run with its own range (bin by bin)."""
    from block_aligner_amd import scores as S
    pairs = W.mixed_reads(n)
    m = S.NucMatrix.new_simple(2, -3)
    b = H.SizedBatchAligner(m, (-5, -1), 100, H.TRACE | H.X_DROP | H.CIGAR_EQ, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, percent=(0.01, 0.1))
    b.run()
    times = sorted(b.run() for _ in range(3))
    ms = times[0]
    res = b.results()
    if res["status"].any():
        raise RuntimeError("bench.py: sized batch: pairs failed on the device")
    runs, off = b.cigars(res["cigar_len"])
    classes = b.classes()
    checked = 0
    if o is not None:
        ln = np.maximum(pairs.q_len, pairs.r_len)
        lo = np.array([H.percent_len(int(x), 0.01) for x in ln]); hi = np.array([H.percent_len(int(x), 0.1) for x in ln])
        for key in sorted(set(zip(lo.tolist(), hi.tolist()))):
            idx = np.nonzero((lo == key[0]) & (hi == key[1]))[0]
            sub = pairs.subset(idx)
            ref = o.batch_align(m, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, (-5, -1), key, 100, ("trace", "x_drop"), cigar_eq=True, threads=cores)
            ok = (np.array_equal(ref["scores"], res["score"][idx]) and np.array_equal(ref["query_idx"], res["query_idx"][idx])
                  and np.array_equal(ref["reference_idx"], res["reference_idx"][idx]) and int(res["cells"][idx].sum()) == ref["cells"]
                  and np.array_equal(ref["cig_len"], res["cigar_len"][idx]))
            if ok:
                for kk in range(0, len(idx), max(1, len(idx) // 64)):   # (the runs of a spread of every bin's pairs; all lengths were compared above)
                    p = int(idx[kk])
                    want = ref["cig_ops"][int(ref["cig_off"][kk]): int(ref["cig_off"][kk]) + int(ref["cig_len"][kk])]
                    ok = ok and np.array_equal(want, runs[int(off[p]): int(off[p + 1])])
            if not ok:
                raise RuntimeError(f"bench.py: GPU results differ from the oracle on the sized batch, block range {key}; number is invalid")
            checked += len(idx)
    b.close()
    cells = int(res["cells"].sum())
    gc = cells / (ms * 1e-3) / 1e9
    return {"config": f"per-pair block ranges: {n} reads of 1..40 kbp (log-uniform), 10 % edits, percent_len 1 %..10 % per pair, X-drop 100, traceback",
            "kernel": "ba_sized_batch: " + ", ".join(f"{c[0]}..{c[1]} x{c[2]}" for c in classes), "gcups": round(gc, 1),
            "valu_frac": round(gc * 1e9 * 20 / 1e12 / VALU_PEAK_INT16_TOPS, 4), "ops_per_cell": 20, "kernel_ms": round(ms, 3), "launches": 3, "pairs": n,
            "parity_checked_pairs": checked, "ranges": len(classes)}


def measure_e2e(np, H, W, w, sets=4):
    """PCIe-inclusive rate (never `value`): `sets` sets of the batch stream from host memory through two batch objects -- while one
    set is aligned the host reloads the other object with the next set; scores, end positions and the gathered CIGAR runs (written by
    the device straight into page-locked host buffers behind the kernels) come back to host arrays."""
    import time as _t
    p = w.pairs
    args = (p.pool, p.q_off, p.q_len, p.r_off, p.r_len)
    bs = [W.make_batch(H, w), W.make_batch(H, w)]
    cap = int(p.q_len.astype(np.int64).sum() + p.r_len.astype(np.int64).sum()) // 8 + (1 << 20)   # runs per set: far below one run per 8 residues
    bufs = [H.pinned_array(cap), H.pinned_array(cap)]
    for b in bs:
        b.run()
    t0 = _t.perf_counter()
    cells = 0
    cur, nxt = 0, 1
    bs[cur].reload(*args); bs[cur].launch(); bs[cur].compact_cigars(bufs[cur])
    for s in range(sets):
        if s + 1 < sets:
            bs[nxt].reload(*args)
        bs[cur].wait()
        if s + 1 < sets:
            bs[nxt].launch(); bs[nxt].compact_cigars(bufs[nxt])
        res = bs[cur].results()
        runs, off = bs[cur].cigars(res["cigar_len"], out=bufs[cur])
        if res["status"].any():
            raise RuntimeError("bench.py: pairs failed on the device in the end-to-end pass")
        cells += int(res["cells"].sum())
        cur, nxt = nxt, cur
    dt = _t.perf_counter() - t0
    cig_mb = round(int(off[-1]) * 4 / 1e6, 1)
    del runs
    for b in bs:
        b.close()
    for buf in bufs:
        H.free_pinned(buf)
    return {"gcups": round(cells / dt / 1e9, 1), "sets": sets, "pairs_per_set": len(p), "seconds": round(dt, 3), "cigar_mb_per_set": cig_mb,
            "what": "host pool -> device (reload) -> align -> scores + CIGAR runs in host memory, two batch objects alternating; PCIe and host packing included"}


def run_multibatch(a):
    """--multibatch: one process drives --gpus devices through the library's own launcher (ba_multibatch_*: ba_host.cpp ba_shard_slices cuts
    ONE pair list into contiguous cost-balanced slices, one batch per device, each built by its own host thread and launched on its own
    stream; results come back in the caller's order; no collective on the data path). Weak scaling by default (--pairs per device),
    --strong for one list of --pairs pairs. Same timed region and the same JSON line as the torchrun form."""
    import numpy as np
    from block_aligner_amd import workloads as W
    t0 = time.time()
    trace = not a.no_trace
    n_total = a.pairs if a.strong else a.pairs * a.gpus
    w = W.config3(n_total, a.len, a.edits, a.tail, seed=1234, trace=trace, workers=a.gen_workers or min(32, usable_cpus()))
    t_gen = time.time() - t0
    import torch
    from block_aligner_amd import hip as H
    if H.device_count() < a.gpus:
        raise RuntimeError(f"bench.py --multibatch --gpus {a.gpus}: only {H.device_count()} HIP devices visible (there is no CPU fallback)")
    size = (H.percent_len(a.len, 0.01), H.percent_len(a.len, 0.1))
    mode = H.X_DROP | ((H.TRACE | H.CIGAR_EQ) if trace else 0)
    p = w.pairs
    t0 = time.time()
    mb = H.MultiBatchAligner(w.matrix, w.gaps, size, w.x_drop, mode, p.pool, p.q_off, p.q_len, p.r_off, p.r_len, list(range(a.gpus)))
    t_setup = time.time() - t0

    def sync_all():
        for d in range(a.gpus):
            torch.cuda.synchronize(d)

    for _ in range(a.warmup):
        mb.run()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms, per_dev = [], []
    for _ in range(a.steps):                             # (each run launches every device's slice, then waits for all of them)
        kernel_ms.append(mb.run())
        per_dev.append(mb.kernel_ms())
    sync_all()
    elapsed = time.perf_counter() - t0
    res = mb.results()
    if res["status"].any():
        raise RuntimeError(f"{int((res['status'] != 0).sum())} pairs failed on the device")
    cells = int(res["cells"].sum())
    gcups = cells * a.steps / elapsed / 1e9
    k_ms = float(np.mean(kernel_ms))                     # the slowest device's kernel time per step (HIP events on its stream)
    ops = w.ops_per_cell
    tops = cells / (k_ms * 1e-3) * ops / 1e12
    parts = [int(x) for x in mb.parts()]
    rescored = 0
    runs = off = None
    if trace:
        runs, off = mb.cigars(res["cigar_len"])
        rescored = self_check(np, w, res, runs, off, range(0, n_total, max(1, n_total // 256)))
    mb.close()
    # the oracle as checker and reported CPU baseline, on a bounded sample: the first pairs of EVERY device's slice (the checker runs after
    # the timed region and is never the thing measured)
    cpu = None
    if not a.no_cpu_baseline:
        from oracle.oracle_py import Oracle, build
        build()
        o = Oracle("avx2")
        cores = usable_cpus()
        per = a.cpu_baseline_pairs or max(16, min(8000 * cores, 32000) // a.gpus)
        w.size = size
        secs = cells_cpu = checked = runs_checked = 0
        for d in range(a.gpus):
            lo = parts[d]; hi = min(parts[d + 1], lo + per)
            if hi <= lo:
                continue
            idx = np.arange(lo, hi)
            sub_w = W.Workload(w.name, p.subset(idx), w.matrix, w.gaps, size, w.x_drop, w.mode, cigar_eq=w.cigar_eq)
            sub_res = {k: v[lo:hi] for k, v in res.items()}
            sub_runs = sub_off = None
            if trace:
                sub_off = off[lo:hi + 1] - off[lo]
                sub_runs = runs[int(off[lo]): int(off[hi])]
            ref, nr = oracle_compare(np, o, sub_w, sub_res, sub_runs, sub_off, hi - lo, cores)
            secs += ref["seconds"]; cells_cpu += ref["cells"]; checked += hi - lo; runs_checked += nr
        cpu = {"value": round(cells_cpu / secs / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
               "sample": f"the first {per} pairs of each of the {a.gpus} slices, {cores} threads, {secs:.1f} s; AVX2 restatement of block-aligner v0.5.1 "
                         f"(oracle/), not the Rust crate", "cpu_model": cpu_model(), "parity_checked_pairs": checked, "cigar_runs_checked": runs_checked}
    out = {
        "metric": "GCUPS (DP cells/s) on 10 kbp DNA X-drop batch; bit-exact score+CIGAR vs AVX2",
        "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if a.strong else "weak",
        "vs_baseline": None, "dtype": "i16 (saturating lane scores) + i32 block offsets", "data": "synthetic",
        "config": {"workload": f"config 3: {a.pairs} pairs{' in total' if a.strong else '/GPU'} x {a.len} bp random DNA, {a.edits} edits, +{a.tail} bp random tails, "
                               f"NucMatrix(2,-3), gaps(-5,-1), X-drop {w.x_drop}, block {size[0]}..{size[1]}, " + ("traceback to =/X CIGAR" if trace else "score only"),
                   "launcher": "ba_multibatch_* (one process, one host thread and one stream per device)", "pairs_total": n_total, "slice_bounds": parts,
                   "block": list(size), "trace": trace, "parallelism": f"{a.gpus} x independent shard", "gen_s": round(t_gen, 1), "setup_s": round(t_setup, 1),
                   "cells_per_step": cells},
        "pairs_per_s": round(n_total * a.steps / elapsed, 1), "computed_cells": cells, "cigars_rescored": rescored,
        "roofline": {"bound": "valu-int16", "achieved": round(tops, 3), "peak": round(VALU_PEAK_INT16_TOPS * a.gpus, 1), "unit": "Tint16-op/s",
                     "frac": round(tops / (VALU_PEAK_INT16_TOPS * a.gpus), 5), "traffic": None, "kernel_ms": round(k_ms, 3),
                     "kernel_ms_per_device": [round(float(x), 3) for x in np.mean(np.array(per_dev, np.float64), axis=0)],
                     "note": "all devices together; kernel_ms = the slowest device's slice per step (HIP events on every slice's own stream); "
                             "traffic: PMC passes are collected on one device (the N = 1 line quotes them)"},
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    if a.multibatch:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            print("bench.py --multibatch is the one-process form: run it without torch.distributed.run", file=sys.stderr)
            sys.exit(2)
        return run_multibatch(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if a.gpus > 1 and world == 1:
            # `python bench.py --gpus N` without a launcher: the one-process form (ba_multibatch_*: one host thread and one stream per
            # device). Nothing has touched a GPU yet and nothing is re-launched: the same workload, timed region and JSON line.
            print(f"bench.py: --gpus {a.gpus} without torch.distributed.run: one process, the library's multi-GPU launcher (--multibatch)", file=sys.stderr)
            return run_multibatch(a)
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {a.gpus}", file=sys.stderr)

    import numpy as np
    from block_aligner_amd import workloads as W
    from block_aligner_amd.shard import job_slice, reduce_job, shard_seed

    # ---- synthetic workload (forked workers: must happen before any GPU initialisation)
    t0 = time.time()
    workers = a.gen_workers or min(32, max(1, usable_cpus() // max(1, world)))
    trace = not a.no_trace
    if a.strong:
        # one global list (the same on every rank: same seed), of which this rank aligns its cost-balanced contiguous slice
        # (block_aligner_amd/shard.py balanced_slices = the library's ba_shard_slices rule; no collective on the data path)
        w = W.config3(a.pairs, a.len, a.edits, a.tail, seed=1234, trace=trace, workers=workers)
        sub, lo, hi = job_slice(w.pairs, rank, world)
        if sub is None:
            raise RuntimeError(f"rank {rank}: empty slice ({a.pairs} pairs over {world} ranks)")
        w.pairs = sub
    else:
        w = W.config3(a.pairs, a.len, a.edits, a.tail, seed=shard_seed(1234, rank), trace=trace, workers=workers)
    pairs = w.pairs
    n_rank = len(pairs)
    t_gen = time.time() - t0

    import torch
    import torch.distributed as dist
    from block_aligner_amd import hip as H
    if H.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    H.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    w.size = (H.percent_len(a.len, 0.01), H.percent_len(a.len, 0.1))   # 1 % .. 10 % of the length -> 128 .. 1024
    size = w.size

    t0 = time.time()
    batch = W.make_batch(H, w)
    t_setup = time.time() - t0
    info = batch.info()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        batch.run()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(a.steps):
        kernel_ms.append(batch.run())          # launches on the library's stream and waits for it
    sync_all()
    elapsed = time.perf_counter() - t0
    res = batch.results()
    retried_main = batch.retried()
    if res["status"].any():
        raise RuntimeError(f"rank {rank}: {int((res['status'] != 0).sum())} pairs failed on the device")
    cells_rank = int(res["cells"].sum())
    elapsed, cells_total = reduce_job(elapsed, float(cells_rank), device="cuda")

    if rank == 0:
        gcups = cells_total * a.steps / elapsed / 1e9
        k_ms = float(np.mean(kernel_ms))
        gcups_kernel = cells_rank / (k_ms * 1e-3) / 1e9          # this rank's kernel alone, from HIP events on its stream
        achieved_gbs = BYTES_PER_CELL[trace] * cells_rank / (k_ms * 1e-3) / 1e9
        # measured HBM bytes per launch: only from a profile record taken on THESE kernel sources (tools/profile_round.sh writes
        # profiles/rNN_roofline.json with the hash of the sources it ran; a record of another build is not quoted)
        traffic = None
        profile_ref = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_hash import kernel_hash
            here = kernel_hash()
            recs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_roofline.json"))
            for f in reversed(recs):
                t = json.load(open(os.path.join(ROOT, "profiles", f)))
                if t.get("kernel_sources_sha16") == here and t.get("pairs") == a.pairs and t.get("kind") == ("xdrop+trace" if trace else "xdrop"):
                    traffic = t.get("hbm_bytes_per_launch")
                    profile_ref = {"file": "profiles/" + f, "commit": t.get("commit"), "kernel_sources_sha16": here, "valu_busy_per_simd": t.get("valu_busy_per_simd"),
                                   "sq_insts_valu": (t.get("sq") or {}).get("SQ_INSTS_VALU")}
                    break
            if profile_ref is None:
                profile_ref = {"file": None, "kernel_sources_sha16": here, "note": "no profile record of these kernel sources under profiles/: traffic not quoted"}
        except Exception as e:
            profile_ref = {"file": None, "note": f"profile record unreadable: {e}"}
        ops = w.ops_per_cell
        tops = gcups_kernel * 1e9 * ops / 1e12
        # what the launch had to compute: the speculative, untraced rectangles (the chain of grows that closes an X-drop alignment: on no
        # path) need the X-drop recurrence only (14 ops per cell), every other cell the full count. `achieved` / `frac` above price EVERY
        # cell at the full count -- reference-equivalent work; ops_required is the issued-useful view, valu_overhead the instructions the
        # launch executed per required operation (SQ_INSTS_VALU from the profile record of these sources x 128 int16 lanes-halves)
        spec_cells = batch.spec_cells() if trace else 0
        ops_required = (cells_rank - spec_cells) * ops + spec_cells * W.OPS_PER_CELL[("x_drop",)]
        roofline = {"bound": "valu-int16", "achieved": round(tops, 3), "peak": round(VALU_PEAK_INT16_TOPS, 1), "unit": "Tint16-op/s",
                    "frac": round(tops / VALU_PEAK_INT16_TOPS, 5), "traffic": traffic, "profile": profile_ref,
                    "kernel": "ba::%s<class %d, NUC, trace=%d, xdrop=1>" % (info["kernel"], size[1] // 128, int(trace)), "kernel_ms": round(k_ms, 3),
                    "algorithmic_ops_per_cell": ops, "kernel_gcups": round(gcups_kernel, 1),
                    "speculative_untraced_cells": spec_cells, "ops_required": ops_required,
                    "frac_of_required": round(ops_required / (k_ms * 1e-3) / 1e12 / VALU_PEAK_INT16_TOPS, 5),
                    "valu_overhead": (round(profile_ref["sq_insts_valu"] * 128 / ops_required, 3) if profile_ref and profile_ref.get("sq_insts_valu") else None),
                    "hbm": {"achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 5),
                            "algorithmic_bytes_per_cell": BYTES_PER_CELL[trace]},
                    "note": "integer max-plus recurrence: bound by packed-int16 VALU issue (4 cycles per wave64 instruction, "
                            "profiles/r02_valu_rate.md), neither HBM nor MFMA; traffic = measured HBM bytes per launch (PMC)"}

        surviving = int(batch.surviving_cells().sum()) if trace else None
        full_equiv = w.full_matrix_cells()
        runs = off = None
        if trace:
            runs, off = batch.cigars(res["cigar_len"])
        # oracle-independent self-check of the HIP output: a spread of pairs' CIGARs re-walked and re-scored
        rescored = self_check(np, w, res, runs, off, range(0, n_rank, max(1, n_rank // 256))) if trace else 0

        batch.close()      # the 10 kbp batch's trace arena is most of the device memory: free it before the side configurations
        e2e = None
        if trace and world == 1 and not a.no_e2e:
            try:
                e2e = measure_e2e(np, H, W, w)
            except Exception as ex:   # (reported, never fatal for the headline: two arenas may not fit beside another process)
                e2e = {"gcups": None, "error": str(ex)[:200]}
        # ---- parity check + CPU baseline (the oracle is the checker and the "port" baseline, never the product)
        cpu = None
        secondary = None
        runs_checked = 0
        if not a.no_cpu_baseline and world == 1:
            from oracle.oracle_py import Oracle, build
            build()
            o = Oracle("avx2")
            cores = usable_cpus()
            n_s = a.cpu_baseline_pairs or min(n_rank, max(64, 8000 * cores))   # ~1 ms per pair per core -> <= ~10 s on all usable cores
            ref, runs_checked = oracle_compare(np, o, w, res, runs, off, n_s, cores)
            n1 = min(n_s, 2000)                                                  # ~6 s on one core
            sub1 = pairs.subset(np.arange(n1))
            ref1 = o.batch_align(w.matrix, sub1.pool, sub1.q_off, sub1.q_len, sub1.r_off, sub1.r_len, w.gaps, size, w.x_drop, w.mode,
                                 cigar_eq=True, threads=1)
            # sanity anchor (first row): the reference's nanopore_bench at block 32-32 without trace
            # took 239.7 us per 10 kbp pair on its authors' (unstated) machine, one thread
            n32 = min(n_s, 1000)
            sub32 = pairs.subset(np.arange(n32))
            ref32 = o.batch_align(w.matrix, sub32.pool, sub32.q_off, sub32.q_len, sub32.r_off, sub32.r_len, w.gaps, (32, 32), w.x_drop,
                                  ("x_drop",), threads=1)
            cpu = {"value": round(ref["cells"] / ref["seconds"] / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
                   "sample": f"first {n_s} pairs of the same batch, {cores} threads, {ref['seconds']:.1f} s; AVX2 restatement of "
                             f"block-aligner v0.5.1 (oracle/), not the Rust crate",
                   "cpu_model": cpu_model(),
                   "single_thread_gcups": round(ref1["cells"] / ref1["seconds"] / 1e9, 3),
                   "parity_checked_pairs": n_s, "cigar_runs_checked": runs_checked,
                   "sanity_block32_us_per_pair": round(ref32["seconds"] / n32 * 1e6, 1),
                   "sanity_reference_notebook_us_per_pair": 239.7}
            if not a.no_secondary:
                # (counts large enough that every resident wave sees a few dozen pairs: these launches last 5 - 30 ms) ...
                c2, c4 = W.config2(200000, workers=1), W.config4(400000)
                secondary = [secondary_line(np, H, W, o, c2, cores)]
                c2.mode = ("trace", "x_drop"); c2.name += ", traceback"
                secondary.append(secondary_line(np, H, W, o, c2, cores))
                secondary.append(secondary_line(np, H, W, o, c4, cores))
                c4.mode = ("trace",); c4.name += ", traceback"
                secondary.append(secondary_line(np, H, W, o, c4, cores))
                secondary.append(secondary_line(np, H, W, o, W.config5(80000), cores))
                # the special alignment modes (scan_block.rs:89: LOCAL_START, FREE_QUERY_END_GAPS), on the per-pair kernel
                secondary.append(secondary_line(np, H, W, o, W.config_local(50000), cores))
                secondary.append(secondary_line(np, H, W, o, W.config_free_end(50000), cores))
                # every pair with its own block range (round 5: ba_sized_batch_*)
                secondary.append(secondary_sized(np, H, W, o, cores))
                # the headline workload at the batch sizes 8 and 4 GPUs get of the north-star's 100 000 pairs (BASELINE.json configs[2]: "1 -> 8 GPUs";
                # the first pairs of the same seeded set): what strong scaling would start from -- a launch of one or two rounds of the machine's
                # 15 360 slots, bound by a pair's chain of steps and one path's walk rather than by issue slots (round 6)
                if trace and not a.strong:
                    for n_part in (12500, 25000):
                        wp = W.config3(n_part, a.len, a.edits, a.tail, seed=1234, trace=True, workers=min(8, usable_cpus()), size=size)
                        wp.name += " (%d of the north-star's 100 000: the share of %d GPUs)" % (n_part, 100000 // n_part)
                        secondary.append(secondary_line(np, H, W, o, wp, cores))
                    # reads above 12.8 kbp: percent_len (lib.rs:109-111) starts them at 256 cells -- 13 kbp pairs at 1 % .. 10 % = 256 .. 2048 as
                    # examples/nanopore_bench_global.rs:144-171 sizes them; round 6: k_multi with two slots of 256 cells per wave
                    wl = W.config3(30000, 13000, 1300, a.tail, seed=4321, trace=True, workers=min(8, usable_cpus()), size=(H.percent_len(13000, 0.01), H.percent_len(13000, 0.1)))
                    wl.name += ", block %d..%d (percent_len 1 %% .. 10 %% of 13 kbp)" % wl.size
                    secondary.append(secondary_line(np, H, W, o, wl, cores))
                    # ... and reads above 25.6 kbp at 512 cells: 32 kbp pairs at 1 % .. 10 % = 512 .. 4096 -- a range that ends in the row-tiled class and is
                    # launched in the 2048-cell one (round 6: the class bet), k_multi with one slot of 512 cells per wave
                    wl = W.config3(5000, 32000, 3200, a.tail, seed=4322, trace=True, workers=min(8, usable_cpus()), size=(H.percent_len(32000, 0.01), H.percent_len(32000, 0.1)))
                    wl.name += ", block %d..%d (percent_len 1 %% .. 10 %% of 32 kbp)" % wl.size
                    secondary.append(secondary_line(np, H, W, o, wl, cores))
                # ... and the same configurations at the batch sizes of the reference's own harnesses (BASELINE.json: 10 k pairs,
                # examples/nanopore_bench.rs:73-95; 7 k protein pairs, examples/uc_bench.rs:79-104; 11 k PSSMs, examples/pssm_bench.rs:94-100):
                # a few pairs per wave, bound by the longest pair's chain of steps rather than by the machine
                c2s, c4s = W.config2(10000, workers=1), W.config4(7000)
                secondary.append(secondary_line(np, H, W, o, c2s, cores))
                c2s.mode = ("trace", "x_drop"); c2s.name += ", traceback"
                secondary.append(secondary_line(np, H, W, o, c2s, cores))
                secondary.append(secondary_line(np, H, W, o, c4s, cores))
                c4s.mode = ("trace",); c4s.name += ", traceback"
                secondary.append(secondary_line(np, H, W, o, c4s, cores))
                secondary.append(secondary_line(np, H, W, o, W.config5(11000), cores))
        out = {
            "metric": "GCUPS (DP cells/s) on 10 kbp DNA X-drop batch; bit-exact score+CIGAR vs AVX2",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if a.strong else "weak",
            "vs_baseline": None, "dtype": "i16 (saturating lane scores) + i32 block offsets", "data": "synthetic",
            "config": {"workload": f"config 3: {a.pairs} pairs{' in total' if a.strong else '/GPU'} x {a.len} bp random DNA, {a.edits} edits, +{a.tail} bp random tails, "
                                   f"NucMatrix(2,-3), gaps(-5,-1), X-drop {w.x_drop}, block {size[0]}..{size[1]}, "
                                   + ("traceback to =/X CIGAR" if trace else "score only"),
                       "pairs_per_gpu": n_rank, "pairs_total": a.pairs if a.strong else a.pairs * world, "block": list(size), "trace": trace, "parallelism": f"{world} x independent shard",
                       "grid_waves": info["grid"], "lds_bytes_per_wave": info["lds_bytes_per_wave"],
                       "trace_arena_gb": round(info["trace_arena_bytes"] / 1e9, 2), "gen_s": round(t_gen, 1), "setup_s": round(t_setup, 1),
                       "cells_per_step": cells_total},
            "pairs_per_s": round((a.pairs if a.strong else a.pairs * world) * a.steps / elapsed, 1),
            "full_matrix_equiv_gcups": round(full_equiv * (1 if a.strong and world == 1 else world) * a.steps / elapsed / 1e9, 1),
            "computed_cells": cells_rank, "surviving_cells": surviving, "retried": retried_main,
            "cigar_runs_checked": runs_checked, "cigars_rescored": rescored, "e2e_gcups": (e2e or {}).get("gcups"), "e2e": e2e,
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary,
        }
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
