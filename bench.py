#!/usr/bin/env python3
"""bench.py — block-aligner hot path on MI355X: GCUPS on the 10 kbp DNA X-drop + traceback batch (BASELINE.json config 3).

A "step" is one pass of the batch launcher (one persistent kernel launch, one wavefront per pair) over the whole
per-GPU batch; inputs (PaddedBytes images, matrix, scratch) are resident in HBM before the timed region. Pairs shard
embarrassingly: with --gpus N every rank aligns its own batch of --pairs pairs (weak scaling, no collective on the
data path; torch.distributed only provides the barrier and the max-over-ranks of the elapsed time).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0. GCUPS counts computed DP cells: the sum over every block-fill call of columns
iterated x block height, counted on the device and checked against the oracle in the tests.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# roofline constants (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz, HBM3E 8 TB/s). A wave64 VALU instruction
# occupies its SIMD for 4 cycles (measured for v_pk_add_i16 / v_pk_max_i16 / v_max_i32_dpp, tools/dev/valu_rate.hip),
# i.e. 64 lanes per CU per clock; packed i16 doubles that: 256 x 64 x 2 x 2.4e9 = 78.6 T int16-op/s.
HBM_PEAK_GBS = 8000.0
VALU_PEAK_INT16_TOPS = 256 * 64 * 2 * 2.4e9 / 1e12
OPS_PER_CELL = {"xdrop": 14, "xdrop+trace": 20, "global": 11}      # algorithmic int16 ops per DP cell
BYTES_PER_CELL = {"xdrop": 0.008, "xdrop+trace": 0.52, "global": 0.008}   # algorithmic HBM bytes per cell


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=100000, help="pairs per GPU (config 3: 100k)")
    ap.add_argument("--len", type=int, default=10000)
    ap.add_argument("--edits", type=int, default=1000)
    ap.add_argument("--tail", type=int, default=500)
    ap.add_argument("--no-trace", action="store_true", help="score-only variant (not the headline workload)")
    ap.add_argument("--cpu-baseline-pairs", type=int, default=0, help="0 = size the sample for ~15 s of CPU work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {a.gpus}", file=sys.stderr)
        if a.gpus > 1 and world == 1:
            sys.exit(2)

    import numpy as np
    from block_aligner_amd import scores as S, synth
    from block_aligner_amd.shard import reduce_job, shard_seed

    # ---- synthetic workload (forked workers: must happen before any GPU initialisation)
    t0 = time.time()
    workers = a.gen_workers or min(32, max(1, usable_cpus() // max(1, world)))
    pairs = synth.make_pairs(a.pairs, a.len, a.edits, a.tail, synth.DNA, seed=shard_seed(1234, rank), workers=workers)
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

    matrix = S.NucMatrix.new_simple(2, -3)       # examples/nanopore_bench.rs:73-79
    gaps, x_drop = (-5, -1), 100
    size = (H.percent_len(a.len, 0.01), H.percent_len(a.len, 0.1))   # 1 % .. 10 % of the length -> 128 .. 1024
    trace = not a.no_trace
    mode = H.X_DROP | (H.TRACE | H.CIGAR_EQ if trace else 0)
    kind = "xdrop+trace" if trace else "xdrop"

    t0 = time.time()
    batch = H.BatchAligner(matrix, gaps, size, x_drop, mode, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len)
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
    if res["status"].any():
        raise RuntimeError(f"rank {rank}: {int((res['status'] != 0).sum())} pairs failed on the device")
    cells_rank = int(res["cells"].sum())
    elapsed, cells_total = reduce_job(elapsed, float(cells_rank), device="cuda")

    out = None
    if rank == 0:
        gcups = cells_total * a.steps / elapsed / 1e9
        k_ms = float(np.mean(kernel_ms))
        gcups_kernel = cells_rank / (k_ms * 1e-3) / 1e9          # this rank's kernel alone, from HIP events on its stream
        alg_bytes = BYTES_PER_CELL[kind] * cells_rank
        achieved_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                t = json.load(open(tf))
                if t.get("pairs") == a.pairs and t.get("kind") == kind:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": "ba::k_align<8, NUC, trace=%d, xdrop=1>" % int(trace), "kernel_ms": round(k_ms, 3),
                    "algorithmic_bytes_per_cell": BYTES_PER_CELL[kind],
                    "note": "integer max-plus recurrence: VALU-issue bound by construction, neither HBM nor MFMA; see valu_roofline"}
        valu = {"bound": "valu-int16", "achieved": round(gcups_kernel * 1e9 * OPS_PER_CELL[kind] / 1e12, 3),
                "peak": round(VALU_PEAK_INT16_TOPS, 1), "unit": "Tint16-op/s",
                "frac": round(gcups_kernel * 1e9 * OPS_PER_CELL[kind] / 1e12 / VALU_PEAK_INT16_TOPS, 5),
                "algorithmic_ops_per_cell": OPS_PER_CELL[kind]}

        # ---- parity spot-check + CPU baseline (the oracle is the checker and the "port" baseline, never the product)
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            from oracle.oracle_py import Oracle, build
            build()
            o = Oracle("avx2")
            modes = ("trace", "x_drop") if trace else ("x_drop",)
            cores = usable_cpus()
            n_s = a.cpu_baseline_pairs or min(a.pairs, max(64, 8000 * cores))   # ~1 ms per pair per core -> <= ~10 s on all usable cores
            sub = pairs.subset(np.arange(n_s))
            ref = o.batch_align(matrix, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, gaps, size, x_drop, modes,
                                cigar_eq=True, threads=cores)
            n1 = min(n_s, 2000)                                                  # ~6 s on one core
            sub1 = pairs.subset(np.arange(n1))
            ref1 = o.batch_align(matrix, sub1.pool, sub1.q_off, sub1.q_len, sub1.r_off, sub1.r_len, gaps, size, x_drop, modes,
                                 cigar_eq=True, threads=1)
            ok = (np.array_equal(ref["scores"], res["score"][:n_s]) and np.array_equal(ref["query_idx"], res["query_idx"][:n_s])
                  and np.array_equal(ref["reference_idx"], res["reference_idx"][:n_s]))
            if trace:
                ok = ok and np.array_equal(ref["cig_len"], res["cigar_len"][:n_s])
            cells_ok = int(res["cells"][:n_s].sum()) == ref["cells"]
            if not (ok and cells_ok):
                raise RuntimeError("bench.py: GPU results differ from the oracle on the CPU-baseline sample; number is invalid")
            cpu = {"value": round(ref["cells"] / ref["seconds"] / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
                   "sample": f"first {n_s} pairs of the same batch, {cores} threads, {ref['seconds']:.1f} s; AVX2 restatement of "
                             f"block-aligner v0.5.1 (oracle/), not the Rust crate",
                   "single_thread_gcups": round(ref1["cells"] / ref1["seconds"] / 1e9, 3),
                   "parity_checked_pairs": n_s}
        out = {
            "metric": "GCUPS (DP cells/s) on 10 kbp DNA X-drop batch; bit-exact score+CIGAR vs AVX2",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i16 (saturating lane scores) + i32 block offsets", "data": "synthetic",
            "config": {"workload": f"config 3: {a.pairs} pairs/GPU x {a.len} bp random DNA, {a.edits} edits, +{a.tail} bp random tails, "
                                   f"NucMatrix(2,-3), gaps(-5,-1), X-drop {x_drop}, block {size[0]}..{size[1]}, "
                                   + ("traceback to =/X CIGAR" if trace else "score only"),
                       "pairs_per_gpu": a.pairs, "block": list(size), "trace": trace, "parallelism": f"{world} x independent shard",
                       "grid_waves": info["grid"], "lds_bytes_per_wave": info["lds_bytes_per_wave"],
                       "trace_arena_gb": round(info["trace_arena_bytes"] / 1e9, 2), "gen_s": round(t_gen, 1), "setup_s": round(t_setup, 1),
                       "cells_per_step": cells_total},
            "roofline": roofline, "valu_roofline": valu, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    batch.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
