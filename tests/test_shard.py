"""The N > 1 path: shard arithmetic and the elapsed/cells reductions, over gloo with world_size 2 on CPU."""
import os
import socket

import torch.multiprocessing as mp

from block_aligner_amd import shard, synth


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 100000):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_rank_shards_are_distinct_and_reproducible():
    a0 = synth.make_pairs(8, 200, 20, 10, seed=shard.shard_seed(1234, 0))
    a1 = synth.make_pairs(8, 200, 20, 10, seed=shard.shard_seed(1234, 1))
    b0 = synth.make_pairs(8, 200, 20, 10, seed=shard.shard_seed(1234, 0))
    assert a0.pool.tobytes() == b0.pool.tobytes()
    assert a0.pool.tobytes() != a1.pool.tobytes()


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    el, tot = shard.reduce_job(1.0 + rank, 10.0 * (rank + 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, el, tot))


def test_reduce_job_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert out == [(0, 2.0, 30.0), (1, 2.0, 30.0)]   # max elapsed, summed cells, identical on every rank


def test_cost_balanced_slices_match_the_library():
    """ba_shard_slices (what ba_multibatch_create cuts by) == shard.balanced_slices; slices are contiguous, cover everything
    and differ in cost by at most one pair."""
    import numpy as np
    from block_aligner_amd import hip as H
    rng = np.random.default_rng(2)
    for n in (1, 2, 17, 5000):
        ql = rng.integers(0, 3000, n).astype(np.uint32); rl = rng.integers(0, 3000, n).astype(np.uint32)
        if n == 5000:
            ql[:50] = 200000          # a few very long pairs at the front
        for parts in (1, 2, 3, 8):
            lib_b = [int(x) for x in H.shard_slices(ql, rl, parts)]
            assert lib_b == shard.balanced_slices(ql, rl, parts)
            assert lib_b[0] == 0 and lib_b[-1] == n and all(a <= b for a, b in zip(lib_b, lib_b[1:]))
            cost = ql.astype(np.int64) + rl + 16
            sums = [int(cost[a:b].sum()) for a, b in zip(lib_b, lib_b[1:])]
            if n >= parts * 4:
                assert max(sums) - min(sums) <= 2 * int(cost.max())


def _slice_worker(rank, world, port, q):
    """Each rank aligns its cost-balanced slice of ONE global pair list (the oracle stands in for the device on CPU) and
    rank 0 merges the slices in caller order."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from block_aligner_amd import scores as S
    from oracle.oracle_py import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pairs = synth.make_pairs(64, (50, 900), (0, 60), 20, seed=99)          # the same global list on every rank
    b = shard.balanced_slices(pairs.q_len, pairs.r_len, world)
    lo, hi = b[rank], b[rank + 1]
    sub = pairs.subset(np.arange(lo, hi))
    o = Oracle("avx2")
    m = S.NucMatrix.new_simple(2, -3)
    ref = o.batch_align(m, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, (-5, -1), (32, 128), 50, ("x_drop",))
    mine = torch.zeros(len(pairs), 3, dtype=torch.int64)
    mine[lo:hi, 0] = torch.from_numpy(ref["scores"].astype(np.int64)); mine[lo:hi, 1] = torch.from_numpy(ref["query_idx"].astype(np.int64))
    mine[lo:hi, 2] = torch.from_numpy(ref["reference_idx"].astype(np.int64))
    dist.reduce(mine, dst=0, op=dist.ReduceOp.SUM)                          # slices are disjoint: the sum is the merge
    if rank == 0:
        full = o.batch_align(m, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 128), 50, ("x_drop",))
        ok = (np.array_equal(mine[:, 0].numpy(), full["scores"]) and np.array_equal(mine[:, 1].numpy(), full["query_idx"])
              and np.array_equal(mine[:, 2].numpy(), full["reference_idx"]))
        q.put(("merged", ok, b))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_merges_in_caller_order_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slice_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tag, ok, bounds = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
    assert tag == "merged" and ok and bounds[0] == 0 and bounds[-1] == 64 and 0 < bounds[1] < 64


def _ragged_worker(rank, world, port, q):
    """World 4, ragged slices of one global list (one pair outweighs many: a slice may be empty), traceback included: every rank
    aligns its slice (the oracle stands in for the device on CPU), rank 0 gathers scores and CIGAR strings in caller order."""
    import numpy as np
    import torch.distributed as dist
    from block_aligner_amd import scores as S
    from oracle.oracle_py import Oracle, cigar_runs_to_string
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    lists = []
    for n in [6000] + [int(x) for x in rng.integers(0, 120, 37)] + [0, 0, 2500]:
        a = synth.rand_str(rng, n, synth.DNA)
        lists.append((synth.mutate(rng, a, n // 10, synth.DNA).tobytes() if n else b"", a.tobytes()))
    pairs = synth.PairSet.from_lists(lists)                                   # the same global list on every rank
    sub, lo, hi = shard.job_slice(pairs, rank, world)
    o = Oracle("avx2")
    m = S.NucMatrix.new_simple(2, -3)
    mine = []
    if sub is not None:
        ref = o.batch_align(m, sub.pool, sub.q_off, sub.q_len, sub.r_off, sub.r_len, (-5, -1), (32, 256), 60, ("trace", "x_drop"), cigar_eq=True)
        for k in range(hi - lo):
            runs = ref["cig_ops"][int(ref["cig_off"][k]): int(ref["cig_off"][k]) + int(ref["cig_len"][k])]
            mine.append((lo + k, int(ref["scores"][k]), cigar_runs_to_string(runs)))
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, mine))
    if rank == 0:
        full = o.batch_align(m, pairs.pool, pairs.q_off, pairs.q_len, pairs.r_off, pairs.r_len, (-5, -1), (32, 256), 60, ("trace", "x_drop"), cigar_eq=True)
        merged = sorted(x for _, _, part in gathered for x in part)
        ok = [i for i, _, _ in merged] == list(range(len(pairs)))
        for i, sc, cg in merged:
            runs = full["cig_ops"][int(full["cig_off"][i]): int(full["cig_off"][i]) + int(full["cig_len"][i])]
            ok = ok and sc == int(full["scores"][i]) and cg == cigar_runs_to_string(runs)
        spans = [(a, b) for a, b, _ in gathered]
        q.put(("merged", ok, spans))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_slices_gloo_world4():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    tag, ok, spans = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert tag == "merged" and ok
    assert spans[0][0] == 0 and spans[-1][1] == 41 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [b - a for a, b in spans]
    assert min(sizes) < max(sizes)           # ragged: the 6000-base pair fills a slice almost alone
