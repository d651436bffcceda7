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
