"""Rank -> shard arithmetic and the (tiny) distributed reductions the bench needs. Pairs are independent, so there is
no collective on the data path: every rank aligns its own batch; ranks only agree on elapsed time and total cells."""
from __future__ import annotations


def shard_seed(base_seed: int, rank: int) -> int:
    """Distinct, reproducible input stream per rank (weak scaling: every rank generates its own batch)."""
    return base_seed + 100003 * rank


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous cost-balanced slice [lo, hi) of a global batch of equal-cost pairs (strong-scaling helper)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_slices(q_len, r_len, parts: int):
    """Boundaries of `parts` contiguous slices of near-equal summed cost |q| + |r| + 16 -- the rule of the library's
    ba_shard_slices (ba_host.cpp), restated in numpy so a launcher can cut a global pair list the same way per rank."""
    import numpy as np
    cost = np.asarray(q_len, dtype=np.uint64) + np.asarray(r_len, dtype=np.uint64) + np.uint64(16)
    pre = np.concatenate([[0], np.cumsum(cost)]).astype(np.uint64)
    n, total = len(cost), int(pre[-1])
    bounds = [0]
    for k in range(1, parts):
        target = total // parts * k + total % parts * k // parts
        lo = int(np.searchsorted(pre, np.uint64(target), side="left"))
        bounds.append(max(min(lo, n), bounds[-1]))
    bounds.append(n)
    return bounds


def job_slice(pairs, rank: int, world: int):
    """Strong scaling: rank's cost-balanced contiguous slice of ONE global pair list -> (PairSet of the slice, lo, hi). An empty
    slice (more ranks than pairs, or one pair that outweighs the rest) is returned as None."""
    import numpy as np
    b = balanced_slices(pairs.q_len, pairs.r_len, world)
    lo, hi = b[rank], b[rank + 1]
    return (pairs.subset(np.arange(lo, hi)) if hi > lo else None), lo, hi


def reduce_job(elapsed_s: float, cells: float, device=None):
    """-> (max elapsed over ranks, total cells over ranks). Works on any initialised backend (nccl on GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    el = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    tot = torch.tensor([cells], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(el.item()), float(tot.item())
