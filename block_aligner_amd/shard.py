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
