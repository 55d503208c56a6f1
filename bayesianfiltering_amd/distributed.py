"""Multi-GPU: shard the trajectory batch, gather posterior summaries.

Trajectories are independent, so the batch axis is partitioned into contiguous blocks -- one
process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` on
CPU for tests), no communication during the scan -- and the per-trajectory posterior summaries
are exchanged once with an all-gather.  The reference has no distributed code;
this is the engine's own scaling layer.
"""
from typing import Callable, Tuple


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the batch owned by ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_summaries(local, batch: int, force: bool = False):
    """All-gather per-trajectory summaries.  ``local``: (b_local, ...) tensor holding this rank's
    block of a (batch, ...) array sharded by :func:`shard_bounds`.  Returns the full (batch, ...)
    tensor on every rank.  Ragged shards are padded to the largest block for the collective.
    ``force``: issue the collective even in a one-rank group (the same RCCL call on one GPU)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    bmax = (batch + world - 1) // world
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != bmax:
        send = torch.zeros((bmax,) + tail, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * bmax,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous())
    if batch == world * bmax:
        return recv
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(batch, r, world)
        parts.append(recv[r * bmax: r * bmax + (hi - lo)])
    return torch.cat(parts, dim=0)


def filter_sharded(local_filter: Callable, emissions, *args, summary: Callable, **kwargs):
    """Run ``local_filter(emissions[lo:hi], *args, **kwargs)`` on this rank's block of the batch and
    all-gather ``summary(result)`` (a (b_local, ...) tensor).  Returns (local_result, gathered)."""
    import torch.distributed as dist
    batch = emissions.shape[0]
    if dist.is_initialized():
        lo, hi = shard_bounds(batch, dist.get_rank(), dist.get_world_size())
    else:
        lo, hi = 0, batch
    result = local_filter(emissions[lo:hi], *args, **kwargs)
    return result, all_gather_summaries(summary(result), batch)
