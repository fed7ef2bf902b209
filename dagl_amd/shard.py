"""Multi-GPU sharding of the block: plain image-batch data parallel, one process per GPU.

Samples are independent in the reference (``for ... in zip`` over the batch,
DN_Gray/model/dagl.py:245) and its inference driver batches independent tiles
(``forward_chop``, DN_Gray/model/__init__.py:195-214), so the forward path
shards with NO exchange step: rank r owns a contiguous slice of the images /
tiles.  RCCL (``backend="nccl"``) is only used for barriers, the max-over-ranks
timing reduction and, optionally, gathering the outputs.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` owned by ``rank``; sizes differ by at most one, earlier ranks
    take the remainder."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """Distinct synthetic-input seed per rank (weak scaling: every rank has its own images)."""
    return seed * 1000 + rank


def reduce_max_seconds(seconds: float, dist=None, device: Optional[torch.device] = None) -> float:
    """MAX over ranks of a wall-clock duration (all_reduce on the job's backend; identity without a group)."""
    if dist is None or not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def forward_sharded(module, x_all: torch.Tensor, dist=None, gather: bool = True):
    """Run ``module`` on this rank's slice of the batch ``x_all`` ([B,...], identical on every rank) and, if
    ``gather``, all_gather the slices back into a full ``[B,...]`` output on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return module(x_all)
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(x_all.shape[0], rank, world)
    mine = module(x_all[lo:hi].contiguous()) if hi > lo else None
    if not gather:
        return mine
    # slices may differ by one image: gather fixed-size padded slices, then trim
    per = -(-x_all.shape[0] // world)
    probe = mine if mine is not None else module(x_all[:1].contiguous())
    buf = torch.zeros((per,) + tuple(probe.shape[1:]), dtype=probe.dtype, device=probe.device)
    if mine is not None:
        buf[: hi - lo] = mine
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = []
    for r in range(world):
        l, h = shard_range(x_all.shape[0], r, world)
        parts.append(outs[r][: h - l])
    return torch.cat(parts, dim=0)
