"""Sample-batch sharding across the GPUs of one node.

SinDDM's sampler has no cross-sample operation (no normalisation layers), so the
only parallelism that makes sense for a single small image is independent diffusion chains: one
process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI), rank r runs its own
`local_batch` chains through ALL scales with zero communication, and results are collected with one
all-gather per scale output (10-20 MB per rank: < 1 ms on 7x153 GB/s xGMI links).
Works unchanged on CPU with the gloo backend (used by the world_size=2 tests).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as td


def is_dist() -> bool:
    return td.is_available() and td.is_initialized()


def rank() -> int:
    return td.get_rank() if is_dist() else 0


def world_size() -> int:
    return td.get_world_size() if is_dist() else 1


def shard_sizes(global_batch: int, world: int) -> List[int]:
    """Chains per rank: as even as possible, the first (global_batch % world) ranks get one more."""
    base, rem = divmod(int(global_batch), int(world))
    return [base + (1 if r < rem else 0) for r in range(world)]


def local_batch(global_batch: int) -> int:
    return shard_sizes(global_batch, world_size())[rank()]


def gather_batch(local: torch.Tensor, global_batch: int) -> torch.Tensor:
    """All-gather the per-rank sample shards along dim 0 (uneven shards are padded to the largest
    shard for the collective and trimmed afterwards).  Identity on a single process."""
    if not is_dist() or world_size() == 1:
        return local
    sizes = shard_sizes(global_batch, world_size())
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    pad = pad.contiguous()
    out = torch.empty((world_size() * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    td.all_gather_into_tensor(out, pad)
    parts = [out[r * mx: r * mx + sizes[r]] for r in range(world_size())]
    return torch.cat(parts, dim=0)


def seed_for_rank(base_seed: int) -> int:
    """Every rank draws its own noise stream (seed + rank)."""
    return int(base_seed) + rank()
