"""Sample-batch sharding across the GPUs of one node.

SinDDM's sampler has no cross-sample operation (no normalisation layers), so the
only parallelism that makes sense for a single small image is independent diffusion chains: one
process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI), rank r runs its own
`local_batch` chains through ALL scales with zero communication, and results are collected with one
all-gather per scale output (10-20 MB per rank: < 1 ms on 7x153 GB/s xGMI links).
Works unchanged on CPU with the gloo backend (used by the world_size=2 tests).

Data-parallel training (an extension of reference trainer.py:189-224): the training batch
is sharded the same way, every rank back-propagates `(b_r / B) * mean-loss` of its shard, and ONE all-reduce of
the flat 4.4 MB gradient buffer per optimizer step (RCCL ring over xGMI: ~8 MB moved per GPU) gives every rank
the global-batch gradient before the fused Adam/EMA kernel; parameters stay bit-identical across ranks because
they all apply the same reduced gradient.
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


def _skip(force: bool) -> bool:
    """Collectives are identities on a single process -- unless `force` asks for the real call (one-rank process
    group: the RCCL code path of the multi-GPU runs, executable on a 1-GPU box; tests/test_gpu_rccl.py)."""
    return not is_dist() or (world_size() == 1 and not force)


def gather_batch(local: torch.Tensor, global_batch: int, force: bool = False, pad_to: int = 0) -> torch.Tensor:
    """All-gather the per-rank sample shards along dim 0 (uneven shards are padded to the largest
    shard for the collective and trimmed afterwards).  Identity on a single process.  `pad_to` (tests): pad the shards
    to at least this many samples, as ranks with a smaller shard do."""
    if _skip(force):
        return local
    sizes = shard_sizes(global_batch, world_size())
    mx = max(max(sizes), int(pad_to))
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


def shard_offset(global_batch: int) -> int:
    """Global index of this rank's first sample."""
    return sum(shard_sizes(global_batch, world_size())[:rank()])


def allreduce_sum_(t: torch.Tensor, force: bool = False) -> torch.Tensor:
    """In-place SUM all-reduce (identity on a single process).  With the gloo backend device tensors are
    staged through the host, so the 2-process tests can share one GPU; nccl (RCCL) reduces in place."""
    if _skip(force):
        return t
    if t.is_cuda and td.get_backend() == "gloo":
        h = t.detach().cpu()
        td.all_reduce(h, op=td.ReduceOp.SUM)
        t.copy_(h)
    else:
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def broadcast_int(v: int, src: int = 0, force: bool = False) -> int:
    """Agree on a host integer (e.g. the seed of the scale-pick generator)."""
    if _skip(force):
        return int(v)
    obj = [int(v)]
    td.broadcast_object_list(obj, src=src)
    return int(obj[0])


def broadcast_(t: torch.Tensor, src: int = 0, force: bool = False) -> torch.Tensor:
    """In-place broadcast (identity on a single process); host-staged under gloo like allreduce_sum_."""
    if _skip(force):
        return t
    if t.is_cuda and td.get_backend() == "gloo":
        h = t.detach().cpu()
        td.broadcast(h, src=src)
        t.copy_(h)
    else:
        td.broadcast(t, src=src)
    return t
