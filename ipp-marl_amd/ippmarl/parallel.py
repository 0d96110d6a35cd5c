"""Data parallelism over independent environments: one process per GPU, torch.distributed (RCCL = backend "nccl" on
ROCm) over xGMI.  Envs share nothing but the network weights, so the only collective on the path is the gradient
average of the two small convnets (actor 2.28 M + critic 2.31 M float32 = 18.3 MB), done as ONE flat bucket per
optimizer step: the transfer is latency-bound on a full xGMI mesh, so fewer, larger messages win.
``fc2`` of both nets never receives a gradient (unused layer kept for checkpoint compatibility) and is skipped.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of n_total items owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def episode_ids(first_episode: int, wave: int, envs_per_rank: int, rank: int, world: int) -> torch.Tensor:
    """Episode numbers of one rollout wave.  Episode numbers seed everything (truth, start cells, Philox streams),
    so a given episode evolves identically whatever rank/batch it lands in."""
    start = first_episode + wave * envs_per_rank * world + rank * envs_per_rank
    return torch.arange(start, start + envs_per_rank, dtype=torch.int64)


class GradAllReducer:
    """Averages the gradients of a module across ranks with one flat all-reduce (call after backward())."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.bytes_reduced = 0

    def __call__(self, *modules: torch.nn.Module):
        if self.world == 1:
            return
        grads: List[torch.Tensor] = [p.grad for m in modules for p in m.parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        self.bytes_reduced += flat.numel() * flat.element_size()


def broadcast_module(module: torch.nn.Module, src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Same initial weights on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def all_reduce_mean_scalar(value: float, device, group: Optional[dist.ProcessGroup] = None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return float(t[0]) / dist.get_world_size(group)
