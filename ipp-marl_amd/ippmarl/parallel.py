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
    """Gradient averaging across ranks without staging copies.

    ``attach(*modules)`` lays the gradients of all the modules' trainable parameters out in ONE persistent flat buffer and
    makes every ``p.grad`` a view into it (``fc2`` of the convnets is skipped: the reference keeps that layer but never
    uses it, so it never has a gradient).  ``reducer(module, ...)`` then all-reduces the span of the buffer that holds the
    named modules in a single call -- no ``torch.cat``, no copy-back; modules attached next to each other (the trainer
    attaches critic then actor) go out together.  Optimizers must clear gradients with ``zero_grad(set_to_none=False)``
    (``GradAllReducer.zero``) so that the views survive.
    """

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.bytes_reduced = 0
        self.calls = 0
        self.flat: Optional[torch.Tensor] = None
        self.spans = {}   # id(module) -> (lo, hi) in elements
        self.skip = self.unused_fc2
        self.modules: List[torch.nn.Module] = []

    @staticmethod
    def unused_fc2(name: str) -> bool:
        """The default ``skip`` of attach(): ``fc2`` of the reference's convnets exists but is never used in forward."""
        return name.startswith("fc2")

    def _trainable(self, module: torch.nn.Module):
        return [p for n, p in module.named_parameters() if p.requires_grad and not self.skip(n)]

    def attach(self, *modules: torch.nn.Module, skip=None):
        """``skip(name) -> bool``: parameters that never receive a gradient and stay out of the buffer (default: the unused
        ``fc2``).  A parameter that is skipped here but does get a gradient would silently never be averaged: check_covered()
        (called by the trainer after its first backward pass) catches that."""
        self.skip = skip or self.unused_fc2
        self.modules = list(modules)
        params = [p for m in modules for p in self._trainable(m)]
        assert params, "attach: nothing to reduce"
        assert all(p.dtype == params[0].dtype and p.device == params[0].device for p in params), \
            "attach: all gradients share one flat buffer, so the parameters must share dtype and device"
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for m in modules:
            lo = off
            for p in self._trainable(m):
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.spans[id(m)] = (lo, off)
        return self

    def check_covered(self):
        """After a backward pass: every gradient of the attached modules lives in the flat buffer (a skipped parameter that
        received a gradient would diverge across ranks without any error)."""
        lo = self.flat.data_ptr()
        hi = lo + self.flat.numel() * self.flat.element_size()
        for m in self.modules:
            for n, p in m.named_parameters():
                if p.grad is not None and not (lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError(f"GradAllReducer: parameter {n} has a gradient outside the reduced buffer (skipped by "
                                       "attach(skip=...) although it is used)")

    def zero(self, *modules: torch.nn.Module):
        """Clears the gradients of the modules in place (their views into the flat buffer stay)."""
        for m in modules:
            lo, hi = self.spans[id(m)]
            self.flat[lo:hi].zero_()

    def __call__(self, *modules: torch.nn.Module):
        if self.world == 1:
            return
        if self.flat is None or any(id(m) not in self.spans for m in modules):   # unattached modules: staged path
            grads: List[torch.Tensor] = [p.grad for m in modules for p in m.parameters() if p.grad is not None]
            if not grads:
                return
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            self.bytes_reduced += flat.numel() * flat.element_size()
            self.calls += 1
            return
        spans = sorted(self.spans[id(m)] for m in modules)
        merged = [list(spans[0])]
        for lo, hi in spans[1:]:
            if lo == merged[-1][1]:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        for lo, hi in merged:   # adjacent modules: one call
            view = self.flat[lo:hi]
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            view.div_(self.world)
            self.bytes_reduced += view.numel() * view.element_size()
            self.calls += 1


def max_over_ranks(value: float, device, group: Optional[dist.ProcessGroup] = None) -> float:
    """The slowest rank's figure (bench.py times a region per rank and reports the maximum)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


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
