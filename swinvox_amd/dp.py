"""Sample-sharded data parallelism: one process per GPU, one bucketed gradient all-reduce per step over RCCL/xGMI.

Replaces the reference's single-process torch.nn.DataParallel (core/train.py:156-161): the batch dimension B is
sharded across ranks (views of one sample stay together: cross-view attention and the merger couple them), BatchNorm
statistics stay per rank (as under DataParallel), and the only collective is the gradient all-reduce (sum / world).
Each module's gradients become ready together (one autograd node per module), so its bucket is launched from a
post-accumulate hook while the backward of the earlier modules is still running (refiner -> merger -> decoder -> encoder).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

from .optim import flat_region


def sync_module_states(modules: Sequence[torch.nn.Module], group=None, src: int = 0) -> int:
    """Every parameter and buffer (BatchNorm running statistics, num_batches_tracked) of `modules` := rank `src`'s, in
    coalesced broadcasts - what DataParallel's replicate() does every step (core/train.py:156-161) and DDP does once at
    construction.  Without it ranks that seeded or loaded differently would average gradients of different models.
    In-place copies: parameters that are views of a flat solver buffer stay views.  Returns the number of tensors synced."""
    tensors = []
    for m in modules:
        tensors += [p.data for p in m.parameters()] + [b.data for b in m.buffers()]
    if tensors and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        pg = group if group is not None else dist.group.WORLD
        dist._broadcast_coalesced(pg, tensors, 256 << 20, src)
    return len(tensors)


class GradAllReducer:
    def __init__(self, modules: Sequence[torch.nn.Module], bucket_bytes: int = 64 << 20, group=None, sync_states: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if sync_states and self.world > 1:
            sync_module_states(modules, group)
        self.buckets: List[List[torch.nn.Parameter]] = []
        self._pending = []
        self._countdown = {}
        self._bucket_of = {}
        for m in modules:
            cur, size = [], 0
            for p in m.parameters():
                if not p.requires_grad:
                    continue
                cur.append(p)
                size += p.numel() * p.element_size()
                if size >= bucket_bytes:
                    self.buckets.append(cur)
                    cur, size = [], 0
            if cur:
                self.buckets.append(cur)
        self._handles = []
        if self.world > 1:
            for bi, b in enumerate(self.buckets):
                for p in b:
                    self._bucket_of[p] = bi
                    self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset()

    def _reset(self):
        self._countdown = {bi: len(b) for bi, b in enumerate(self.buckets)}
        self._pending = []

    def _launch(self, bi: int):
        grads = [p.grad for p in self.buckets[bi]]
        flat = flat_region(grads)          # a HipModule's gradients are views of one buffer: reduce it in place
        if flat is not None:
            grads = None
        else:
            flat = _flatten_dense_tensors(grads)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, grads))

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._countdown[bi] -= 1
        if self._countdown[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Wait for the in-flight buckets, write the averaged gradients back.  Call once after backward()."""
        if self.world == 1:
            return
        for bi, left in self._countdown.items():   # parameters that received no gradient this step
            if left > 0 and all(p.grad is not None for p in self.buckets[bi]):
                self._launch(bi)
        for work, flat, grads in self._pending:
            work.wait()
            flat.div_(self.world)
            if grads is not None:
                for g, r in zip(grads, _unflatten_dense_tensors(flat, grads)):
                    g.copy_(r)
        self._reset()

    def remove(self):
        for h in self._handles:
            h.remove()


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of dim 0 (samples); shards must be equal-sized (reference uses drop_last=True)."""
    assert t.shape[0] % world == 0, "global batch must divide evenly over ranks"
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n]
