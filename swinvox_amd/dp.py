"""Sample-sharded data parallelism: one process per GPU, one bucketed gradient all-reduce per step over RCCL/xGMI.

Replaces the reference's single-process torch.nn.DataParallel (core/train.py:156-161): the batch dimension B is
sharded across ranks (views of one sample stay together: cross-view attention and the merger couple them), BatchNorm
statistics stay per rank (as under DataParallel), parameters and buffers are made rank 0's at construction (DataParallel
replicates them every step), and the only collective of a step is the gradient all-reduce (sum / world).

Two ways to drive it:
  * eager (hooks=True): a module's gradients become ready together (one autograd node per module), so its buckets are launched
    from post-accumulate hooks while the backward of the earlier modules is still running (refiner -> merger -> decoder ->
    encoder).  A HipModule that finishes groups of its parameters early inside its backward (the Encoder: neck + fusion head,
    Swin backbone, ResNet trunk - models/encoder.py) announces each group through `grad_ready_hook`; that group's slice of the
    module's flat gradient buffer is reduced as soon as the announcing stream gets there, beside the rest of the backward.
  * graph replay (hooks=False): the step is one hipGraph (graph.GraphedStep) whose gradients are static buffers;
    `reduce_all()` after the replay launches every bucket, in place.
finish() waits, divides by the world size and returns; `stats()` reports payload bytes, bucket count and the event-timed
exposed (non-overlapped) communication time per step.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

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
    def __init__(self, modules: Sequence[torch.nn.Module], bucket_bytes: int = 64 << 20, group=None, sync_states: bool = True,
                 hooks: bool = True, early_groups: bool = False):
        """early_groups (opt-in): buckets of a module that announces parameter groups from inside its backward start there.  The
        default is the plain schedule - one post-accumulate hook per parameter, a bucket starts when its last gradient has been
        accumulated - which is the one that has run on a GPU (under gloo) and in the CPU tests."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.modules = list(modules)
        if sync_states and self.world > 1:
            sync_module_states(self.modules, group)
        self.buckets: List[List[torch.nn.Parameter]] = []
        self._bucket_of: Dict[torch.nn.Parameter, int] = {}
        self._early_groups: Dict[int, List[List[int]]] = {}     # id(module) -> bucket indices per announced group
        for m in self.modules:
            groups = m.grad_groups() if hooks and early_groups and hasattr(m, "grad_groups") else None
            plists = groups if groups else [[p for p in m.parameters()]]
            early = []
            for pl in plists:
                cur, size, idxs = [], 0, []
                for p in pl:
                    if not p.requires_grad:
                        continue
                    cur.append(p)
                    size += p.numel() * p.element_size()
                    if size >= bucket_bytes:
                        idxs.append(len(self.buckets)); self.buckets.append(cur)
                        cur, size = [], 0
                if cur:
                    idxs.append(len(self.buckets)); self.buckets.append(cur)
                early.append(idxs)
            if groups:
                self._early_groups[id(m)] = early
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
        self.payload_bytes = sum(p.numel() * p.element_size() for b in self.buckets for p in b)
        self._handles = []
        self._early_modules = []
        if self.world > 1 and hooks:
            for p in self._bucket_of:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
            for m in self.modules:
                if id(m) in self._early_groups:
                    m.grad_ready_hook = self._make_early_hook(m)
                    self._early_modules.append(m)
        self._exposed = []            # (event before the first wait, event after the last wait) of each finish()
        self._steps = 0
        self._reset()

    # ---- bookkeeping ---------------------------------------------------------------------------------
    def _reset(self):
        self._countdown = {bi: len(b) for bi, b in enumerate(self.buckets)}
        self._launched = set()
        self._pending = []
        self._early_range = {}        # bucket index -> (first byte, end byte) of the buffer slice an early hook reduced

    def _covered(self, bi: int) -> bool:
        """True when every p.grad of bucket `bi` lies inside the slice its early hook reduced (autograd stole the GradStore view)."""
        rng = self._early_range.get(bi)
        if rng is None:
            return True
        for p in self.buckets[bi]:
            g = p.grad
            if g is None or not (rng[0] <= g.data_ptr() and g.data_ptr() + g.numel() * g.element_size() <= rng[1]):
                return False
        return True

    def _check_early(self, bi: int):
        """An early hook reduced the backward's gradient buffer.  That IS p.grad only when AccumulateGrad takes the returned view
        as the gradient (p.grad was None).  With a gradient already in place (accumulation over several backward passes,
        zero_grad(set_to_none=False)) autograd ADDS the reduced view into a different buffer: the sum of a local gradient and a
        world-summed one, which no later collective can repair - refuse instead of training on it."""
        if not self._covered(bi):
            raise RuntimeError(
                "swinvox_amd.dp: early gradient groups need p.grad to be None before backward() (zero_grad(set_to_none=True), one "
                "backward per step); construct GradAllReducer(early_groups=False) for gradient accumulation")

    def _all_reduce(self, flat, grads):
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat, grads))

    def _launch(self, bi: int):
        if bi in self._launched:
            return
        self._launched.add(bi)
        grads = [p.grad for p in self.buckets[bi]]
        flat = flat_region(grads)          # a HipModule's gradients are views of one buffer: reduce it in place
        if flat is not None:
            grads = None
        else:
            flat = _flatten_dense_tensors(grads)
        self._all_reduce(flat, grads)

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._countdown[bi] -= 1
        if self._countdown[bi] == 0:
            if bi in self._early_range:
                self._check_early(bi)
            self._launch(bi)

    def _make_early_hook(self, module):
        early = self._early_groups[id(module)]

        def hook(group_index: int, views: Sequence[torch.Tensor]):
            """Called INSIDE the module's backward, on the stream that completed the group: `views` are the gradient views
            (GradStore) of the group's parameters in registration order.  Collectives are enqueued behind the current stream."""
            it = iter(views)
            for bi in early[group_index]:
                vs = [next(it) for _ in self.buckets[bi]]
                flat = flat_region(vs)
                if flat is None:         # not views of one buffer: leave the bucket to the post-accumulate hooks
                    continue
                self._launched.add(bi)
                self._early_range[bi] = (flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size())
                self._all_reduce(flat, None)
        return hook

    # ---- step API --------------------------------------------------------------------------------------
    def reduce_all(self):
        """Launch every bucket now (graph-replay mode: gradients are complete on the current stream).  Call finish() next."""
        if self.world == 1:
            return
        for bi, b in enumerate(self.buckets):
            if all(p.grad is not None for p in b):
                self._launch(bi)

    def finish(self):
        """Wait for the in-flight buckets, write the averaged gradients back.  Call once after backward()."""
        if self.world == 1:
            return
        for bi in self._early_range:               # buckets whose post-accumulate hooks did not all fire are checked here
            self._check_early(bi)
        for bi, left in self._countdown.items():   # parameters that received no gradient this step
            if left > 0 and bi not in self._launched and all(p.grad is not None for p in self.buckets[bi]):
                self._launch(bi)
        timed = torch.cuda.is_available() and self._pending and self._pending[0][1].is_cuda and len(self._exposed) < 64
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, flat, grads in self._pending:
            work.wait()
            flat.div_(self.world)
            if grads is not None:
                for g, r in zip(grads, _unflatten_dense_tensors(flat, grads)):
                    g.copy_(r)
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._exposed.append((e0, e1))
        self._steps += 1
        self._reset()

    def stats(self, reset: bool = True) -> dict:
        """payload / bucket count and the exposed communication time: elapsed GPU time between the first wait of finish() and
        the completion of the last bucket's division - by then the backward's own kernels have been enqueued before it on the
        same stream, so this is the part of the all-reduce that did not hide behind compute (synchronises)."""
        ms = None
        if self._exposed:
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)
            if reset:
                self._exposed = []
        return {"world_size": self.world, "payload_bytes": self.payload_bytes, "buckets": len(self.buckets),
                "bucket_bytes": [sum(p.numel() * p.element_size() for p in b) for b in self.buckets],
                "early_groups": sum(len(v) for v in self._early_groups.values()), "exposed_ms_per_step": ms}

    def isolated_allreduce_ms(self, repeats: int = 3) -> Optional[float]:
        """Calibration: all buckets reduced back to back on throw-away buffers with nothing else running (event-timed)."""
        if self.world == 1 or not torch.cuda.is_available():
            return None
        dev = next(iter(self._bucket_of)).device
        if dev.type != "cuda":
            return None
        bufs = [torch.zeros(sum(p.numel() for p in b), device=dev) for b in self.buckets]
        times = []
        for _ in range(repeats + 1):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            works = [dist.all_reduce(t, group=self.group, async_op=True) for t in bufs]
            for w in works:
                w.wait()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        return sum(times[1:]) / repeats

    def remove(self):
        for h in self._handles:
            h.remove()
        for m in self._early_modules:
            m.grad_ready_hook = None


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous shard of dim 0 (samples); shards must be equal-sized (reference uses drop_last=True)."""
    assert t.shape[0] % world == 0, "global batch must divide evenly over ranks"
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n]
