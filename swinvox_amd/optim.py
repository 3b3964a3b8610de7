"""Flat-buffer solvers: the reference's per-module clip_grad_norm_(1.0) + torch.optim.Adam / SGD step
(core/train.py:98-131 construction, :279-292 clip and step) as two streaming launches per module.

A module's parameters are re-pointed at views of ONE contiguous fp32 buffer (FlatLayout: registration order, every
parameter on a 64-byte boundary); the backward of a HipModule already returns its gradients as views of a buffer with the
same layout (models/_base.py GradStore), so the step reads the gradient buffer in place: sv_grad_sumsq (the norm the
reference takes over ~340 tensors) and sv_adam_step / sv_sgd_step, which apply the clip coefficient on the device - no
host synchronisation, 32 bytes of HBM traffic per parameter instead of the ~20 passes of the stock foreach path.
Parameter identity, state_dict keys and in-place load_state_dict are unchanged (checkpoint compatibility, core/train.py:358-369).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import hip
from .hip import call, ptr

ALIGN = 16   # floats: every parameter starts on a 64-byte boundary of the flat buffers


class FlatLayout:
    """Offsets of a parameter list inside a flat fp32 buffer + the template that cuts such a buffer into
    parameter-shaped views with one C++ call (padding gaps are cut as throw-away 1-D views)."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.shapes = [tuple(p.shape) for p in params]
        self.offsets: List[int] = []
        self._template, self._keep = [], []
        total = 0
        for s in self.shapes:
            n = 1
            for d in s:
                n *= d
            self.offsets.append(total)
            self._keep.append(len(self._template))
            self._template.append(torch.empty(s, device="meta"))
            total += n
            pad = -total % ALIGN
            if pad:
                self._template.append(torch.empty(pad, device="meta"))
                total += pad
        self.total = max(total, ALIGN)

    def matches(self, params: Sequence[torch.Tensor]) -> bool:
        return len(params) == len(self.shapes) and all(tuple(p.shape) == s for p, s in zip(params, self.shapes))

    def views(self, flat: torch.Tensor) -> List[torch.Tensor]:
        out = torch._utils._unflatten_dense_tensors(flat, self._template)
        return [out[i] for i in self._keep]

    def locate(self, tensors: Sequence[Optional[torch.Tensor]]) -> Optional[int]:
        """Address of the flat buffer the tensors are the views of (same layout, in place), or None."""
        t0 = tensors[0]
        if t0 is None:
            return None
        base = t0.data_ptr()
        for t, off in zip(tensors, self.offsets):
            if t is None or t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != base + 4 * off:
                return None
        st = t0.untyped_storage()
        if base + 4 * self.total > st.data_ptr() + st.nbytes() or base % 16:
            return None
        return base


def flat_region(tensors: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """The slice of the common 1-D base buffer that consecutive layout views cover (gaps = zero padding), or None."""
    b = getattr(tensors[0], "_base", None)
    if b is None or b.dim() != 1 or not b.is_contiguous():
        return None
    end = None
    for t in tensors:
        if t._base is not b or not t.is_contiguous():
            return None
        a = t.data_ptr()
        if end is not None and not (end <= a < end + 4 * ALIGN):
            return None
        end = a + t.numel() * t.element_size()
    lo = (tensors[0].data_ptr() - b.data_ptr()) // b.element_size()
    hi = (end - b.data_ptr()) // b.element_size()
    return b[lo:hi]


class _FlatSolver(torch.optim.Optimizer):
    """One parameter group, one flat buffer.  step(clip_norm=..., grad_scale=...) fuses the reference's
    clip_grad_norm_ (and the 1/world of a data-parallel mean) into the update."""

    def __init__(self, params, defaults):
        params = [p for p in params if p.requires_grad]     # the reference filters the encoder this way (core/train.py:99)
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("flat solvers take one parameter group (the reference builds one solver per module)")
        self._ps: List[torch.nn.Parameter] = self.param_groups[0]["params"]
        hip.check_cuda(*self._ps)
        for p in self._ps:
            if p.dtype != torch.float32:
                raise RuntimeError("flat solvers expect float32 parameters")
        self.layout = FlatLayout(self._ps)
        dev = self._ps[0].device
        self.flat_p = torch.zeros(self.layout.total, device=dev)
        self._gflat = None
        self.norm_slots = torch.zeros(16, dtype=torch.float64, device=dev)
        self.skipped = torch.zeros(1, dtype=torch.int64, device=dev)   # steps the device skipped (non-finite gradient)
        self.steps = 0
        self._adopt()

    def _adopt(self):
        """Copy the parameters into the flat buffer and re-point them at its views (again after module.to()/.cuda())."""
        views = self.layout.views(self.flat_p)
        with torch.no_grad():
            torch._foreach_copy_(views, [p.data for p in self._ps])
            for p, v in zip(self._ps, views):
                p.data = v

    def _grad_address(self) -> Optional[int]:
        grads = [p.grad for p in self._ps]
        n_none = sum(g is None for g in grads)
        if n_none == len(grads):
            return None
        if n_none:
            raise RuntimeError("flat solver: some parameters of the module have a gradient and some do not; a HipModule returns "
                               "all of them from one backward - use torch.optim for partially frozen modules")
        addr = self.layout.locate(grads)
        if addr is not None:
            return addr
        if self._gflat is None:
            self._gflat = torch.zeros(self.layout.total, device=self.flat_p.device)
            self._gviews = self.layout.views(self._gflat)
        torch._foreach_copy_(self._gviews, grads)
        return self._gflat.data_ptr()

    def _prepare(self, clip_norm, grad_scale):
        if self.layout.locate([p.data for p in self._ps]) != self.flat_p.data_ptr():
            self._adopt()
        g = self._grad_address()
        if g is None:
            return None, None
        # the squared norm is always taken: it carries the clip coefficient AND the inf / NaN check that skips the step
        self.norm_slots.zero_()
        call("sv_grad_sumsq", g, self.layout.total, float(grad_scale), ptr(self.norm_slots))
        return g, ptr(self.norm_slots)

    def grad_norm(self) -> torch.Tensor:
        """L2 norm of the (scaled) gradient measured by the last step (device tensor, no synchronisation)."""
        return self.norm_slots.sum().sqrt().float()

    def skipped_steps(self) -> int:
        """Number of steps skipped because the gradient held an inf / NaN (host read: synchronises)."""
        return int(self.skipped.item())

    def state_dict(self):
        g = self.param_groups[0]
        return {"steps": self.steps, "skipped": self.skipped.clone(), "buffers": {k: v.clone() for k, v in self._buffers().items()},
                "param_group": {k: v for k, v in g.items() if k != "params"}}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        if "skipped" in sd:
            self.skipped.copy_(sd["skipped"])
        for k, v in self._buffers().items():
            v.copy_(sd["buffers"][k])
        self.param_groups[0].update(sd["param_group"])


class FlatAdam(_FlatSolver):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics (coupled L2 decay, no amsgrad)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)

    def _buffers(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    @torch.no_grad()
    def step(self, closure=None, *, clip_norm: Optional[float] = None, grad_scale: float = 1.0):
        if closure is not None:
            raise RuntimeError("flat solvers do not take a closure")
        g, slots = self._prepare(clip_norm, grad_scale)
        if g is None:
            return None
        self.steps += 1
        h = self.param_groups[0]
        call("sv_adam_step", ptr(self.flat_p), g, ptr(self.exp_avg), ptr(self.exp_avg_sq), self.layout.total, float(h["lr"]),
             float(h["betas"][0]), float(h["betas"][1]), float(h["eps"]), float(h["weight_decay"]), self.steps, float(grad_scale),
             slots, float(clip_norm or 0.0), ptr(self.skipped))
        return None


class FlatSGD(_FlatSolver):
    """torch.optim.SGD(params, lr, momentum, weight_decay) semantics (dampening 0, no nesterov)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.momentum_buffer = torch.zeros_like(self.flat_p)

    def _buffers(self):
        return {"momentum_buffer": self.momentum_buffer}

    @torch.no_grad()
    def step(self, closure=None, *, clip_norm: Optional[float] = None, grad_scale: float = 1.0):
        if closure is not None:
            raise RuntimeError("flat solvers do not take a closure")
        g, slots = self._prepare(clip_norm, grad_scale)
        if g is None:
            return None
        self.steps += 1
        h = self.param_groups[0]
        call("sv_sgd_step", ptr(self.flat_p), g, ptr(self.momentum_buffer), self.layout.total, float(h["lr"]), float(h["momentum"]),
             float(h["weight_decay"]), self.steps, float(grad_scale), slots, float(clip_norm or 0.0), ptr(self.skipped))
        return None
