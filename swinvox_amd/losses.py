"""torch.nn.BCEWithLogitsLoss() (mean reduction) of the reference's training / test loops (core/train.py:165,249,255;
core/test.py:97,128,132) on the HIP kernel `sv_bce_logits`: one streaming pass for the loss, one for the gradient
(sigmoid(x) - t) * grad_out / n with the upstream scalar read on the device (no host synchronisation)."""
from __future__ import annotations

import torch

from . import hip
from .hip import call, ptr


class _BceLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        hip.check_cuda(x, t)
        if x.shape != t.shape:
            raise ValueError(f"Target size ({tuple(t.shape)}) must be the same as input size ({tuple(x.shape)})")   # torch's wording
        if x.dtype != torch.float32 or t.dtype != torch.float32:
            raise RuntimeError("swinvox_amd: bce_with_logits expects float32 logits and targets")
        x, t = x.contiguous(), t.contiguous()
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        call("sv_bce_logits", ptr(x), ptr(t), x.numel(), ptr(loss), None, None)
        ctx.save_for_backward(x, t)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        x, t = ctx.saved_tensors
        dx = torch.empty_like(x)
        g = dloss.reshape(1).to(torch.float32).contiguous()
        call("sv_bce_logits", ptr(x), ptr(t), x.numel(), None, ptr(dx), ptr(g))
        return dx, None


def bce_with_logits(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean over all elements of max(x, 0) - x t + log(1 + exp(-|x|)); differentiable wrt `logits`."""
    return _BceLogits.apply(logits, target)


class BCEWithLogitsLoss(torch.nn.Module):
    """Drop-in for the `torch.nn.BCEWithLogitsLoss()` instance the reference builds at core/train.py:165."""

    def forward(self, logits, target):
        return bce_with_logits(logits, target)
