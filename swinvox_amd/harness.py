"""Thin counterparts of the reference's hot loops around the four modules:

  train_step(...)  = core/train.py:222-297 for one batch: clamp inputs, forward with the USE_MERGER / USE_REFINER /
                     EPOCH_START_USE_* gating, encoder + refiner BCE losses, zero_grad, ONE backward, per-module
                     clip_grad_norm_(1.0), optimizer steps in the reference order (no GradScaler: bf16/fp32 MFMA math
                     needs no loss scaling).  With a GradAllReducer the gradients are averaged over ranks before clipping.
  evaluate(...)    = core/test.py:114-153 for a batch: fp32-style eval forward, losses x10, sigmoid-threshold IoU at
                     cfg.TEST.VOXEL_THRESH computed on the device (sv_iou_counts), one host sync for the whole batch.
  make_solvers(...) = core/train.py:98-152 (Adam with the reference betas / weight decay / per-module learning rates,
                     or SGD), MultiStepLR schedulers.
Everything outside the model forward/backward is stock PyTorch (optimizers, clipping): plumbing, not the hot path.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import hip
from .hip import call, ptr

_bce = torch.nn.functional.binary_cross_entropy_with_logits


def make_solvers(nets, cfg):
    enc, dec, mer, ref = nets
    t = cfg.TRAIN
    lrs = (t.ENCODER_LEARNING_RATE, t.DECODER_LEARNING_RATE, t.MERGER_LEARNING_RATE, t.REFINER_LEARNING_RATE)
    mods = (enc, dec, mer, ref)
    if t.POLICY == "adam":
        opts = [torch.optim.Adam(m.parameters(), lr=lr, betas=tuple(t.BETAS), weight_decay=t.WEIGHT_DECAY) for m, lr in zip(mods, lrs)]
    elif t.POLICY == "sgd":
        opts = [torch.optim.SGD(m.parameters(), lr=lr, momentum=t.MOMENTUM, weight_decay=t.WEIGHT_DECAY) for m, lr in zip(mods, lrs)]
    else:
        raise Exception("[FATAL] Unknown optimizer %s." % t.POLICY)   # reference core/train.py:133
    ms = (t.ENCODER_LR_MILESTONES, t.DECODER_LR_MILESTONES, t.MERGER_LR_MILESTONES, t.REFINER_LR_MILESTONES)
    scheds = [torch.optim.lr_scheduler.MultiStepLR(o, milestones=list(m), gamma=t.GAMMA) for o, m in zip(opts, ms)]
    return opts, scheds


def forward_losses(nets, cfg, images, gt, epoch_idx: int = 0):
    enc, dec, mer, ref = nets
    use_merger = cfg.NETWORK.USE_MERGER and epoch_idx >= cfg.TRAIN.EPOCH_START_USE_MERGER
    use_refiner = cfg.NETWORK.USE_REFINER and epoch_idx >= cfg.TRAIN.EPOCH_START_USE_REFINER
    raw, vol = dec(enc(images))
    if use_merger:
        volume = mer(raw, vol)
    else:   # torch.mean(generated_volumes, dim=1), core/train.py:246
        volume = _MeanViews.apply(vol)
    enc_loss = _bce(volume, gt)
    if use_refiner:
        volume = ref(volume)
        ref_loss = _bce(volume, gt)
        total = enc_loss + ref_loss
    else:
        ref_loss, total = enc_loss, enc_loss
    return total, enc_loss, ref_loss, volume, (use_merger, use_refiner)


class _MeanViews(torch.autograd.Function):
    """mean over the view axis (sv_mean_views) with its (trivial) backward."""

    @staticmethod
    def forward(ctx, vol):
        hip.check_cuda(vol)
        B, V = vol.shape[:2]
        ctx.V = V
        vol = vol.contiguous()
        out = torch.empty(B, 32, 32, 32, device=vol.device)
        call("sv_mean_views", ptr(vol), ptr(out), B, V, 32768)
        return out

    @staticmethod
    def backward(ctx, d):
        return (d[:, None] / ctx.V).expand(-1, ctx.V, -1, -1, -1).contiguous()


def train_step(nets, solvers, cfg, images, gt, epoch_idx: int = 0, reducer=None):
    """One optimisation step; returns (encoder_loss, refiner_loss) as device tensors (no host sync)."""
    images = images.clamp(-1, 1)
    gt = gt.clamp(0, 1)
    total, el, rl, _, (use_merger, use_refiner) = forward_losses(nets, cfg, images, gt, epoch_idx)
    for n in nets:
        n.zero_grad(set_to_none=True)
    total.backward()
    if reducer is not None:
        reducer.finish()
    for n in nets:
        ps = [p for p in n.parameters() if p.grad is not None]
        if ps:
            torch.nn.utils.clip_grad_norm_(ps, max_norm=1.0)
    enc_s, dec_s, mer_s, ref_s = solvers
    enc_s.step()
    dec_s.step()
    if use_refiner:
        ref_s.step()
    if use_merger:
        mer_s.step()
    return el.detach(), rl.detach()


@torch.no_grad()
def evaluate(nets, cfg, images, gt, epoch_idx: int = 0):
    """Returns (encoder_loss*10, refiner_loss*10, iou[B, n_thresholds]) - IoU per sample and threshold as in
    core/test.py:141-153 (1.0 when prediction and ground truth are both empty)."""
    total, el, rl, volume, _ = forward_losses(nets, cfg, images, gt, epoch_idx)
    ths = torch.tensor(list(cfg.TEST.VOXEL_THRESH), dtype=torch.float32, device=volume.device)
    B = volume.shape[0]
    counts = torch.empty(B, len(ths), 2, device=volume.device)
    call("sv_iou_counts", ptr(volume.contiguous()), ptr(gt.contiguous()), ptr(ths), len(ths), B, 32768, ptr(counts))
    inter, union = counts[..., 0], counts[..., 1]
    iou = torch.where(union > 0, inter / union.clamp_min(1), torch.ones_like(inter))
    return el * 10, rl * 10, iou
