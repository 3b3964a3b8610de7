"""Thin counterparts of the reference's hot loops around the four modules:

  train_step(...)  = core/train.py:222-297 for one batch: clamp inputs, forward with the USE_MERGER / USE_REFINER /
                     EPOCH_START_USE_* gating, encoder + refiner BCE losses, zero_grad, ONE backward, per-module
                     clip_grad_norm_(1.0), optimizer steps in the reference order (no GradScaler: bf16/fp32 MFMA math
                     needs no loss scaling).  With a GradAllReducer the gradients are averaged over ranks before clipping.
  evaluate(...)    = core/test.py:114-153 for a batch: fp32-style eval forward, losses x10, sigmoid-threshold IoU at
                     cfg.TEST.VOXEL_THRESH computed on the device (sv_iou_counts), one host sync for the whole batch.
  make_solvers(...) = core/train.py:98-152 (Adam with the reference betas / weight decay / per-module learning rates,
                     or SGD), MultiStepLR schedulers; by default the flat-buffer solvers of optim.py, which fuse the
                     per-module clip_grad_norm_(1.0) into the update (two launches per module).
  checkpoint_dict / load_checkpoint = core/train.py:347-369 / :163-190 (same keys, `module.` prefixes accepted).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import hip
from .hip import call, ptr
from .losses import bce_with_logits as _bce      # torch.nn.BCEWithLogitsLoss() of core/train.py:165 on sv_bce_logits
from .optim import FlatAdam, FlatSGD, _FlatSolver


def make_solvers(nets, cfg, fused: bool = True):
    """fused=True: optim.FlatAdam / FlatSGD (clip + step in two launches per module); False: the stock torch optimizers."""
    enc, dec, mer, ref = nets
    t = cfg.TRAIN
    lrs = (t.ENCODER_LEARNING_RATE, t.DECODER_LEARNING_RATE, t.MERGER_LEARNING_RATE, t.REFINER_LEARNING_RATE)
    mods = (enc, dec, mer, ref)
    if t.POLICY == "adam":
        mk = (lambda ps, lr: FlatAdam(ps, lr=lr, betas=tuple(t.BETAS), weight_decay=t.WEIGHT_DECAY)) if fused else \
             (lambda ps, lr: torch.optim.Adam(ps, lr=lr, betas=tuple(t.BETAS), weight_decay=t.WEIGHT_DECAY))
    elif t.POLICY == "sgd":
        mk = (lambda ps, lr: FlatSGD(ps, lr=lr, momentum=t.MOMENTUM, weight_decay=t.WEIGHT_DECAY)) if fused else \
             (lambda ps, lr: torch.optim.SGD(ps, lr=lr, momentum=t.MOMENTUM, weight_decay=t.WEIGHT_DECAY))
    else:
        raise Exception("[FATAL] Unknown optimizer %s." % t.POLICY)   # reference core/train.py:133
    opts = [mk([p for p in m.parameters() if p.requires_grad], lr) for m, lr in zip(mods, lrs)]
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
    enc_s, dec_s, mer_s, ref_s = solvers
    active = [(nets[0], enc_s, True), (nets[1], dec_s, True), (nets[3], ref_s, use_refiner), (nets[2], mer_s, use_merger)]
    for n, sol, _ in active:                         # clip every module (core/train.py:279-282) ...
        if not isinstance(sol, _FlatSolver):
            ps = [p for p in n.parameters() if p.grad is not None]
            if ps:
                torch.nn.utils.clip_grad_norm_(ps, max_norm=1.0)
    for n, sol, on in active:                        # ... then step in the reference order (:287-292)
        if on:
            if isinstance(sol, _FlatSolver):
                sol.step(clip_norm=1.0)              # norm + clip + update fused on the flat buffers
            else:
                sol.step()
    return el.detach(), rl.detach()


@torch.no_grad()
def voxel_metrics(volume, gt, thresholds):
    """IoU and F-score of sigmoid(volume) >= th against gt for every sample and threshold, as core/test.py:141-163 defines
    them (IoU 1.0 when prediction and ground truth are both empty; the 1e-8 epsilons of the F-score), from one kernel pass
    over [B, 32^3] and no host synchronisation.  Returns (iou [B, n_th], fscore [B, n_th]) device tensors."""
    hip.check_cuda(volume, gt)
    B = volume.shape[0]
    S_ = volume[0].numel()
    ths = torch.tensor(list(thresholds), dtype=torch.float32, device=volume.device)
    counts = torch.empty(B, len(ths), 4, device=volume.device)
    call("sv_iou_counts", ptr(volume.contiguous()), ptr(gt.contiguous()), ptr(ths), len(ths), B, S_, ptr(counts))
    tp, union, fp, fn = counts.unbind(-1)
    iou = torch.where(union > 0, tp / union.clamp_min(1), torch.ones_like(tp))
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    return iou, 2 * precision * recall / (precision + recall + 1e-8)


@torch.no_grad()
def evaluate(nets, cfg, images, gt, epoch_idx: int = 0, with_fscore: bool = False):
    """Returns (encoder_loss*10, refiner_loss*10, iou[B, n_thresholds]) (+ fscore[B, n_thresholds] with_fscore=True) for a
    batch, per sample and threshold as in core/test.py:120-163.  The nets are switched to eval() for the pass (core/test.py:100-104:
    running BatchNorm statistics, no dropout / drop-path, no statistics update) and put back into their previous mode."""
    was_training = [n.training for n in nets]
    for n in nets:
        n.eval()
    try:
        total, el, rl, volume, _ = forward_losses(nets, cfg, images, gt, epoch_idx)
    finally:
        for n, t in zip(nets, was_training):
            n.train(t)
    iou, fs = voxel_metrics(volume, gt, cfg.TEST.VOXEL_THRESH)
    return (el * 10, rl * 10, iou, fs) if with_fscore else (el * 10, rl * 10, iou)


def aggregate_by_taxonomy(taxonomy_ids: Sequence, per_sample):
    """core/test.py:187-203: per-taxonomy mean of per-sample metric rows and the sample-weighted overall mean.
    `per_sample` is [N, n_thresholds] (tensor or nested list).  Returns ({taxonomy_id: (n_samples, mean_row)}, overall_row)."""
    rows = torch.as_tensor(per_sample, dtype=torch.float64).cpu()
    assert rows.shape[0] == len(taxonomy_ids)
    out, order = {}, []
    for i, t in enumerate(taxonomy_ids):
        if t not in out:
            out[t] = []
            order.append(t)
        out[t].append(rows[i])
    table = {t: (len(out[t]), torch.stack(out[t]).mean(0)) for t in order}
    overall = sum(n * m for n, m in table.values()) / len(taxonomy_ids)
    return {t: (n, m.tolist()) for t, (n, m) in table.items()}, overall.tolist()

_SCALER_STATE = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}


def checkpoint_dict(nets, cfg, epoch_idx: int, best_iou: float, best_epoch: int, module_prefix: bool = True):
    """The dict core/train.py:347-369 saves.  The reference checkpoints DataParallel-wrapped modules on a GPU machine, so its
    keys carry a `module.` prefix; module_prefix=True writes the same keys so that core/test.py:81-90 and the resume path
    core/train.py:171-186 load the file unchanged.  Tensors are cloned (parameters may be views of a solver's flat buffer).
    `scaler_state_dict` holds the state of an untouched GradScaler(init_scale=2**16): there is no loss scaling here."""
    enc, dec, mer, ref = nets
    pre = "module." if module_prefix else ""

    def sd(m):
        return {pre + k: v.detach().clone() for k, v in m.state_dict().items()}

    ck = {"epoch_idx": int(epoch_idx), "best_iou": float(best_iou), "best_epoch": int(best_epoch),
          "encoder_state_dict": sd(enc), "decoder_state_dict": sd(dec), "scaler_state_dict": dict(_SCALER_STATE)}
    if cfg.NETWORK.USE_REFINER:
        ck["refiner_state_dict"] = sd(ref)
    if cfg.NETWORK.USE_MERGER:
        ck["merger_state_dict"] = sd(mer)
    return ck


def load_checkpoint(nets, cfg, checkpoint):
    """Counterpart of core/train.py:171-186 / core/test.py:81-90: `checkpoint` is a path (read with weights_only=True) or an
    already loaded dict; keys with or without the DataParallel `module.` prefix are accepted.  Parameters are copied in place
    (flat-solver views stay valid).  Returns (epoch_idx, best_iou, best_epoch)."""
    if not isinstance(checkpoint, dict):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
    enc, dec, mer, ref = nets

    def strip(sd):
        return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}

    enc.load_state_dict(strip(checkpoint["encoder_state_dict"]))
    dec.load_state_dict(strip(checkpoint["decoder_state_dict"]))
    if cfg.NETWORK.USE_REFINER:
        ref.load_state_dict(strip(checkpoint["refiner_state_dict"]))
    if cfg.NETWORK.USE_MERGER:
        mer.load_state_dict(strip(checkpoint["merger_state_dict"]))
    return checkpoint["epoch_idx"], checkpoint.get("best_iou", -1), checkpoint.get("best_epoch", -1)
