"""Learning-rate range test on the HIP train step (reference utils/lr_finder.py:19-268).

`lr_range_test` does what the reference's `find_lr` does between its data loader and its plot: four solvers started at START_LR (merger and
refiner at START_LR / 10, lr_finder.py:84-92; Adam with the reference betas and NO weight decay, as there), one training step per batch with the
learning rates multiplied by (END_LR / START_LR) ** (1 / (N - 1)) after every batch (:118,222-231), the loss smoothed exponentially with AVG_BETA
(:206-207), a stop when the smoothed loss exceeds ten times the best loss after the tenth batch (:212-214), the suggested learning rate from the
steepest descent of the smoothed curve (`suggest_lr`, :243-256) and the modules' initial state restored at the end (:270-276).  The batches come from
any iterable of (images, ground-truth volumes); the forward / backward / clip / update of a batch is harness.train_step on the flat solvers, and
USE_MERGER / USE_REFINER act without the EPOCH_START gates (the reference's finder has none).  `next_n_views_rendering` is the per-epoch
random view count of core/train.py:335-337.
"""
from __future__ import annotations

import copy
import random
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import harness
from .config import Cfg
from .optim import FlatAdam, FlatSGD

LR_FINDER_DEFAULTS = dict(START_LR=1e-7, END_LR=1e-2, NUM_BATCHES_TO_TEST=1000, AVG_BETA=0.98)   # reference config.py:138-142


def suggest_lr(lrs: Sequence[float], smoothed: Sequence[float]) -> Optional[float]:
    """lr_finder.py:243-256: in the (at most) 50 points before the minimum of the smoothed loss take the point of steepest descent
    (numpy.gradient of loss over lr) and step five points back.  None when the curve is too short or the minimum comes first."""
    if len(smoothed) <= 5:
        return None
    i_min = int(np.argmin(smoothed))
    lo = max(0, i_min - 50)
    if lo >= i_min:
        return None
    seg_lr, seg_loss = np.asarray(lrs, dtype=np.float64)[lo:i_min], np.asarray(smoothed, dtype=np.float64)[lo:i_min]
    if len(seg_lr) <= 1:
        return None
    steepest = int(np.argmin(np.gradient(seg_loss, seg_lr)))
    return float(seg_lr[max(0, steepest - 5)])


def smooth(losses: Sequence[float], beta: float):
    """lr_finder.py:206: s_0 = l_0, s_i = beta * s_{i-1} + (1 - beta) * l_i."""
    out = []
    for i, l in enumerate(losses):
        out.append(l if i == 0 else out[-1] * beta + l * (1.0 - beta))
    return out


def next_n_views_rendering(cfg, rng: random.Random = random) -> int:
    """core/train.py:335-337: with TRAIN.UPDATE_N_VIEWS_RENDERING the next epoch renders randint(1, CONST.N_VIEWS_RENDERING) views."""
    n = int(cfg.CONST.N_VIEWS_RENDERING)
    return rng.randint(1, n) if cfg.TRAIN.get("UPDATE_N_VIEWS_RENDERING", False) else n


def lr_range_test(nets, cfg, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], start_lr: Optional[float] = None, end_lr: Optional[float] = None,
                  num_batches: Optional[int] = None, avg_beta: Optional[float] = None, restore: bool = True) -> dict:
    """Returns {"lrs", "losses", "smoothed", "suggested_lr", "suggested_merger_refiner_lr", "diverged_at"}; one host read of the loss per batch
    (the divergence test needs it, as in the reference)."""
    lf = dict(LR_FINDER_DEFAULTS)
    lf.update(cfg.get("LR_FINDER", {}))
    start_lr = lf["START_LR"] if start_lr is None else start_lr
    end_lr = lf["END_LR"] if end_lr is None else end_lr
    num_batches = lf["NUM_BATCHES_TO_TEST"] if num_batches is None else num_batches
    avg_beta = lf["AVG_BETA"] if avg_beta is None else avg_beta
    if num_batches < 2:
        raise ValueError("lr_range_test needs at least two batches")
    enc, dec, mer, ref = nets
    initial = [copy.deepcopy(n.state_dict()) for n in nets] if restore else None
    t = cfg.TRAIN
    if t.POLICY == "adam":
        mk = lambda m, lr: FlatAdam([p for p in m.parameters() if p.requires_grad], lr=lr, betas=tuple(t.BETAS))
    elif t.POLICY == "sgd":
        mk = lambda m, lr: FlatSGD([p for p in m.parameters() if p.requires_grad], lr=lr, momentum=t.MOMENTUM)
    else:
        raise Exception(f"[FATAL] Unknown optimizer: {t.POLICY}")     # lr_finder.py:104
    solvers = [mk(enc, start_lr), mk(dec, start_lr), mk(mer, start_lr / 10.0), mk(ref, start_lr / 10.0)]
    ungated = Cfg(cfg)                      # the finder applies USE_MERGER / USE_REFINER from the first batch on
    ungated.TRAIN = Cfg(cfg.TRAIN)
    ungated.TRAIN.EPOCH_START_USE_MERGER = 0
    ungated.TRAIN.EPOCH_START_USE_REFINER = 0
    mult = (end_lr / start_lr) ** (1.0 / (num_batches - 1))
    was_training = [n.training for n in nets]
    for n in nets:
        n.train()
    lrs, losses, smoothed, best, diverged_at = [], [], [], float("inf"), None
    try:
        for i, (images, gt) in enumerate(batches):
            if i >= num_batches:
                break
            lr = solvers[0].param_groups[0]["lr"]
            lrs.append(lr)
            el, rl = harness.train_step(nets, solvers, ungated, images, gt, epoch_idx=0)
            uses_refiner = bool(cfg.NETWORK.USE_REFINER)
            loss = float(el + rl) if uses_refiner else float(el)      # total_loss.item(), lr_finder.py:204
            losses.append(loss)
            smoothed.append(loss if i == 0 else smoothed[-1] * avg_beta + loss * (1.0 - avg_beta))
            best = min(best, loss)
            if smoothed[-1] > 10.0 * best and i > 10:
                diverged_at = lr
                break
            for s in solvers:
                s.param_groups[0]["lr"] *= mult
    finally:
        if restore:
            for n, sd in zip(nets, initial):
                n.load_state_dict(sd)
        for n, tr in zip(nets, was_training):
            n.train(tr)
    sug = suggest_lr(lrs, smoothed)
    return {"lrs": lrs, "losses": losses, "smoothed": smoothed, "suggested_lr": sug,
            "suggested_merger_refiner_lr": None if sug is None else sug / 10.0, "diverged_at": diverged_at}
