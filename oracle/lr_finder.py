"""CPU restatement of the reference's learning-rate finder loop (utils/lr_finder.py:84-256) on the oracle modules - TEST INFRASTRUCTURE ONLY
(tests/ may import it; the product never does).  Plain torch.optim solvers, one step per batch, no autocast / GradScaler (the parity modes
compare in fp32), the smoothing, the divergence stop and the suggestion rule written out as in the reference.  Parity unpinned against a run
of the reference itself: its module imports matplotlib, cv2-based loaders and a dataset that are not available here; the loop below follows the
reference text line by line and is pinned only through the shared oracle modules (tests/golden)."""
from __future__ import annotations

import numpy as np
import torch

from .model import train_step_loss


def range_test(nets, cfg, batches, start_lr, end_lr, num_batches, avg_beta, betas=(0.9, 0.999)):
    enc, dec, mer, ref = nets
    mk = lambda m, lr: torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=lr, betas=betas)     # lr_finder.py:85-92
    solvers = [mk(enc, start_lr), mk(dec, start_lr), mk(mer, start_lr / 10.0), mk(ref, start_lr / 10.0)]
    mult = (end_lr / start_lr) ** (1 / (num_batches - 1))                                                           # :118
    lrs, losses, smoothed, best = [], [], [], float("inf")
    for n in nets:
        n.train()
    for i, (images, gt) in enumerate(batches):
        if i >= num_batches:
            break
        lrs.append(solvers[0].param_groups[0]["lr"])                                                                # :137-138
        total, *_ = train_step_loss(nets, cfg, images, gt, epoch_idx=10 ** 9)                                       # :146-158 (no epoch gates)
        for s in solvers:
            s.zero_grad()
        total.backward()
        for n in nets:                                                                                              # :179-184
            torch.nn.utils.clip_grad_norm_(n.parameters(), max_norm=1.0)
        for s in solvers:
            s.step()
        loss = float(total.detach())
        losses.append(loss)
        smoothed.append(loss if i == 0 else smoothed[-1] * avg_beta + loss * (1 - avg_beta))                        # :206
        best = min(best, loss)
        if smoothed[-1] > 10 * best and i > 10:                                                                     # :212
            break
        for s in solvers:                                                                                           # :216-231
            for g in s.param_groups:
                g["lr"] *= mult
    return lrs, losses, smoothed


def suggest(lrs, smoothed):
    """utils/lr_finder.py:243-256."""
    if len(smoothed) > 5:
        min_loss_idx = np.argmin(smoothed)
        start = max(0, min_loss_idx - 50)
        if start < min_loss_idx:
            seg_lrs = np.array(lrs)[start:min_loss_idx]
            seg_loss = np.array(smoothed)[start:min_loss_idx]
            if len(seg_lrs) > 1:
                grad = np.gradient(seg_loss, seg_lrs)
                k = np.argmin(grad)
                return float(seg_lrs[max(0, k - 5)])
    return None
