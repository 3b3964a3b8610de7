"""Harness counterparts of core/train.py / core/test.py around the HIP modules: gating knobs, optimisation
step order, on-device IoU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
import swinvox_amd as S  # noqa: E402
from swinvox_amd import harness  # noqa: E402
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner  # noqa: E402


def _nets(dev, cfg):
    torch.manual_seed(0)
    nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    for n in nets:
        O.seeded_weights_(n, seed=7)
        n.to(dev).train()
    return nets


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_train_steps_reduce_loss_and_eval_iou(dev, storage):
    cfg = S.default_cfg()
    cfg.TRAIN.ENCODER_LEARNING_RATE = cfg.TRAIN.DECODER_LEARNING_RATE = 1e-3
    cfg.TRAIN.REFINER_LEARNING_RATE = cfg.TRAIN.MERGER_LEARNING_RATE = 1e-3
    nets = _nets(dev, cfg)
    solvers, _ = harness.make_solvers(nets, cfg)
    g = torch.Generator().manual_seed(3)
    x = (0.5 * torch.randn(2, 2, 3, 224, 224, generator=g)).to(dev)
    gt = (torch.rand(2, 32, 32, 32, generator=g) < 0.1).float().to(dev)
    S.set_math("bf16")
    S.set_storage(storage)
    try:
        losses = []
        for _ in range(6):
            el, rl = harness.train_step(nets, solvers, cfg, x, gt)
            losses.append(float(el + rl))
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        for n in nets:
            n.eval()
        el10, rl10, iou, fsc = harness.evaluate(nets, cfg, x, gt, with_fscore=True)
        refined = nets[3](nets[2](*nets[1](nets[0](x))))
    finally:
        S.set_math("f32")
    ref = np.array(O.iou_at_thresholds(refined.detach().cpu(), gt.cpu()))
    assert iou.shape == (2, 4) and np.abs(iou.cpu().numpy() - ref).max() < 2e-3
    reff = np.array(O.fscore_at_thresholds(refined.detach().cpu(), gt.cpu()))
    assert fsc.shape == (2, 4) and np.abs(fsc.cpu().numpy() - reff).max() < 2e-3
    # both empty -> IoU 1 (core/test.py:153); prediction empty, ground truth not -> IoU 0 and F-score 0
    z = torch.full((2, 32, 32, 32), -9.0, device=dev)
    g2 = torch.zeros(2, 32, 32, 32, device=dev)
    g2[1, 0, 0, 0] = 1
    i2, f2 = harness.voxel_metrics(z, g2, cfg.TEST.VOXEL_THRESH)
    assert i2.tolist() == [[1.0] * 4, [0.0] * 4] and f2.tolist() == [[0.0] * 4, [0.0] * 4]


def test_gating_without_merger_and_refiner(dev):
    """USE_MERGER / USE_REFINER off -> mean over views (core/train.py:246) and refiner_loss = encoder_loss (:257)."""
    cfg = S.default_cfg()
    cfg.NETWORK.USE_MERGER, cfg.NETWORK.USE_REFINER = False, False
    nets = _nets(dev, cfg)
    for n in nets:
        n.stochastic = False        # dropout / drop-path off so the forward can be repeated
    g = torch.Generator().manual_seed(4)
    x = (0.5 * torch.randn(1, 3, 3, 224, 224, generator=g)).to(dev)
    gt = (torch.rand(1, 32, 32, 32, generator=g) < 0.1).float().to(dev)
    total, el, rl, volume, flags = harness.forward_losses(nets, cfg, x, gt)
    assert flags == (False, False) and float(rl) == float(el) == float(total)
    with torch.no_grad():
        _, vol = nets[1](nets[0](x))
    # BatchNorm is in train mode: recompute with the same batch statistics is not bit-identical, so compare the mean itself
    assert float((volume.detach() - vol.mean(1)).abs().max()) < 5e-3 * float(vol.abs().max())
    total.backward()
    assert nets[2].layer1[0].weight.grad is None and nets[0].layer3[0].weight.grad is not None
