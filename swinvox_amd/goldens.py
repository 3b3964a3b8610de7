"""The weight / input recipe behind the committed golden vectors (tests/golden/*.npz), stated for the HIP modules.

The golden weights are never committed (the refiner alone is 140 MB): they are a deterministic per-tensor-name fill at
default-init scale followed by the calibration of SURVEY 8c (one train-mode pass sets the BatchNorm running statistics,
then the last layer of decoder / refiner is rescaled so that eval-mode logits have std ~2 - with the reference's own
init_weights every eval logit is ~1e-10 and a 1e-3 check says nothing).  tests/golden/make_golden.py runs this recipe on the
CPU oracle; this file runs the same recipe on the product's modules, so that a caller on a GPU box (bench.py's
`iou_delta_vs_oracle`) can rebuild the golden network without the oracle and compare with the stored fp32 outputs.
The calibration pass runs in exact-fp32 mode; its result differs from the CPU calibration by fp32 rounding only.
"""
from __future__ import annotations

import math
import zlib

import torch
import torch.nn as nn


def synth_images(B: int, V: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)


def synth_gt(B: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed + 1000)
    return (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float()


@torch.no_grad()
def seeded_fill_(module: nn.Module, seed: int) -> None:
    """Generator seeded by crc32(tensor name) + 7919 * seed: Conv/Linear weights U(-1, 1) * sqrt(3 / fan_in), small biases, norm
    gains around 1, jittered BatchNorm running statistics, relative-position tables N(0, 0.2)."""
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if not t.dtype.is_floating_point:
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_mean":
            v = 0.1 * torch.randn(t.shape, generator=g)
        elif leaf == "running_var":
            v = 0.5 + torch.rand(t.shape, generator=g)
        elif leaf == "relative_position_bias_table":
            v = 0.2 * torch.randn(t.shape, generator=g)
        elif t.dim() <= 1 or (leaf in ("weight", "bias") and ".layer_norm." in name):
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g) if leaf == "weight" else 0.05 * torch.randn(t.shape, generator=g)
        else:
            bound = 1.0 / math.sqrt(max(t[0].numel(), 1))
            v = (torch.rand(t.shape, generator=g) * 2 - 1) * bound * math.sqrt(3.0)
        t.copy_(v.to(t.device))


@torch.no_grad()
def calibrate_(nets, images: torch.Tensor, logit_std: float = 2.0) -> None:
    enc, dec, mer, ref = nets
    bns = [m for n in nets for m in n.modules() if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d))]
    sto = [n.stochastic for n in nets]
    for n in nets:
        n.train()
        n.stochastic = False
    for m in bns:
        m.momentum = 1.0
    raw, vol = dec(enc(images))
    ref(mer(raw, vol))
    for m in bns:
        m.momentum = 0.1
    for n, s in zip(nets, sto):
        n.eval()
        n.stochastic = s
    raw, vol = dec(enc(images))
    dec.layer5[0].weight.mul_(logit_std / float(vol.std().clamp_min(1e-12)))
    raw, vol = dec(enc(images))
    merged = mer(raw, vol)
    out = ref(merged)
    delta = out * 2 - merged                               # output of the refiner's last layer
    ref.layer8[0].weight.mul_(logit_std / float(delta.std().clamp_min(1e-12)))


def golden_case(dev: torch.device, B: int, V: int, seed: int):
    """(eval-mode HIP nets holding the golden weights, images [B,V,3,224,224], ground truth [B,32,32,32]) on `dev`."""
    from . import ops
    from .config import default_cfg
    from .models import Decoder, Encoder, Merger, Refiner
    cfg = default_cfg()
    nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    for i, n in enumerate(nets):
        seeded_fill_(n, 100 + i)
        n.to(dev)
    math_mode, store = ops.get_math(), ops.get_storage()
    ops.set_math("f32")
    try:
        calibrate_(nets, synth_images(2, 2, 1234).to(dev))
    finally:
        ops.set_math(math_mode)
        ops.set_storage(store)
    for n in nets:
        n.eval()
    return nets, synth_images(B, V, seed).to(dev), synth_gt(B, seed).to(dev)
