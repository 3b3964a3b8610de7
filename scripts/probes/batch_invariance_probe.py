#!/usr/bin/env python3
"""Eval forward of 64 samples vs the same samples as 32 x 2, with the halo kernels off / sized by the call (1) / always on (2) in BOTH arms:
how much of the difference is the kernels' choice (rounding-level differences amplified by the fixture) and how much tile scheduling."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swinvox_amd as S
from swinvox_amd import goldens, ops
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0)
B, V = 64, 8
torch.manual_seed(1234)
cfg = S.default_cfg()
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for i, n in enumerate(nets):
    goldens.seeded_fill_(n, 300 + i); n.to(dev)
g = torch.Generator().manual_seed(5)
images = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
ops.set_math("bf16"); ops.set_storage("bf16")
for n in nets: n.train()
with torch.no_grad():
    for mom in (0.1, 1.0):
        for n in nets:
            for m in n.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.momentum = mom
        raw, vol = nets[1](nets[0](images[:4])); nets[3](nets[2](raw, vol))
for n in nets: n.eval()
def fwd(x):
    f = nets[0](x); raw, vol = nets[1](f); m = nets[2](raw, vol); return f, vol, m, nets[3](m)
res = {}
for mode in (0, 1, 2):
    ops.set_conv_halo(mode)
    with torch.no_grad():
        whole = [t.clone() for t in fwd(images)]
        parts = [[], [], [], []]
        for b0 in range(0, B, 2):
            for lst, t in zip(parts, fwd(images[b0:b0 + 2])): lst.append(t.clone())
    res[mode] = whole
    rep = {}
    for name, w, p in zip(("features", "gen_volumes", "merged", "refined"), whole, parts):
        p = torch.cat(p, 0); d = (w - p).abs()
        rep[name] = (round(float(d.max() / w.abs().max()), 5), round(float(d.mean() / w.abs().mean()), 5))
    print("halo mode", mode, "64 vs 32x2 (max, mean relative):", rep)
for a, b in ((0, 1), (0, 2)):
    d = (res[a][0] - res[b][0]).abs()
    print(f"features, B = 64, mode {a} vs mode {b}: max {float(d.max() / res[a][0].abs().max()):.5f} mean {float(d.mean() / res[a][0].abs().mean()):.5f}")
