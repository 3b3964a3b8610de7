#!/usr/bin/env python3
"""ResNet trunk on real renderings (eval, 512 images, bf16), each halo mode carrying ITS OWN activations forward: where does the difference
between the engine (0) and the halo kernels (2) grow?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swinvox_amd as S
from swinvox_amd import goldens, ops
from swinvox_amd.models import Encoder
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
enc = Encoder(S.default_cfg()); goldens.seeded_fill_(enc, 300); enc.to(dev)
g = torch.Generator().manual_seed(5)
images = (0.5 * torch.randn(64, 8, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
ops.set_math("bf16"); ops.set_storage("bf16")
enc.train()
with torch.no_grad():
    for mom in (0.1, 1.0):
        for m in enc.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.momentum = mom
        enc(images[:4])
enc.eval()
I = 512
acts = {}
for mode in (0, 2):
    ops.set_conv_halo(mode)
    with torch.no_grad():
        x, gq, _ = enc._stem_fwd(ops.to_store(images.view(I, 3, 224, 224)), I, False)
        lst = [("stem+pool", x.float())]
        for li in (4, 5, 6):
            for bi, blk in enumerate(enc.resnet[li]):
                x, gq, _ = blk.fwd(x, I, gq, False)
                lst.append((f"layer{li - 3}.{bi}", x.float()))
    acts[mode] = lst
for (name, a), (_, b) in zip(acts[0], acts[2]):
    d = (a - b).abs()
    print(f"{name:12s} max rel {float(d.max() / a.abs().max()):.5f}  mean rel {float(d.mean() / a.abs().mean()):.6f}  share of elements that differ {float((d > 0).float().mean()):.5f}")
