#!/usr/bin/env python3
"""ResNet layers 1-3 alone (eval mode, 512 images, bf16): output with the halo kernels on (2) vs off (0), block by block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swinvox_amd as S
from swinvox_amd import goldens, ops
from swinvox_amd.models import Encoder
dev = torch.device("cuda", 0)
torch.manual_seed(1)
enc = Encoder(S.default_cfg()); goldens.seeded_fill_(enc, 300); enc.to(dev).eval()
ops.set_math("bf16"); ops.set_storage("bf16")
n = 512
for li, (C, H) in ((4, (64, 56)), (5, (256, 56)), (6, (512, 28))):
    x = torch.randn(n * H * H, C, device=dev).bfloat16().relu()
    for bi, blk in enumerate(enc.resnet[li]):
        outs = []
        for mode in (0, 2):
            ops.set_conv_halo(mode)
            with torch.no_grad():
                a, g, _ = blk.fwd(x, n, (1, H, H), False)
            torch.cuda.synchronize()
            outs.append(a.float())
        d = (outs[0] - outs[1]).abs()
        print(f"layer{li - 3} block {bi}: in {tuple(x.shape)} out {tuple(outs[0].shape)}  max rel {float(d.max() / outs[0].abs().max()):.5f}  mean rel {float(d.mean() / outs[0].abs().mean()):.6f}")
        x, H = outs[0].bfloat16(), g[1]
