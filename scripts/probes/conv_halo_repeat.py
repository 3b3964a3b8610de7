#!/usr/bin/env python3
"""Repeatability of the halo-tile kernels at the bench shape: the stored output must be bit-identical from run to run (tiles are drawn
at run time, but a tile's result does not depend on who computes it) and equal to the gather engine's up to single bf16 roundings."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
n = int(os.environ.get("SV_I", "512"))
for name, sp, H, ci, k in (("3x3", ConvSpec.conv2d(64, 64, 3, 1, 1), 56, 64, 3), ("stem", ConvSpec.conv2d(16, 64, 4, 1, 2, og_fixed=(1, 112, 112)), 112, 16, 4)):
    M = n * H * H
    x = torch.randn(M, ci, device=dev).bfloat16()
    w = torch.randn(64, ci, k, k, device=dev) / math.sqrt(ci * k * k)
    wf = ops.pack_one(sp, w, "f")
    outs, sts = [], []
    for mode in (2, 2, 2, 0):
        ops.set_conv_halo(mode)
        out = ops.empty(M, 64, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 128, dtype=torch.float64, device=dev)
        sp.forward(x, n, (1, H, H), wf, out, stats=stats)
        torch.cuda.synchronize()
        outs.append(out); sts.append(stats.sum(0))
    for i in (1, 2):
        d = (outs[i].float() - outs[0].float()).abs()
        nz = int((d > 0).sum())
        print(f"{name}: run {i} vs run 0: {nz} elements differ (max {float(d.max()):.3e}); statistics differ by {float((sts[i] - sts[0]).abs().max() / sts[0].abs().max()):.2e}")
        if nz:
            rows = torch.nonzero((d > 0).any(1)).flatten()
            r = rows.cpu().numpy()
            print("   rows", r[:12], "... count", len(r), " image/row/col of first:", r[0] // (H * H), (r[0] // H) % H, r[0] % H)
    if ci == 64:
        dy = torch.randn(M, 64, device=dev).bfloat16()
        wd = ops.pack_one(sp, w, "d")
        dxs = []
        for mode in (2, 2, 2, 0):
            ops.set_conv_halo(mode)
            dx = ops.empty(M, 64, device=dev)
            sp.dgrad(dy, n, (1, H, H), wd, dx)
            torch.cuda.synchronize()
            dxs.append(dx)
        for i in (1, 2, 3):
            d = (dxs[i].float() - dxs[0].float()).abs()
            print(f"{name} data gradient: run {i}{' (engine)' if i == 3 else ''} vs run 0: {int((d > 0).sum())} elements differ (max {float(d.max()):.3e})")
    d = (outs[3].float() - outs[0].float()).abs()
    print(f"{name}: engine vs halo: max {float(d.max()):.3e}  mean {float(d.mean()):.3e}  (|y| mean {float(outs[3].float().abs().mean()):.3e})")
