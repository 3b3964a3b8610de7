#!/usr/bin/env python3
"""Halo-tile kernels (csrc/conv_halo.hip) against the gather engine on the calls they take: 3 x 3 / 64 -> 64 at 56 x 56 (forward with BatchNorm
statistics, data gradient) and the 4 x 4 stem at 112 x 112, 512 images, bf16."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
n = int(os.environ.get("SV_I", "512"))
for name, sp, H, ci, k in (("3x3 64->64 @56", ConvSpec.conv2d(64, 64, 3, 1, 1), 56, 64, 3),
                           ("stem 4x4 16->64 @112", ConvSpec.conv2d(16, 64, 4, 1, 2, og_fixed=(1, 112, 112)), 112, 16, 4)):
    M = n * H * H
    x = torch.randn(M, ci, device=dev).bfloat16()
    dy = torch.randn(M, 64, device=dev).bfloat16()
    w = torch.randn(64, ci, k, k, device=dev) / math.sqrt(ci * k * k)
    wf, wd = ops.pack_one(sp, w, "f"), ops.pack_one(sp, w, "d")
    out, dx = ops.empty(M, 64, device=dev), ops.empty(M, ci, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 128, dtype=torch.float64, device=dev)
    flops = 2.0 * M * k * k * ci * 64
    mb = (M * ci + M * 64) * 2 / 1e6
    for mode in (0, 2):
        ops.set_conv_halo(mode)
        tf = timeit(lambda: sp.forward(x, n, (1, H, H), wf, out, stats=stats))
        line = f"{name}  mode {mode}: forward+stats {tf:7.1f} us ({flops / tf / 1e6:6.1f} TF/s, {mb / tf * 1e3:5.0f} GB/s)"
        if ci == 64:
            td = timeit(lambda: sp.dgrad(dy, n, (1, H, H), wd, dx))
            line += f"   data gradient {td:7.1f} us ({flops / td / 1e6:6.1f} TF/s)"
        print(line)
# ---- 3 x 3 / stride 1 with 128 / 256 channels: gather engine against the blocked halo kernel
for H, C in ((28, 128), (14, 256)):
    sp = ConvSpec.conv2d(C, C, 3, 1, 1)
    M = n * H * H
    x = torch.randn(M, C, device=dev).bfloat16()
    dy = torch.randn(M, C, device=dev).bfloat16()
    w = torch.randn(C, C, 3, 3, device=dev) / math.sqrt(9 * C)
    wf, wd = ops.pack_one(sp, w, "f"), ops.pack_one(sp, w, "d")
    out, dx = ops.empty(M, C, device=dev), ops.empty(M, C, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 2 * C, dtype=torch.float64, device=dev)
    flops = 2.0 * M * 9 * C * C
    for mode in (0, 2):
        ops.set_conv_halo(mode)
        tf = timeit(lambda: sp.forward(x, n, (1, H, H), wf, out, stats=stats))
        td = timeit(lambda: sp.dgrad(dy, n, (1, H, H), wd, dx))
        print(f"3x3 {C}->{C} @{H}  mode {mode}: forward+stats {tf:7.1f} us ({flops / tf / 1e6:6.1f} TF/s)   data gradient {td:7.1f} us ({flops / td / 1e6:6.1f} TF/s)")
ops.set_conv_halo(1)
# ---- weight gradient of the 3 x 3 / stride 1 convolutions (sv_conv_wgrad): wgrad_kernel against the halo-tile kernel
for H, C, st in ((56, 64, 1), (28, 128, 1), (14, 256, 1), (7, 256, 1), (56, 256, 2), (28, 256, 2), (56, 128, 2), (14, 256, 2)):
    sp = ConvSpec.conv2d(C, C, 3, st, 1)
    Ho = (H - 1) // st + 1
    M, Mo = n * H * H, n * Ho * Ho
    x = torch.randn(M, C, device=dev).bfloat16()
    dy = torch.randn(Mo, C, device=dev).bfloat16()
    dw, db = torch.zeros(C, C, 3, 3, device=dev), torch.zeros(C, device=dev)
    flops = 2.0 * Mo * 9 * C * C
    line = f"3x3 {C:3d}->{C:3d} @{H:2d} stride {st} weight gradient{' + bias' if st == 2 else ''}:"
    for mode in (0, 2):
        ops.set_conv_halo_wgrad(mode)
        t = timeit(lambda: sp.wgrad(dy, x, n, (1, H, H), dw, db=db if st == 2 else None))
        line += f"   mode {mode}: {t:7.1f} us ({flops / t / 1e6:6.1f} TF/s)"
    print(line)
