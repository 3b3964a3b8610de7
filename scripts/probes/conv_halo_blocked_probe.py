import math, os, sys, torch
sys.path.insert(0, '/root/repo')
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
for n, H, C in ((130, 28, 128), (300, 14, 256), (512, 28, 128), (512, 14, 256)):
    sp = ConvSpec.conv2d(C, C, 3, 1, 1)
    M = n * H * H
    x = torch.randn(M, C, device=dev).bfloat16()
    w = torch.randn(C, C, 3, 3, device=dev) / math.sqrt(9 * C)
    wf = ops.pack_one(sp, w, "f")
    outs = []
    for mode, st in ((2, True), (2, False), (0, False)):
        ops.set_conv_halo(mode)
        out = ops.empty(M, C, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 2 * C, dtype=torch.float64, device=dev) if st else None
        sp.forward(x, n, (1, H, H), wf, out, stats=stats)
        torch.cuda.synchronize()
        outs.append(out.float())
    for i, name in ((0, "halo+stats"), (1, "halo")):
        d = (outs[i] - outs[2]).abs()
        bad = (d > 0.05).any(1)
        print(n, H, C, name, "max diff", float(d.max()), "rows off", int(bad.sum()), "of", M, "first bad rows", torch.nonzero(bad).flatten()[:6].tolist())
