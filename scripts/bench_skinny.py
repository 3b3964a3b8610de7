#!/usr/bin/env python3
"""1 x 1 convolutions that read a wide activation once and write a narrow one (conv1 of the ResNet bottlenecks, the data gradient of their
conv3): how fast is the activation streamed?  512 images, bf16, with and without the BatchNorm statistics epilogue.  SV_GEMM_WIDE=2 lets the
wide kernel take BatchNorm producers with K < 1024."""
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
n = 512
print("SV_GEMM_WIDE =", os.environ.get("SV_GEMM_WIDE", "1"))
for H, K, N in ((56, 256, 64), (56, 256, 128), (28, 512, 128), (28, 512, 256), (14, 1024, 256), (14, 1024, 512)):
    M = n * H * H
    sp = ConvSpec.conv2d(K, N, 1, 1, 0)
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, 1, 1, device=dev) / math.sqrt(K)
    wf = ops.pack_one(sp, w, "f")
    out = ops.empty(M, N, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 2 * N, dtype=torch.float64, device=dev)
    t0 = timeit(lambda: sp.forward(x, n, (1, H, H), wf, out))
    t1 = timeit(lambda: sp.forward(x, n, (1, H, H), wf, out, stats=stats))
    mb = M * (K + N) * 2 / 1e6
    fl = 2.0 * M * K * N
    print(f"{H:3d}^2 {K:5d} -> {N:4d}: plain {t0:6.0f} us ({mb / t0:5.2f} TB/s, {fl / t0 / 1e6:5.0f} TF/s)   with statistics {t1:6.0f} us ({mb / t1:5.2f} TB/s, {fl / t1 / 1e6:5.0f} TF/s)")
