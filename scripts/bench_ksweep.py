#!/usr/bin/env python3
"""K sweep of the Linear forward kernel at fixed M, N: separates the per-tile fixed cost from the per-K-step cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
if os.environ.get("SV_LIB"):
    hip.LIB_PATH = os.environ["SV_LIB"]      # A/B builds of the library
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, N in ((200704, 128), (200704, 256), (65536, 128), (65536, 512)):
    for K in (64, 128, 256, 512, 1024):
        sp = ConvSpec.linear(K, N)
        x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev) * 0.05
        wf = sp.pack_fwd(w); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: sp.forward(x, M, (1, 1, 1), wf, y))
        tiles = (M // 128) * ((N + 127) // 128)
        print(f"M={M} N={N} K={K:5d}: {t:7.1f} us  tiles={tiles}  per-tile-slot {t / max(tiles / 512, 1):6.2f} us  {2.0*M*N*K/t/1e6:6.1f} TF/s  {(M*K+M*N)*2/t/1e3:6.0f} GB/s")
