#!/usr/bin/env python3
"""Dense-layer shapes of the B=64 x V=8 step (forward / data-gradient of the Swin stage 2-3 Linears and the ResNet layer-3 1x1 convs) on the
contraction engine.  Run twice to compare the wide kernel with the 128-wide ones:

  python scripts/bench_gemm_wide.py            # wide kernel where eligible
  SV_GEMM_WIDE=0 python scripts/bench_gemm_wide.py
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S  # noqa: E402
from swinvox_amd import hip  # noqa: E402
from swinvox_amd.ops import ACT_GELU, ConvSpec  # noqa: E402

SHAPES = [  # M, K, N, epilogue
    (100352, 384, 1536, "agelu"), (401408, 768, 192, "resscale"), (401408, 768, 192, ""), (401408, 192, 768, "gelu"), (401408, 192, 768, "agelu"), (1605632, 384, 192, ""),
    (100352, 1536, 384, "agelu"), (100352, 1536, 384, "resscale"), (100352, 1024, 256, "arelu"), (100352, 1024, 256, "stats"), (25088, 3072, 768, "agelu"),
    (100352, 384, 1536, "gelu"), (100352, 1536, 384, "res"), (100352, 384, 1152, "bias"), (100352, 384, 384, "res"), (100352, 1536, 384, ""),
    (100352, 384, 1536, ""), (25088, 768, 3072, "gelu"), (25088, 3072, 768, "res"), (25088, 768, 2304, "bias"), (25088, 768, 768, "res"),
    (100352, 256, 1024, "stats"), (100352, 1024, 256, "stats"), (401408, 128, 512, "stats"), (401408, 512, 128, "stats"),
    (401408, 192, 576, "bias"), (401408, 192, 192, "res"), (1605632, 64, 256, "stats"), (1605632, 256, 64, "stats"), (1605632, 96, 288, "bias")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--wgrad", action="store_true", help="time the weight gradients of the same layers instead (A/B with SV_WGRAD_WIDE=0)")
    ap.add_argument("--cold", action="store_true", help="sweep 1 GB through the caches before every timed launch (the state a layer finds inside a step)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    hip.load()
    S.set_math("bf16"); S.set_storage("bf16")
    tot = 0.0
    sweep_a = torch.zeros(512 << 20, dtype=torch.uint8, device=dev) if a.cold else None
    sweep_b = torch.zeros(512 << 20, dtype=torch.uint8, device=dev) if a.cold else None
    for M, K, N, epi in SHAPES:
        sp = ConvSpec.linear(K, N)
        x = torch.randn(M, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev) * 0.05
        wp = sp.pack_fwd(w)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        kw = {}
        if epi in ("gelu", "bias", "res", "stats"):
            kw["bias"] = torch.randn(N, device=dev)
        if epi == "gelu":
            kw["act"] = ACT_GELU; kw["pre_act"] = torch.empty_like(y)
        if epi == "res":
            kw["residual"] = torch.randn(M, N, device=dev).bfloat16(); kw["ldr"] = N
        if epi in ("agelu", "arelu"):
            kw["act_grad_src"] = torch.randn(M, N, device=dev).bfloat16()
            kw["act_grad_kind"] = ACT_GELU if epi == "agelu" else 1
        if epi == "resscale":
            kw["bias"] = torch.randn(N, device=dev)
            kw["residual"] = torch.randn(M, N, device=dev).bfloat16(); kw["ldr"] = N
            kw["row_scale"] = torch.rand(M // 49, device=dev); kw["rows_per_scale"] = 49
        if epi == "stats":
            kw["stats"] = torch.zeros(16, 2 * N, dtype=torch.float64, device=dev)
        fn = lambda: sp.forward(x, M, (1, 1, 1), wp, y, **kw)
        if a.wgrad:
            if epi not in ("", "gelu", "stats"):
                continue
            dw, db = torch.zeros(N, K, device=dev), (torch.zeros(N, device=dev) if epi == "gelu" else None)
            dy = torch.randn(M, N, device=dev).bfloat16()
            fn = lambda: sp.wgrad(dy, x, M, (1, 1, 1), dw, db=db, async_ok=False)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if a.cold:
            us = 0.0
            for _ in range(a.iters):
                sweep_b.copy_(sweep_a)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                us += e0.elapsed_time(e1) / a.iters * 1e3
        else:
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
        tot += us
        flops = 2.0 * M * K * N
        print(f"M={M:8d} K={K:5d} N={N:5d} {epi:6s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF/s", flush=True)
    print(f"TOTAL {tot:.1f} us  (SV_GEMM_WIDE={os.environ.get('SV_GEMM_WIDE', '1')} SV_WGRAD_WIDE={os.environ.get('SV_WGRAD_WIDE', '1')})")


if __name__ == "__main__":
    main()
