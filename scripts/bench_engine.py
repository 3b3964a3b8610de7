#!/usr/bin/env python3
"""Micro-benchmark of the contraction engine on the shapes that dominate bench.py (B=8, V=8): forward / data-gradient /
weight-gradient of Linear and Conv layers, bf16 storage.  Prints time, algorithmic TFLOP/s and GB/s per shape.

  python scripts/bench_engine.py [--iters 20] [--filter fwd|dgrad|wgrad]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S  # noqa: E402
from swinvox_amd import hip, ops  # noqa: E402
from swinvox_amd.ops import ConvSpec  # noqa: E402

LINEAR = [(12544, 384, 1536), (12544, 1536, 384), (12544, 384, 1152), (12544, 384, 384), (200704, 96, 384), (200704, 384, 96),
          (200704, 96, 288), (200704, 96, 96), (50176, 192, 768), (50176, 768, 192), (50176, 192, 576), (3136, 768, 3072), (3136, 3072, 768)]
CONV = [  # n, H, cin, cout, k, s, p
    (64, 56, 64, 256, 1, 1, 0), (64, 56, 256, 64, 1, 1, 0), (64, 56, 64, 64, 3, 1, 1), (512, 56, 64, 64, 3, 1, 1), (512, 28, 128, 128, 3, 1, 1), (512, 14, 256, 256, 3, 1, 1), (64, 28, 128, 512, 1, 1, 0), (64, 28, 512, 128, 1, 1, 0),
    (64, 28, 128, 128, 3, 1, 1), (64, 14, 256, 1024, 1, 1, 0), (64, 14, 1024, 256, 1, 1, 0), (64, 14, 256, 256, 3, 1, 1), (64, 56, 256, 256, 3, 2, 1)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--filter", default="")
    ap.add_argument("--storage", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    hip.load()
    S.set_math("bf16")
    S.set_storage(a.storage)
    dt = torch.bfloat16 if a.storage == "bf16" else torch.float32
    esz = 2 if a.storage == "bf16" else 4
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}

    def report(kind, tag, us, flops, nbytes):
        tot[kind] += us
        print(f"{kind:6s} {tag:44s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s  {nbytes / us / 1e3:7.0f} GB/s")

    cases = [("lin", (M, 1, K, N, 1, 1, 0)) for (M, K, N) in LINEAR] + [("conv", c) for c in CONV]
    for typ, (n, H, cin, cout, k, s, p) in cases:
        sp = ConvSpec.linear(cin, cout) if typ == "lin" else ConvSpec.conv2d(cin, cout, k, s, p)
        grid = (1, 1, 1) if typ == "lin" else (1, H, H)
        og = sp.out_grid(grid)
        Min, Mout = n * grid[1] * grid[2], n * og[1] * og[2]
        x = torch.randn(Min, cin, device=dev).to(dt)
        dy = torch.randn(Mout, cout, device=dev).to(dt)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05 if typ == "conv" else torch.randn(cout, cin, device=dev) * 0.05
        wf, wd = sp.pack_fwd(w), sp.pack_dgrad(w)
        y, dx = torch.empty(Mout, cout, device=dev, dtype=dt), torch.empty(Min, cin, device=dev, dtype=dt)
        dw = torch.zeros_like(w)
        flops = 2.0 * Mout * k * k * cin * cout
        nbytes = esz * (Min * cin + Mout * cout) + esz * k * k * cin * cout
        tag = f"n={n} H={H} {cin}->{cout} k{k} s{s}"
        if not a.filter or a.filter == "fwd":
            report("fwd", tag, timeit(lambda: sp.forward(x, n, grid, wf, y), a.iters), flops, nbytes)
        if not a.filter or a.filter == "dgrad":
            report("dgrad", tag, timeit(lambda: sp.dgrad(dy, n, grid, wd, dx), a.iters), flops, nbytes)
        if not a.filter or a.filter == "wgrad":
            report("wgrad", tag, timeit(lambda: sp.wgrad(dy, x, n, grid, dw), a.iters), flops, nbytes)
    print("TOTAL us", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
