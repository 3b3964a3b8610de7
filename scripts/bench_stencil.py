#!/usr/bin/env python3
"""The five stencil launches of the merger (models/merger.py) at the bench shape (I = B*V images of 32^3 voxels), bf16 storage:
time, algorithmic GB/s (operand + result bytes once) and the fraction of 8 TB/s.

  python scripts/bench_stencil.py [--images 512] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S  # noqa: E402
from swinvox_amd import hip, ops  # noqa: E402
from swinvox_amd.ops import call, ptr  # noqa: E402

VOX = 32768


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    hip.load()
    S.set_math("bf16"); S.set_storage("bf16")
    I, M = a.images, a.images * VOX
    bf = torch.bfloat16
    x12 = torch.randn(M, 12, device=dev).to(bf)
    y12 = torch.empty(M, 12, device=dev, dtype=bf)
    planes = torch.randn(4, M, 12, device=dev).to(bf)
    dplanes = torch.empty(4, M, 12, device=dev, dtype=bf)
    w1 = (torch.randn(16, 27, 16, device=dev) * 0.05).to(bf)
    w3 = (torch.randn(16, 27, 48, device=dev) * 0.05).to(bf)
    wd3 = (torch.randn(48, 27, 16, device=dev) * 0.05).to(bf)
    bias = torch.randn(9, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 18, dtype=torch.float64, device=dev)
    dw1, dw3 = torch.zeros(9, 9, 27, device=dev), torch.zeros(9, 36, 27, device=dev)
    db = torch.zeros(9, device=dev)
    ws1, ws3 = torch.zeros(16 * 9 * 9 * 27, device=dev), torch.zeros(16 * 9 * 36 * 27, device=dev)
    cases = {
        "fwd  9->9   (G=1, NT=1, stats)": (lambda: call("sv_stencil3_fwd", ptr(x12), 12, 12, 1, ptr(w1), 1, ptr(bias), ptr(y12), 12, 0, 9, None, 0, ptr(stats), I, 32, 32, 32, 0, 0), 2 * M * 24),
        "dgrad 9->9  (G=1, NT=1, +=)": (lambda: call("sv_stencil3_fwd", ptr(x12), 12, 12, 1, ptr(w1), 1, None, ptr(y12), 12, 0, 9, ptr(y12), 12, None, I, 32, 32, 32, 0, 0), 3 * M * 24),
        "fwd  36->9  (G=3, NT=1, planes)": (lambda: call("sv_stencil3_fwd", ptr(planes), 12, 48, 3, ptr(w3), 1, ptr(bias), ptr(y12), 12, 0, 9, None, 0, ptr(stats), I, 32, 32, 32, M * 12, 0), 5 * M * 24),
        "dgrad 9->36 (G=1, NT=3, planes)": (lambda: call("sv_stencil3_fwd", ptr(x12), 12, 12, 1, ptr(wd3), 3, None, ptr(dplanes), 12, 0, 48, None, 0, None, I, 32, 32, 32, 0, M * 12), 5 * M * 24),
        "wgrad 9x9   (G=1)": (lambda: call("sv_stencil3_wgrad", ptr(x12), 12, 12, 1, ptr(y12), 12, 12, ptr(dw1), ptr(db), ptr(ws1), 9, 9, 16, 9, I, 32, 32, 32, 0), 2 * M * 24),
        "wgrad 9x36  (G=3, planes)": (lambda: call("sv_stencil3_wgrad", ptr(planes), 12, 48, 3, ptr(y12), 12, 12, ptr(dw3), ptr(db), ptr(ws3), 9, 36, 12, 9, I, 32, 32, 32, M * 12), 5 * M * 24),
    }
    tot = 0.0
    for name, (fn, nbytes) in cases.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        tot += us
        print(f"{name:34s} {us:9.1f} us  {nbytes / us / 1e3:7.0f} GB/s  ({nbytes / us / 1e3 / 8000:.2f} of 8 TB/s)", flush=True)
    print(f"TOTAL {tot:.1f} us")


if __name__ == "__main__":
    main()
