"""Whole optimisation step (harness.train_step: forward + backward + clip + solver) with the flat-buffer solvers and with
the stock torch.optim sequence, next to forward+backward alone.  python scripts/bench_train_step.py [--batch 32]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S  # noqa: E402
from swinvox_amd import harness  # noqa: E402
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = S.default_cfg()
S.set_math("bf16")
S.set_storage("bf16")
x = (0.5 * torch.randn(a.batch, a.views, 3, 224, 224, device=dev)).clamp(-1, 1)
gt = (torch.rand(a.batch, 32, 32, 32, device=dev) < 0.1).float()


def timed(fn, n):
    fn()
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for fused in (True, False, None):
    torch.manual_seed(0)
    nets = [m(cfg).to(dev).train() for m in (Encoder, Decoder, Merger, Refiner)]
    if fused is None:
        def step():
            for n in nets:
                n.zero_grad(set_to_none=True)
            harness.forward_losses(nets, cfg, x, gt)[0].backward()
        name = "forward+backward only"
    else:
        solvers, _ = harness.make_solvers(nets, cfg, fused=fused)
        def step():
            harness.train_step(nets, solvers, cfg, x, gt)
        name = "train_step, flat solvers" if fused else "train_step, torch.optim + clip_grad_norm_"
    ms = timed(step, a.steps)
    print(f"{name:45s} {ms:8.2f} ms/step  {a.batch * a.views / ms * 1e3:8.1f} views/s", flush=True)
    del nets
    torch.cuda.empty_cache()
