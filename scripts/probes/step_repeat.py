#!/usr/bin/env python3
"""Two identical training steps at the bench shape (same seeds): which parameter gradients differ, and by how much (L1-relative)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swinvox_amd as S
from swinvox_amd import goldens, ops
from swinvox_amd.losses import bce_with_logits as bce
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0)
B, V = int(os.environ.get("SV_B", "64")), 8
torch.manual_seed(1234)
cfg = S.default_cfg()
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for i, n in enumerate(nets):
    goldens.seeded_fill_(n, 300 + i); n.to(dev); n.train()
g = torch.Generator().manual_seed(5)
images = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float().to(dev)
ops.set_math("bf16"); ops.set_storage("bf16")
if os.environ.get("SV_NO_OVERLAP"):
    S.set_overlap(False)
if os.environ.get("SV_ENC_ONLY"):
    nets = nets[:1]
def run():
    torch.manual_seed(77)
    for n in nets: n.zero_grad(set_to_none=True)
    f = nets[0](images)
    if len(nets) == 1:
        total = f.float().square().mean()
    else:
        raw, vol = nets[1](f); merged = nets[2](raw, vol); refined = nets[3](merged)
        total = bce(merged, gt) + bce(refined, gt)
    total.backward(); torch.cuda.synchronize()
    return float(total), [p.grad.clone() for n in nets for p in n.parameters()]
names = [f"{type(n).__name__}.{k}" for n in nets for k, _ in n.named_parameters()]
run()
for rep in range(int(os.environ.get("SV_REPS", "2"))):
    l1, g1 = run(); l2, g2 = run()
    rows = []
    for k, a, b in zip(names, g1, g2):
        den = float(a.abs().sum())
        if den > 0 and float(a.abs().max()) >= 1e-5:
            rows.append((float((a - b).abs().sum()) / den, k))
    rows.sort(reverse=True)
    print(f"rep {rep}: loss {l1:.9f} / {l2:.9f}; gradients differing by > 1e-5: {sum(r[0] > 1e-5 for r in rows)} of {len(rows)}; worst:", [(f"{e:.2e}", k) for e, k in rows[:6]])
    print("   all:", [k for e, k in rows if e > 1e-5])
