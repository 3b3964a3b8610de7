#!/usr/bin/env python3
"""cProfile of the host side of one bench step (B=8, V=8): where the Python / ctypes enqueue time goes."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
import oracle as O
from swinvox_amd import hip
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
cfg = S.default_cfg()
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for n in nets:
    n.apply(O.init_weights); n.to(dev).train()
B, V = 8, 8
images = (0.5 * torch.randn(B, V, 3, 224, 224)).clamp(-1, 1).to(dev)
gt = (torch.rand(B, 32, 32, 32) < 0.1).float().to(dev)
bce = torch.nn.functional.binary_cross_entropy_with_logits
def step():
    for n in nets:
        for p in n.parameters(): p.grad = None
    raw, vol = nets[1](nets[0](images)); merged = nets[2](raw, vol); refined = nets[3](merged)
    (bce(merged, gt) + bce(refined, gt)).backward()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(3): step()
host = time.perf_counter() - t0
pr.disable(); torch.cuda.synchronize()
print("host ms/step under cProfile", host / 3 * 1e3)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(35)
