#!/usr/bin/env python3
"""Host-side cost of one bench step with an idle GPU queue: forward enqueue, backward enqueue, and cProfile of both
(the backward runs in this thread through torch.autograd.grad-free manual call of the module backward)."""
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
def fwd():
    for n in nets:
        for p in n.parameters(): p.grad = None
    raw, vol = nets[1](nets[0](images)); merged = nets[2](raw, vol); refined = nets[3](merged)
    return bce(merged, gt) + bce(refined, gt)
for _ in range(3): fwd().backward()
torch.cuda.synchronize()
tf = tb = 0.0
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); loss = fwd(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter(); loss.backward(); t3 = time.perf_counter()
    tf += t1 - t0; tb += t3 - t2
print(f"host enqueue with idle queue: forward {tf/5*1e3:.2f} ms, backward {tb/5*1e3:.2f} ms")
# cProfile of the backward: run the engine inline in this thread
torch.cuda.synchronize()
loss = fwd(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
with torch.autograd.set_multithreading_enabled(False):
    loss.backward()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
