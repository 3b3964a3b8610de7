"""Host / GPU cost of one step: eager vs hipGraph replay, three streams vs one (what swinvox_amd/graph.py buys on this stack)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd.graph import GraphedStep
from swinvox_amd.helpers import init_weights
from swinvox_amd.losses import bce_with_logits as bce
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
S.set_math("bf16"); S.set_storage("bf16")
cfg = S.default_cfg()
torch.manual_seed(0)
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for n in nets:
    n.apply(init_weights); n.to(dev).train()
x = (0.5 * torch.randn(B, 8, 3, 224, 224)).clamp(-1, 1).to(dev)
gt = (torch.rand(B, 32, 32, 32) < 0.1).float().to(dev)

def compute():
    for n in nets:
        for p in n.parameters():
            p.grad = None
    raw, vol = nets[1](nets[0](x))
    merged = nets[2](raw, vol)
    total = bce(merged, gt) + bce(nets[3](merged), gt)
    total.backward()
    return total.detach()

def timeit(fn, steps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    return h / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3

for overlap in (True, False):
    S.set_overlap(overlap)
    for _ in range(3):
        compute()
    print(f"overlap={overlap} eager   host {timeit(compute)[0]:7.2f} ms  wall {timeit(compute)[1]:7.2f} ms", flush=True)
    g = GraphedStep(compute, warmup=1, device=dev)
    h, w = timeit(g)
    print(f"overlap={overlap} graph   host {h:7.2f} ms  wall {w:7.2f} ms", flush=True)
    del g
# host cost of the eager enqueue with an idle queue: enqueue one step, wait, repeat
S.set_overlap(True)
hs = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); compute(); hs.append((time.perf_counter() - t0) * 1e3); torch.cuda.synchronize()
print("eager host enqueue with an idle queue (ms):", [round(v, 1) for v in hs])
