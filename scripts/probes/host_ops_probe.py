import os, sys, torch
sys.path.insert(0, '/root/repo')
import swinvox_amd as S
from swinvox_amd import ops
from swinvox_amd.losses import bce_with_logits as bce
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0)
cfg = S.default_cfg()
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for n in nets: n.to(dev).train()
B, V = 8, 8
images = torch.randn(B, V, 3, 224, 224, device=dev); gt = (torch.rand(B, 32, 32, 32, device=dev) < 0.1).float()
ops.set_math("bf16"); ops.set_storage("bf16")
def step():
    for n in nets: n.zero_grad(set_to_none=True)
    f = nets[0](images); raw, vol = nets[1](f); m = nets[2](raw, vol); r = nets[3](m)
    (bce(m, gt) + bce(r, gt)).backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.count:5d} {e.self_cpu_time_total/1e3:8.2f} ms  {e.key[:90]}")
