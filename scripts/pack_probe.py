"""Which weight packs bypass the per-module batched pack (ops.PackCache)?  Counts ops.pack_one calls per step by layer geometry."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import ops
from swinvox_amd.helpers import init_weights
from swinvox_amd.losses import bce_with_logits as bce
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda:0")
S.set_math("bf16"); S.set_storage("bf16")
cfg = S.default_cfg()
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for n in nets:
    n.apply(init_weights); n.to(dev).train()
x = (0.5 * torch.randn(2, 2, 3, 224, 224)).clamp(-1, 1).to(dev)
gt = (torch.rand(2, 32, 32, 32) < 0.1).float().to(dev)
cnt = collections.Counter()
orig = ops.pack_one
def probe(spec, w, kind):
    fr = traceback.extract_stack(limit=6)
    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[:-1][-4:])
    cnt[(kind, spec.cin, spec.cout, spec.k, spec.transposed, isinstance(w, torch.nn.Parameter), ops._CTX.packs is not None, where)] += 1
    return orig(spec, w, kind)
ops.pack_one = probe
for step in range(3):
    cnt.clear()
    for n in nets:
        n.zero_grad(set_to_none=True)
    raw, vol = nets[1](nets[0](x)); m = nets[2](raw, vol)
    (bce(m, gt) + bce(nets[3](m), gt)).backward()
    torch.cuda.synchronize()
    print("step", step, "pack_one calls:", sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:25]:
    print(v, k)
