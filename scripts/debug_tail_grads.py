import copy, sys, torch
sys.path.insert(0, '.')
import oracle as O, swinvox_amd as S
from swinvox_amd.models import Decoder, Merger, Refiner
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = O.default_cfg()
onets = [O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
for i, n in enumerate(onets):
    O.seeded_weights_(n, seed=101 + i); n.train()
g = torch.Generator().manual_seed(4)
B, V = 2, 2
feat = torch.randn(B, V, 256, 7, 7, generator=g)
gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.1).float()
pn = [Decoder(S.default_cfg()), Merger(S.default_cfg()), Refiner(S.default_cfg())]
for p, o in zip(pn, onets):
    p.load_state_dict(o.state_dict()); p.to(dev).train()
d64 = [copy.deepcopy(n).double() for n in onets]
def run(nets, f, gtt, bce):
    raw, vol = nets[0](f); mer = nets[1](raw, vol); ref = nets[2](mer)
    (bce(mer, gtt) + bce(ref, gtt)).backward()
bce = torch.nn.functional.binary_cross_entropy_with_logits
f1 = feat.clone().requires_grad_(True); run(onets, f1, gt, bce)
f3 = feat.double().requires_grad_(True); run(d64, f3, gt.double(), bce)
f2 = feat.clone().to(dev).requires_grad_(True); run(pn, f2, gt.to(dev), bce)
rows = [("feat", f2.grad, f1.grad, f3.grad)]
for p, o, d in zip(pn, onets, d64):
    for (k, a), (_, b), (_, c) in zip(p.named_parameters(), o.named_parameters(), d.named_parameters()):
        rows.append((type(p).__name__ + "." + k, a.grad, b.grad, c.grad))
for name, gh, g32, g64 in rows:
    gh = gh.detach().cpu().double(); g32 = g32.double()
    sc = float(g64.abs().max()) + 1e-300
    print(f"{name:34s} scale={sc:9.2e} hip_err={float((gh-g64).abs().max())/sc:9.2e} cpu32_err={float((g32-g64).abs().max())/sc:9.2e}")
