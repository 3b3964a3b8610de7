import copy, sys, torch
sys.path.insert(0, '.')
import oracle as O, swinvox_amd as S
from swinvox_amd.models import Decoder, Merger, Refiner
dev = torch.device('cuda:0')
torch.manual_seed(0)
def synth_images(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)
cfg = O.default_cfg()
full = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
for i, n in enumerate(full): O.seeded_weights_(n, seed=100 + i)
O.calibrate_(full, synth_images(2, 2, 1234))
onets = [copy.deepcopy(n).train() for n in full[1:]]
g = torch.Generator().manual_seed(4)
B, V = 2, 2
feat = torch.randn(B, V, 256, 7, 7, generator=g)
gg = torch.Generator().manual_seed(7 + 1000)
gt = (torch.rand(B, 32, 32, 32, generator=gg) < 0.10).float()
pn = [Decoder(S.default_cfg()), Merger(S.default_cfg()), Refiner(S.default_cfg())]
for p, o in zip(pn, onets):
    p.load_state_dict(o.state_dict()); p.to(dev).train()
d64 = [copy.deepcopy(n).double() for n in onets]
bce = torch.nn.functional.binary_cross_entropy_with_logits
def run(nets, f, gtt):
    raw, vol = nets[0](f); mer = nets[1](raw, vol); ref = nets[2](mer)
    (bce(mer, gtt) + bce(ref, gtt)).backward()
    return mer, ref
f1 = feat.clone().requires_grad_(True); run(onets, f1, gt)
f3 = feat.double().requires_grad_(True); m64, r64 = run(d64, f3, gt.double())
f2 = feat.clone().to(dev).requires_grad_(True); mp, rp = run(pn, f2, gt.to(dev))
print("fwd merged rel", float((mp.cpu().double() - m64).abs().max() / m64.abs().max()), "refined", float((rp.cpu().double() - r64).abs().max() / r64.abs().max()))
for p, o, d in zip(pn, onets, d64):
    for (k, a), (_, b), (_, c) in zip(p.named_parameters(), o.named_parameters(), d.named_parameters()):
        if type(p).__name__ != "Refiner": continue
        gh = a.grad.detach().cpu().double(); g32 = b.grad.double(); g64 = c.grad
        sc = float(g64.abs().max()) + 1e-300
        print(f"{type(p).__name__ + '.' + k:30s} scale={sc:9.2e} hip_err={float((gh-g64).abs().max())/sc:9.2e} cpu32_err={float((g32-g64).abs().max())/sc:9.2e}")
# BN statistics of refiner layer2 conv output in the oracle (double): mean/std per channel
x = m64.detach()[:, None]
v16 = d64[2].layer1(x)
y2 = d64[2].layer2[0](v16)
mu, sd = y2.mean((0, 2, 3, 4)), y2.std((0, 2, 3, 4))
print("layer2 conv out: max |mean|/std", float((mu.abs() / sd).max()), "min std", float(sd.min()), "max std", float(sd.max()))
# ---- isolate the weight-gradient contraction of refiner layer2 on the real tensors
from swinvox_amd import ops
from swinvox_amd.ops import ConvSpec
cap = {}
orig = ConvSpec.wgrad
def spy(self, dy, x, n, in_grid, dw, **kw):
    if self.cin == 32 and self.cout == 64 and not self.transposed:
        cap['dy'] = dy.clone(); cap['x'] = x.clone(); cap['n'] = n; cap['grid'] = in_grid; cap['spec'] = self; cap['kw'] = kw
    return orig(self, dy, x, n, in_grid, dw, **kw)
ConvSpec.wgrad = spy
for p in pn:
    p.zero_grad()
f2 = feat.clone().to(dev).requires_grad_(True); run(pn, f2, gt.to(dev))
sp = cap['spec']; n = cap['n']; G = cap['grid']
dyh = cap['dy'].cpu().double().view(n, 17, 17, 17, 64).permute(0, 4, 1, 2, 3)
xh = cap['x'].cpu().double().view(n, 16, 16, 16, 32).permute(0, 4, 1, 2, 3)
w = torch.zeros(64, 32, 4, 4, 4, dtype=torch.float64, requires_grad=True)
torch.nn.functional.conv3d(xh, w, None, padding=2).backward(dyh)
dw = ops.zeros(64, 32, 4, 4, 4, device=dev)
orig(sp, cap['dy'], cap['x'], n, G, dw, **cap['kw'])
e = float((dw.cpu().double() - w.grad).abs().max() / w.grad.abs().max())
print("wgrad kernel on the real layer2 tensors vs host double contraction: rel err", e, " max|dw|", float(w.grad.abs().max()))
print("column sums of dy (should vanish):", float(dyh.sum((0, 2, 3, 4)).abs().max()), " sum|dy|", float(dyh.abs().sum((0,2,3,4)).max()))
g64 = dict(d64[2].named_parameters())['layer2.0.weight'].grad
print("host-double contraction of HIP tensors vs fp64 oracle grad:", float((w.grad - g64).abs().max() / g64.abs().max()))
