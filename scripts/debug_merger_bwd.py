import sys, copy, torch
sys.path.insert(0, '.')
import oracle as O, swinvox_amd as S
from swinvox_amd import ops
from swinvox_amd.ops import ACT_LRELU, call, ptr, zeros, empty
from swinvox_amd.models import Merger
from swinvox_amd.models._base import GradStore
from swinvox_amd.models.decoder import VOX, as_channels_last12, raw_view
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = O.default_cfg()
om = O.Merger(cfg); O.seeded_weights_(om, seed=102); om.train()
om64 = copy.deepcopy(om).double()
pm = Merger(S.default_cfg()); pm.load_state_dict(om.state_dict()); pm.to(dev).train()
B, V = 2, 2
g = torch.Generator().manual_seed(4)
raw = torch.randn(B, V, 9, 32, 32, 32, generator=g); vol = torch.randn(B, V, 32, 32, 32, generator=g)
dout = torch.randn(B, 32, 32, 32, generator=g) * 1e-5
# reference (double) with retained intermediates
x = raw.double().reshape(B * V, 9, 32, 32, 32).requires_grad_(True)
w1 = om64.layer1(x); w2 = om64.layer2(w1); w3 = om64.layer3(w2); w4 = om64.layer4(w3)
cat = torch.cat((w1, w2, w3, w4), 1); z5 = om64.layer5(cat); wl = om64.layer6(z5)
for t in (w1, w2, w3, w4, z5, wl): t.retain_grad()
out = (vol.double() * wl.view(B, V, 32, 32, 32).softmax(1)).sum(1)
out.backward(dout.double())
def cl(t): return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
# HIP forward
rawd, vold = raw.to(dev), vol.to(dev)
outp, tape = pm._fwd(rawd, vold, save=True)
print("fwd out", rel(outp, out))
Bq, Vq, x12, volc, catb, ctx14, w5p, y5, st5, z5b, y6, st6, wlb, outb = tape
print("fwd z5", rel(z5b[:, :9], cl(z5)), "wl", rel(wlb, cl(wl)), "cat", rel(catb.view(-1, 4, 12)[:, :, :9].reshape(-1, 36), cl(cat)))
grads = GradStore(list(pm.parameters()))
I, M, sl = B * V, B * V * VOX, 0.2
doutd = dout.to(dev)
dwl = empty(M, 1, like=volc); dvol = empty(B, V, 32, 32, 32, like=volc)
call("sv_merge_views_bwd", ptr(wlb), ptr(volc), ptr(outb), ptr(doutd), ptr(dwl), ptr(dvol), B, V, VOX)
print("dwl", rel(dwl, cl(wl.grad)))
conv6, bn6 = pm.layer6[0], pm.layer6[1]
dy6 = zeros(M, 4, like=volc)
st6.backward(dwl, 1, wlb, 1, y6, 1, dy6, 4, grads[bn6.weight], grads[bn6.bias], ACT_LRELU, sl)
dz5 = zeros(M, 12, like=volc)
pm._s6.dgrad(dy6, I, (32, 32, 32), pm._s6.pack_dgrad(conv6.weight), dz5, lddy=4, lddx=12)
print("dz5", rel(dz5[:, :9], cl(z5.grad)), "sum check hip", float(dz5[:, :9].double().sum()), "ref", float(z5.grad.sum()))
conv5, bn5 = pm.layer5[0], pm.layer5[1]
dy5 = zeros(M, 12, like=volc)
st5.backward(dz5, 12, z5b, 12, y5, 9, dy5, 12, grads[bn5.weight], grads[bn5.bias], ACT_LRELU, sl)
print("dbeta5", rel(grads[bn5.bias], om64.layer5[1].bias.grad), "dgamma5", rel(grads[bn5.weight], om64.layer5[1].weight.grad))
# what would the exact dbeta be from HIP's own dz5 and z5?
dzp = dz5[:, :9].cpu().double() * torch.where(z5b[:, :9].cpu() > 0, 1.0, 0.2).double()
print("dbeta5 from hip dz5,z5 in double on host:", rel(dzp.sum(0), om64.layer5[1].bias.grad))
dzr = cl(z5.grad) * torch.where(cl(z5) > 0, 1.0, 0.2).double()
print("dbeta5 from ref dz5,z5:", rel(dzr.sum(0), om64.layer5[1].bias.grad))
mask_diff = int(((z5b[:, :9].cpu() > 0) != (cl(z5) > 0)).sum())
print("mask differences", mask_diff, "of", M * 9)
