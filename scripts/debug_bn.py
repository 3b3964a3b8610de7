import torch, sys
import torch.nn.functional as F
sys.path.insert(0, '.')
from swinvox_amd import ops
from swinvox_amd.ops import call, ptr, ACT_LRELU
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (M, C, ld) in ((131072, 9, 12), (131072, 9, 48), (9826, 64, 64), (131072, 1, 1)):
    x = (torch.randn(M, C) * 1.7 + 0.4)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(C)); bn.bias.copy_(0.1 * torch.randn(C))
    bnd = torch.nn.BatchNorm1d(C).double(); bnd.load_state_dict(bn.state_dict())
    bng = torch.nn.BatchNorm1d(C); bng.load_state_dict(bn.state_dict()); bng = bng.to(dev)
    xd64 = x.double().requires_grad_(True)
    z64 = F.leaky_relu(bnd(xd64), 0.2)
    dz = torch.randn(M, C) * 1e-5 + 3e-8
    z64.backward(dz.double())
    xg = torch.zeros(M, ld); xg[:, :C] = x
    xg = xg.to(dev)
    st = ops.BatchNormState(bng, M, True)
    call("sv_bn_stats", ptr(xg), M, C, ld, ptr(st.sums))
    st.finalize()
    zg = torch.zeros(M, ld, device=dev)
    st.apply(xg, ld, zg, ld, ACT_LRELU, 0.2)
    dzg = torch.zeros(M, ld); dzg[:, :C] = dz; dzg = dzg.to(dev)
    dx = torch.zeros(M, ld, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    st.backward(dzg, ld, zg, ld, xg, ld, dx, ld, dg, db, ACT_LRELU, 0.2)
    r = lambda a, b: float((a.cpu().double() - b).abs().max() / (b.abs().max() + 1e-300))
    print(f"M={M} C={C} ld={ld}: z {r(zg[:, :C], z64.detach()):.2e} dx {r(dx[:, :C], xd64.grad):.2e} dgamma {r(dg, bnd.weight.grad):.2e} dbeta {r(db, bnd.bias.grad):.2e}")
