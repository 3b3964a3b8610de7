import torch, math, sys
sys.path.insert(0, '.')
from swinvox_amd import ops, hip
from swinvox_amd.ops import ConvSpec, call, ptr
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (cin, cout, D) in ((32, 64, 16), (1, 32, 32), (64, 128, 8)):
    n = 2
    x = torch.randn(n * D**3, cin) * 1.5 + 0.7
    w = torch.randn(cout, cin, 4, 4, 4) / math.sqrt(cin * 64)
    b = torch.randn(cout)
    sp = ConvSpec.conv3d(cin, cout, 4, 1, 2)
    og = sp.out_grid((D, D, D)); M = n * og[0] * og[1] * og[2]
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = ops.empty(M, cout, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 2 * cout, dtype=torch.float64, device=dev)
    sp.forward(xd, n, (D, D, D), sp.pack_fwd(wd), y, bias=bd, stats=stats)
    torch.cuda.synchronize()
    yd = y.cpu().double()
    st = stats.sum(0).cpu()
    e1 = float(((st[:cout] - yd.sum(0)).abs() / yd.abs().sum(0)).max())
    e2 = float(((st[cout:] - (yd * yd).sum(0)).abs() / (yd * yd).sum(0)).max())
    s2 = torch.zeros(ops.BN_SLOTS, 2 * cout, dtype=torch.float64, device=dev)
    call("sv_bn_stats", ptr(y), M, cout, cout, ptr(s2))
    t = s2.sum(0).cpu()
    f1 = float(((t[:cout] - yd.sum(0)).abs() / yd.abs().sum(0)).max())
    f2 = float(((t[cout:] - (yd * yd).sum(0)).abs() / (yd * yd).sum(0)).max())
    print(f"cin={cin} cout={cout} M={M}: epilogue stats rel err sum={e1:.2e} sumsq={e2:.2e} | bn_stats kernel {f1:.2e} {f2:.2e}")
