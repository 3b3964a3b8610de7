import torch, math, sys
import torch.nn.functional as F
sys.path.insert(0, '.')
from swinvox_amd import ops
from swinvox_amd.ops import ConvSpec
dev = torch.device('cuda:0')
torch.manual_seed(0)
def cl(x):
    n = x.dim(); return x.permute(0, *range(2, n), 1).contiguous()
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))
for (n, D, cin, cout, cm_in, cm_out) in ((4, 32, 9, 9, 12, 12), (4, 32, 9, 1, 12, 4), (2, 16, 36, 9, 48, 12)):
    x = (torch.randn(n, cin, D, D, D).double() * 1.3 + 0.5).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, 3).double() / math.sqrt(cin * 27)).requires_grad_(True)
    y = F.conv3d(x, w, None, padding=1)
    dy = torch.randn(y.shape).double() * 1e-5
    y.backward(dy)
    sp = ConvSpec.conv3d(cin, cout, 3, 1, 1, cin_mem=cm_in, cout_mem=cm_out)
    M = n * D ** 3
    xm = torch.zeros(M, cm_in); xm[:, :cin] = cl(x.detach().float()).reshape(M, cin)
    dym = torch.zeros(M, cm_out); dym[:, :cout] = cl(dy.float()).reshape(M, cout)
    xd, dyd, wd = xm.to(dev), dym.to(dev), w.detach().float().to(dev)
    out = ops.zeros(M, cm_out, device=dev)
    sp.forward(xd, n, (D, D, D), sp.pack_fwd(wd), out, ldi=cm_in, ldc=cm_out)
    dx = ops.zeros(M, cm_in, device=dev)
    sp.dgrad(dyd, n, (D, D, D), sp.pack_dgrad(wd), dx, lddy=cm_out, lddx=cm_in)
    dw = ops.zeros(cout, cin, 3, 3, 3, device=dev)
    sp.wgrad(dyd, xd, n, (D, D, D), dw, lddy=cm_out, ldx=cm_in)
    dw2 = ops.zeros(cout, cin, 3, 3, 3, device=dev)
    sp.wgrad(dyd, xd, n, (D, D, D), dw2, lddy=cm_out, ldx=cm_in)
    print(f"cin={cin} cout={cout} M={M}: fwd {rel(out[:, :cout], cl(y.detach()).reshape(M, cout)):.2e}  dgrad {rel(dx[:, :cin], cl(x.grad).reshape(M, cin)):.2e}"
          f"  wgrad {rel(dw, w.grad):.2e}  wgrad(run2) {rel(dw2, w.grad):.2e}  run1-vs-run2 {rel(dw, dw2):.2e}")
