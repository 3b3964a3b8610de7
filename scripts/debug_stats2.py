import torch, math, sys
sys.path.insert(0, '.')
from swinvox_amd import ops, hip
from swinvox_amd.ops import ConvSpec, call, ptr
dev = torch.device('cuda:0')
torch.manual_seed(0)
def check(sp, n, grid, cin_mem, ldc, label):
    Min = n * grid[0] * grid[1] * grid[2]
    x = torch.zeros(Min, cin_mem); x[:, :sp.cin] = torch.randn(Min, sp.cin) * 1.5 + 0.7
    wshape = (sp.cin, sp.cout) + sp.k if sp.transposed else (sp.cout, sp.cin) + sp.k
    w = torch.randn(wshape) / math.sqrt(sp.cin * sp.taps)
    b = torch.randn(sp.cout)
    og = sp.out_grid(grid); M = n * og[0] * og[1] * og[2]
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = torch.zeros(M, ldc, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 2 * sp.cout, dtype=torch.float64, device=dev)
    sp.forward(xd, n, grid, sp.pack_fwd(wd), y, ldc=ldc, bias=bd, stats=stats)
    torch.cuda.synchronize()
    yd = y[:, :sp.cout].cpu().double(); st = stats.sum(0).cpu()
    e1 = float(((st[:sp.cout] - yd.sum(0)).abs() / yd.abs().sum(0)).max())
    e2 = float(((st[sp.cout:] - (yd * yd).sum(0)).abs() / (yd * yd).sum(0)).max())
    print(f"{label}: M={M} stats rel err sum={e1:.2e} sumsq={e2:.2e}")
check(ConvSpec.conv3d(9, 9, 3, 1, 1, cin_mem=12, cout_mem=12), 4, (32, 32, 32), 12, 9, "merger 9->9 ldc=9")
check(ConvSpec.conv3d(9, 1, 3, 1, 1, cin_mem=12, cout_mem=4), 4, (32, 32, 32), 12, 1, "merger 9->1")
check(ConvSpec.conv3d(32, 8, 4, 2, 1, transposed=True), 4, (16, 16, 16), 32, 8, "decoder tconv 32->8")
check(ConvSpec.conv3d(64, 32, 4, 2, 1, transposed=True), 4, (8, 8, 8), 64, 32, "decoder tconv 64->32")
check(ConvSpec.conv2d(256, 256, 3, 2, 1), 4, (1, 14, 14), 256, 256, "neck conv s2")
