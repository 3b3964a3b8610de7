"""Op-level parity of every C-ABI kernel family against a plain PyTorch fp32 CPU reference of the same op."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from swinvox_amd import hip, ops  # noqa: E402
from swinvox_amd.ops import ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, ConvSpec, call, ptr  # noqa: E402

TOL = 2e-4  # relative to max|ref| (fp32 MFMA vs CPU fp32 differ only by summation order)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def cl(x):      # NC... -> N...C contiguous
    n = x.dim()
    return x.permute(0, *range(2, n), 1).contiguous()


def ncf(x):     # N...C -> NC...
    n = x.dim()
    return x.permute(0, n - 1, *range(1, n - 1)).contiguous()


_KEEP = []


def keep(t):
    """Hold a reference to a device temporary until the test ends: a tensor freed before the (asynchronous) kernel
    launch can be handed to the next allocation by the caching allocator and overwritten."""
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _f32():
    ops.set_math("f32")
    yield
    ops.set_math("f32")
    _KEEP.clear()


def test_library_loads_on_gpu(dev):
    assert hip.load().sv_version() >= 1


# ------------------------------------------------------------------------------------------------ contraction engine
@pytest.mark.parametrize("M,K,N", [(200, 96, 288), (49, 768, 100), (1000, 384, 96), (3, 8192, 2048)])
def test_linear_fwd_bwd(dev, M, K, N):
    g = torch.Generator().manual_seed(M + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    xd, wd, bd, rd = (t.to(dev) for t in (x, w, b, res))
    sp = ConvSpec.linear(K, N)
    out, pre = ops.empty(M, N, device=dev), ops.empty(M, N, device=dev)
    ops.linear_fwd(xd, M, sp, wd, out, bias=bd, act=ACT_GELU, pre_act=pre, residual=rd, ldr=N)
    ref_pre = x @ w.t() + b
    assert rel(pre, ref_pre) < TOL
    assert rel(out, res + F.gelu(ref_pre)) < TOL
    # data / weight / bias gradients
    dy = torch.randn(M, N, generator=g)
    dyd = dy.to(dev)
    dx = ops.empty(M, K, device=dev)
    ops.linear_dgrad(dyd, M, sp, sp.pack_dgrad(wd), dx)
    assert rel(dx, dy @ w) < TOL
    dw, db = ops.zeros(N, K, device=dev), ops.zeros(N, device=dev)
    ops.linear_wgrad(dyd, xd, M, sp, dw, db)
    assert rel(dw, dy.t() @ x) < TOL
    assert rel(db, dy.sum(0)) < TOL
    # activation-gradient epilogue
    dx2 = ops.empty(M, K, device=dev)
    src = torch.randn(M, K, generator=g)
    ops.linear_dgrad(dyd, M, sp, sp.pack_dgrad(wd), dx2, act_grad_src=src.to(dev), act_grad_kind=ACT_GELU)
    s = src.clone().requires_grad_(True)
    F.gelu(s).backward(dy @ w)
    assert rel(dx2, s.grad) < TOL


def test_linear_bf16_math(dev):
    ops.set_math("bf16")
    g = torch.Generator().manual_seed(5)
    M, K, N = 300, 192, 160
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    out = ops.empty(M, N, device=dev)
    ops.linear_fwd(x.to(dev), M, ConvSpec.linear(K, N), w.to(dev), out)
    ref = x.bfloat16().float() @ w.bfloat16().float().t()
    assert rel(out, ref) < 2e-3
    dw = ops.zeros(N, K, device=dev)
    dy = torch.randn(M, N, generator=g)
    ops.linear_wgrad(dy.to(dev), x.to(dev), M, ConvSpec.linear(K, N), dw)
    assert rel(dw, dy.bfloat16().float().t() @ x.bfloat16().float()) < 2e-3


def _is_wide(sp, x, n, grid, wp, out, **epi):
    og = sp.out_grid(grid)
    gm = sp._geom(n, grid, og, sp.cin_mem, sp.cout, sp.cin_mem)
    e = ops._epilogue(epi.pop("ldc", sp.cout), **epi)
    return hip.load().sv_conv_gather_is_wide(ptr(x), ptr(wp), ptr(out), C.byref(gm), C.byref(e), hip.MATH_BF16, hip.BF16)


@pytest.mark.parametrize("mode", ["plain", "bias_gelu_preact", "residual_scale", "act_grad", "stats", "column_slice"])
def test_gemm_wide_kernel(dev, mode):
    """csrc/igemm.hip gemm_wide_kernel (256 x 128 tile, LDS-DMA ring; the deep Linear / 1x1 layers in bf16 storage) against torch on
    bf16-rounded operands, with ragged rows (M % 256 != 0), ragged columns (Co % 128 != 0) and every epilogue feature of the engine."""
    ops.set_math("bf16"); ops.set_storage("bf16")
    g = torch.Generator().manual_seed(77)
    M, K, N = 256 * 130 + 37, (1024 if mode == "stats" else 384), 392   # (the kernel takes BatchNorm producers from K = 1024 on)
    bf = lambda t: t.bfloat16().float()
    x, w = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g)
    acc = x @ w.t()
    sp = ConvSpec.linear(K, N)
    xd, wp = x.to(dev).bfloat16(), sp.pack_fwd(w.to(dev))
    bd = b.to(dev)
    ldc, off = (N, 0) if mode != "column_slice" else (N + 24, 16)
    out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=dev)
    epi, ref, extra = {}, acc, None
    if mode == "bias_gelu_preact":
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        epi = dict(bias=bd, act=ACT_GELU, pre_act=pre)
        ref = F.gelu(acc + b)
        extra = (pre, acc + b)
    elif mode == "residual_scale":
        res = bf(torch.randn(M, N, generator=g))
        scale = torch.rand(M // 49 + 1, generator=g)
        epi = dict(bias=bd, residual=keep(res.to(dev).bfloat16()), ldr=N, row_scale=keep(scale.to(dev)), rows_per_scale=49)
        ref = res + scale[torch.arange(M) // 49, None] * (acc + b)
    elif mode == "act_grad":
        src = bf(torch.randn(M, N, generator=g))
        epi = dict(act_grad_src=keep(src.to(dev).bfloat16()), act_grad_kind=ACT_RELU)
        ref = acc * (src > 0)
    elif mode == "stats":
        stats = torch.zeros(ops.BN_SLOTS, 2 * N, dtype=torch.float64, device=dev)
        epi = dict(bias=bd, stats=stats)
        ref = acc + b
    assert _is_wide(sp, xd, M, (1, 1, 1), wp, out[:, off:], ldc=ldc, **epi) == 1
    sp.forward(xd, M, (1, 1, 1), wp, out[:, off:], ldc=ldc, **epi)
    got = out[:, off:off + N].float()
    assert rel(got, ref) < 6e-3 and float((got.cpu() - ref).abs().mean()) < 2e-3 * float(ref.abs().mean() + 1)
    if mode == "column_slice":
        assert float(out[:, :off].abs().max()) == 0 and float(out[:, off + N:].abs().max()) == 0
    if extra is not None:
        assert rel(extra[0].float(), extra[1]) < 6e-3
    if mode == "stats":
        st, o = stats.sum(0).cpu(), got.cpu().double()
        assert rel(st[:N], o.sum(0)) < 1e-4 and rel(st[N:], (o * o).sum(0)) < 1e-4
    # K not a multiple of the 64-deep slice, K < 192 and small problems stay on the 128-wide kernels
    sp2 = ConvSpec.linear(K + 32, N)
    x2 = torch.zeros(M, K + 32, dtype=torch.bfloat16, device=dev)
    assert _is_wide(sp2, x2, M, (1, 1, 1), sp2.pack_fwd(torch.zeros(N, K + 32, device=dev)), out[:, off:], ldc=ldc) == 0
    assert _is_wide(sp, xd[:4096], 4096, (1, 1, 1), wp, out[:4096, off:], ldc=ldc) == 0
    sp3 = ConvSpec.linear(128, 264)                              # fewer than three 64-deep slices
    x3 = torch.zeros(M, 128, dtype=torch.bfloat16, device=dev)
    assert _is_wide(sp3, x3, M, (1, 1, 1), sp3.pack_fwd(torch.zeros(264, 128, device=dev)), out[:, off:], ldc=ldc) == 0


@pytest.mark.parametrize("K,N,bias", [(392, 264, True), (512, 384, False), (384, 512, True)])
def test_wgrad_wide_kernel(dev, K, N, bias):
    """csrc/igemm.hip wgrad_wide_kernel (dW = dY^T X of the dense layers in bf16 storage: 256 x 128 tiles, both tile orientations, ragged
    tiles) against torch on bf16-rounded operands; dW and the bias gradient are accumulated into what is already there."""
    ops.set_math("bf16"); ops.set_storage("bf16")
    g = torch.Generator().manual_seed(K + N)
    M = 64 * 331
    bf = lambda t: t.bfloat16().float()
    x, dy = bf(torch.randn(M, K, generator=g)), bf(torch.randn(M, N, generator=g))
    sp = ConvSpec.linear(K, N)
    xd, dyd = x.to(dev).bfloat16(), dy.to(dev).bfloat16()
    gm = sp._geom(M, (1, 1, 1), (1, 1, 1), K, N, K)
    assert hip.load().sv_conv_wgrad_is_wide(ptr(dyd), N, ptr(xd), C.byref(gm), K, hip.MATH_BF16, hip.BF16) == 1
    base = torch.randn(N, K, generator=g)
    dw = base.clone().to(dev)
    db = torch.ones(N, device=dev) if bias else None
    sp.wgrad(dyd, xd, M, (1, 1, 1), dw, db=db, async_ok=False)
    ref = dy.t() @ x
    assert rel(dw.cpu() - base, ref) < 2e-3
    if bias:
        assert rel(db.cpu() - 1.0, dy.sum(0)) < 2e-3
    gm2 = sp._geom(M - 64 + 8, (1, 1, 1), (1, 1, 1), K, N, K)     # rows not a multiple of the 64-row slice: the 128-wide kernel
    assert hip.load().sv_conv_wgrad_is_wide(ptr(dyd), N, ptr(xd), C.byref(gm2), K, hip.MATH_BF16, hip.BF16) == 0


CONV2D = [  # cin, cout, k, s, p, H
    (64, 64, 3, 1, 1, 14), (256, 256, 3, 2, 1, 14), (64, 128, 1, 2, 0, 14), (3, 64, 7, 2, 3, 32), (3, 96, 4, 4, 0, 32), (512, 256, 3, 1, 1, 7)]


@pytest.mark.parametrize("cin,cout,k,s,p,H", CONV2D)
def test_conv2d_fwd_dgrad_wgrad(dev, cin, cout, k, s, p, H):
    g = torch.Generator().manual_seed(cin + cout + k)
    n = 3
    x = torch.randn(n, cin, H, H, generator=g, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    sp = ConvSpec.conv2d(cin, cout, k, s, p)
    xd, wd = cl(x.detach()).to(dev), w.detach().to(dev)
    og = sp.out_grid((1, H, H))
    M = n * og[1] * og[2]
    out = ops.empty(M, cout, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 2 * cout, dtype=torch.float64, device=dev)
    sp.forward(xd, n, (1, H, H), sp.pack_fwd(wd), out, bias=b.to(dev), stats=stats)
    ref = cl(y.detach()).reshape(M, cout)
    assert rel(out, ref) < TOL
    st = stats.sum(0)
    assert rel(st[:cout], ref.sum(0)) < 1e-3 and rel(st[cout:], (ref * ref).sum(0)) < 1e-3
    dyd = cl(dy).reshape(M, cout).to(dev)
    if s <= 2:   # the stride-4 patch embedding never needs a data gradient (its input is the image)
        dx = ops.empty(n * H * H, cin, device=dev)
        sp.dgrad(dyd, n, (1, H, H), sp.pack_dgrad(wd), dx)
        assert rel(dx, cl(x.grad).reshape(-1, cin)) < TOL
    dw = ops.zeros(cout, cin, k, k, device=dev)
    sp.wgrad(dyd, xd, n, (1, H, H), dw)
    assert rel(dw, w.grad) < TOL


@pytest.mark.parametrize("cin,cout,k,s,p,H,relu", [(64, 64, 3, 1, 1, 183, False), (128, 136, 3, 2, 1, 270, True), (256, 264, 1, 2, 0, 230, False)])
def test_conv2d_bf16_storage_large_maps(dev, cin, cout, k, s, p, H, relu):
    """Forward convolutions at the row counts of the bench shapes (tens of thousands of rows, hundreds of tiles, a ragged last tile) in
    bf16 storage: 3x3 / stride 1 and 2 and a strided 1x1 with bias (+ ReLU) and the BatchNorm statistics epilogue, against torch fp32 on
    bf16-representable operands.  The statistics must equal the column sums of the STORED output exactly up to summation order - a tile
    computed twice would leave the output intact and show up here."""
    g = torch.Generator().manual_seed(cin + cout + H)
    n = 2
    x = torch.randn(n, cin, H, H, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).bfloat16().float()
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    if relu:
        ref = ref.relu()
    ref = cl(ref)
    sp = ConvSpec.conv2d(cin, cout, k, s, p)
    og = sp.out_grid((1, H, H))
    M = n * og[1] * og[2]
    assert M % 256 != 0
    ref = ref.reshape(M, cout)
    ops.set_math("bf16")
    ops.set_storage("bf16")
    try:
        xd = cl(x).to(dev).bfloat16()
        wp = sp.pack_fwd(w.to(dev))
        for _ in range(2):
            out = ops.empty(M, cout, device=dev)
            stats = torch.zeros(ops.BN_SLOTS, 2 * cout, dtype=torch.float64, device=dev)
            epi = dict(bias=b.to(dev), stats=stats)
            if relu:
                epi["act"] = ACT_RELU
            sp.forward(xd, n, (1, H, H), wp, out, **epi)
            torch.cuda.synchronize()
            assert rel(out.float(), ref) < 6e-3            # bf16 rounding of the stored output
            st, o = stats.sum(0), out.float().cpu().double()
            assert rel(st[:cout], o.sum(0)) < 1e-4 and rel(st[cout:], (o * o).sum(0)) < 1e-4      # fp32 partial sums per tile; one tile twice = 2e-3
    finally:
        ops.set_math("f32")


CONV3D = [  # cin, cout, k, s, p, transposed, D
    (1, 32, 4, 1, 2, False, 8), (32, 64, 4, 1, 2, False, 8), (128, 64, 4, 2, 1, True, 4), (32, 8, 4, 2, 1, True, 8),
    (256, 128, (6, 4, 4), 2, (2, 1, 1), True, 2), (32, 1, 4, 2, 1, True, 8)]


@pytest.mark.parametrize("cin,cout,k,s,p,tr,D", CONV3D)
def test_conv3d_family(dev, cin, cout, k, s, p, tr, D):
    g = torch.Generator().manual_seed(cin * 3 + cout)
    n = 2
    x = torch.randn(n, cin, D, D, D, generator=g, requires_grad=True)
    kk = (k, k, k) if isinstance(k, int) else k
    wshape = (cin, cout) + kk if tr else (cout, cin) + kk
    w = (torch.randn(wshape, generator=g) / math.sqrt(cin * kk[0] * kk[1] * kk[2])).requires_grad_(True)
    y = F.conv_transpose3d(x, w, None, stride=s, padding=p) if tr else F.conv3d(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    cm = 4 if cout == 1 else None
    sp = ConvSpec.conv3d(cin, cout, k, s, p, transposed=tr, cout_mem=cm)
    grid = (D, D, D)
    og = sp.out_grid(grid)
    assert tuple(y.shape[2:]) == og
    M = n * og[0] * og[1] * og[2]
    xd, wd = cl(x.detach()).to(dev), w.detach().to(dev)
    out = ops.empty(M, cout, device=dev)
    sp.forward(xd, n, grid, sp.pack_fwd(wd), out, ldc=cout)
    assert rel(out, cl(y.detach()).reshape(M, cout)) < TOL
    cmem = sp.cout_mem
    dyd = torch.zeros(M, cmem)
    dyd[:, :cout] = cl(dy).reshape(M, cout)
    dyd = dyd.to(dev)
    dx = ops.empty(n * D ** 3, cin, device=dev)
    sp.dgrad(dyd, n, grid, sp.pack_dgrad(wd), dx, lddy=cmem)
    assert rel(dx, cl(x.grad).reshape(-1, cin)) < TOL
    dw = ops.zeros(*wshape, device=dev)
    sp.wgrad(dyd, xd, n, grid, dw, lddy=cmem)
    assert rel(dw, w.grad) < TOL


@pytest.mark.parametrize("n,D,bias", [(2, 8, True), (3, 16, False)])
def test_tconv4s2_halo_brick(dev, n, D, bias):
    """csrc/stencil.hip tconv4s2_fwd_kernel (decoder layer4: ConvTranspose3d 32 -> 8, k4 s2 p1 in bf16 storage) vs torch on bf16-rounded
    operands; ConvSpec.forward must route to it, write the same rows as the generic engine and the BatchNorm sums of the stored values."""
    ops.set_math("bf16"); ops.set_storage("bf16")
    g = torch.Generator().manual_seed(40 + D)
    bf = lambda t: t.bfloat16().float()
    x = bf(torch.randn(n, 32, D, D, D, generator=g))
    w = bf(torch.randn(32, 8, 4, 4, 4, generator=g) / math.sqrt(32 * 8))
    b = torch.randn(8, generator=g) if bias else None
    y = F.conv_transpose3d(x, w, b, stride=2, padding=1)
    sp = ConvSpec.conv3d(32, 8, 4, 2, 1, transposed=True)
    grid, M = (D, D, D), n * (2 * D) ** 3
    assert sp._halo_tconv(grid, None, None, {"bias": None, "stats": None})
    xd, wd = cl(x).to(dev).bfloat16().contiguous(), w.to(dev)
    bd = b.to(dev) if bias else None
    wp = sp.pack_fwd(wd)
    out = torch.empty(M, 8, dtype=torch.bfloat16, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 16, dtype=torch.float64, device=dev)
    names = []
    orig = ops.call
    ops.call = lambda name, *a, **k: (names.append(name), orig(name, *a, **k))[1]
    try:
        sp.forward(xd, n, grid, wp, out, bias=bd, stats=stats)
    finally:
        ops.call = orig
    assert names == ["sv_tconv4s2_fwd"]
    ref = cl(y).reshape(M, 8)
    assert rel(out.float(), ref) < 6e-3                        # one bf16 rounding of the stored value
    st, o = stats.sum(0).cpu(), out.float().cpu().double()
    assert rel(st[:8], o.sum(0)) < 1e-5 and rel(st[8:], (o * o).sum(0)) < 1e-5
    # the generic engine on the same operands (what the layer ran through before): identical up to fp32 summation order
    out2 = torch.empty(M, 8, dtype=torch.bfloat16, device=dev)
    e = ops._epilogue(8, bias=bd)
    gm = sp._geom(n, grid, sp.out_grid(grid), 32, 8, 32)
    call("sv_tconv_gather", ptr(xd), ptr(wp), ptr(out2), C.byref(gm), C.byref(e), hip.MATH_BF16)
    assert rel(out.float(), out2.float()) < 8e-3 and float((out.float() - out2.float()).abs().mean()) < 1e-4
    with pytest.raises(RuntimeError, match="multiple of the 4x4x8 brick"):
        call("sv_tconv4s2_fwd", ptr(xd), ptr(wp), None, ptr(out), None, n, D, D, 4, 32, 8)


def test_padded_channel_conv3d_like_merger(dev):
    """9 real channels stored 12-wide; output written into a column slice of a 48-wide buffer."""
    g = torch.Generator().manual_seed(9)
    n, D = 2, 8
    x = torch.randn(n, 9, D, D, D, generator=g, requires_grad=True)
    w = (torch.randn(9, 9, 3, 3, 3, generator=g) / 15).requires_grad_(True)
    y = F.conv3d(x, w, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    sp = ConvSpec.conv3d(9, 9, 3, 1, 1, cin_mem=12, cout_mem=12)
    M = n * D ** 3
    x12 = torch.zeros(M, 12); x12[:, :9] = cl(x.detach()).reshape(M, 9)
    cat = ops.zeros(M, 48, device=dev)
    wd = w.detach().to(dev)
    sp.forward(x12.to(dev), n, (D, D, D), sp.pack_fwd(wd), cat[:, 12:], ldi=12, ldc=48)
    assert rel(cat[:, 12:21], cl(y.detach()).reshape(M, 9)) < TOL
    assert float(cat[:, :12].abs().max()) == 0 and float(cat[:, 21:].abs().max()) == 0
    dy12 = torch.zeros(M, 12); dy12[:, :9] = cl(dy).reshape(M, 9)
    dx = ops.zeros(M, 12, device=dev)
    sp.dgrad(dy12.to(dev), n, (D, D, D), sp.pack_dgrad(wd), dx, lddy=12, lddx=12)
    assert rel(dx[:, :9], cl(x.grad).reshape(M, 9)) < TOL
    dw = ops.zeros(9, 9, 3, 3, 3, device=dev)
    sp.wgrad(dy12.to(dev), x12.to(dev), n, (D, D, D), dw, lddy=12, ldx=12)
    assert rel(dw, w.grad) < TOL


# ------------------------------------------------------------------------------------------------ normalisation
@pytest.mark.parametrize("rows,Cd", [(37, 96), (100, 768), (10, 3072)])
def test_layernorm_fwd_bwd(dev, rows, Cd):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, Cd, generator=g, requires_grad=True)
    gm, bt = (1 + 0.1 * torch.randn(Cd, generator=g)).requires_grad_(True), (0.1 * torch.randn(Cd, generator=g)).requires_grad_(True)
    y = F.layer_norm(x, (Cd,), gm, bt)
    dy = torch.randn(rows, Cd, generator=g)
    y.backward(dy)
    xd, gd, bd = x.detach().to(dev), gm.detach().to(dev), bt.detach().to(dev)
    out, mean, rstd = ops.layernorm_fwd(xd, gd, bd, rows, Cd)
    assert rel(out, y) < TOL
    dx, dg, db = ops.empty(rows, Cd, device=dev), ops.zeros(Cd, device=dev), ops.zeros(Cd, device=dev)
    ops.layernorm_bwd(dy.to(dev), xd, gd, mean, rstd, dx, dg, db, rows, Cd)
    assert rel(dx, x.grad) < TOL and rel(dg, gm.grad) < TOL and rel(db, bt.grad) < TOL
    # accumulate variant
    base = torch.randn(rows, Cd, generator=g)
    dx2 = base.clone().to(dev)
    ops.layernorm_bwd(dy.to(dev), xd, gd, mean, rstd, dx2, dg, db, rows, Cd, accumulate_dx=True)
    assert rel(dx2, x.grad + base) < TOL


def test_patch_merge_layernorm(dev):
    g = torch.Generator().manual_seed(3)
    I, H, C0 = 2, 8, 32
    x = torch.randn(I, H, H, C0, generator=g, requires_grad=True)
    gm, bt = (1 + 0.1 * torch.randn(4 * C0, generator=g)).requires_grad_(True), torch.zeros(4 * C0, requires_grad=True)
    xm = x.view(I, H // 2, 2, H // 2, 2, C0).permute(0, 1, 3, 4, 2, 5).flatten(3)
    y = F.layer_norm(xm, (4 * C0,), gm, bt)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    rows = I * (H // 2) ** 2
    xd, gd, bd = x.detach().to(dev), gm.detach().to(dev), bt.detach().to(dev)
    out, mean, rstd = ops.layernorm_fwd(xd, gd, bd, rows, 4 * C0, merge_hw=(H, H))
    assert rel(out, y.reshape(rows, -1)) < TOL
    dx, dg, db = ops.empty(I * H * H, C0, device=dev), ops.zeros(4 * C0, device=dev), ops.zeros(4 * C0, device=dev)
    ops.layernorm_bwd(dy.reshape(rows, -1).to(dev), xd, gd, mean, rstd, dx, dg, db, rows, 4 * C0, merge_hw=(H, H))
    assert rel(dx, x.grad.reshape(-1, C0)) < TOL and rel(dg, gm.grad) < TOL


@pytest.mark.parametrize("I,Cc,H", [(3, 96, 14), (37, 192, 7)])   # 37 images: the backward splits the images over several slices (atomics)
def test_ln_image(dev, I, Cc, H):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(I, H, H, Cc, generator=g, requires_grad=True)                 # NHWC data
    w = (1 + 0.1 * torch.randn(Cc, H, H, generator=g)).requires_grad_(True)       # reference [C,H,W] affine
    b = (0.1 * torch.randn(Cc, H, H, generator=g)).requires_grad_(True)
    y = F.layer_norm(x.permute(0, 3, 1, 2), (Cc, H, H), w, b)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    L = Cc * H * H
    wt, bt = w.detach().permute(1, 2, 0).reshape(L).contiguous().to(dev), b.detach().permute(1, 2, 0).reshape(L).contiguous().to(dev)
    xd = x.detach().reshape(I, L).to(dev)
    out, mr = ops.empty(I, L, device=dev), ops.empty(2 * I, device=dev)
    ws = ops.empty(int(hip.load().sv_ln_image_workspace_floats(I, L)), device=dev)
    call("sv_ln_image_fwd", ptr(xd), ptr(wt), ptr(bt), ptr(out), ptr(mr), ptr(ws), I, L, 1e-5, 0.0, 0, None)
    assert rel(out, y.permute(0, 2, 3, 1).reshape(I, L)) < TOL
    dx, dw, db = ops.empty(I, L, device=dev), ops.zeros(L, device=dev), ops.zeros(L, device=dev)
    sums = torch.empty(2 * I, dtype=torch.float64, device=dev)
    dyd = dy.permute(0, 2, 3, 1).reshape(I, L).contiguous().to(dev)
    call("sv_ln_image_bwd", ptr(dyd), ptr(xd), ptr(wt), ptr(mr), ptr(dx), ptr(dw), ptr(db), ptr(sums), I, L, 0.0, 0, None)
    assert rel(dx, x.grad.reshape(I, L)) < TOL
    assert rel(dw, w.grad.permute(1, 2, 0).reshape(L)) < TOL and rel(db, b.grad.permute(1, 2, 0).reshape(L)) < TOL


@pytest.mark.parametrize("act,M,Cc", [(ACT_RELU, 500, 40), (ACT_LRELU, 500, 40), (ACT_NONE, 500, 40),
                                      # C % 256 == 0 with a residual: the activation mask travels as sign words (sv_scale_shift_act_signs /
                                      # sv_bn_bwd_signs), the masked gradient through dres; odd row counts exercise the two-row loop's tail
                                      (ACT_RELU, 1031, 256), (ACT_LRELU, 517, 512), (ACT_RELU, 4099, 1024)])
def test_batchnorm_train_fwd_bwd(dev, act, M, Cc):
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(M, Cc, generator=g) * 2 + 0.5).requires_grad_(True)
    res = torch.randn(M, Cc, generator=g).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(Cc, generator=g)); bn.bias.copy_(0.1 * torch.randn(Cc, generator=g))
    bn_d = torch.nn.BatchNorm1d(Cc)
    bn_d.load_state_dict(bn.state_dict())
    bn_d = bn_d.to(dev)
    fn = {ACT_RELU: F.relu, ACT_LRELU: lambda t: F.leaky_relu(t, 0.2), ACT_NONE: lambda t: t}[act]
    z = fn(bn(x) + res)
    dz = torch.randn(M, Cc, generator=g)
    z.backward(dz)
    st = ops.BatchNormState(bn_d, M, True)
    xd = x.detach().to(dev)
    call("sv_bn_stats", ptr(xd), M, Cc, Cc, ptr(st.sums))
    st.finalize()
    zd = ops.empty(M, Cc, device=dev)
    st.apply(xd, Cc, zd, Cc, act, 0.2, res.detach().to(dev), Cc)
    assert (st.signs is not None) == (Cc % 256 == 0 and act != ACT_NONE)
    assert rel(zd, z) < TOL
    assert rel(bn_d.running_mean, bn.running_mean) < 1e-4 and rel(bn_d.running_var, bn.running_var) < 1e-4
    dx, dres = ops.empty(M, Cc, device=dev), ops.empty(M, Cc, device=dev)
    dg, db = ops.zeros(Cc, device=dev), ops.zeros(Cc, device=dev)
    st.backward(dz.to(dev), Cc, zd, Cc, xd, Cc, dx, Cc, dg, db, act, 0.2, dres, Cc)
    assert rel(dx, x.grad) < 5e-4 and rel(dg, bn.weight.grad) < 5e-4 and rel(db, bn.bias.grad) < 5e-4
    assert rel(dres, res.grad) < 1e-6          # the residual branch's gradient = dz * act'(z)


# ------------------------------------------------------------------------------------------------ attention
def _ref_window_attention(qkv, table, I, H, C, heads, shift):
    import oracle as O
    from oracle.model import rel_pos_index, shift_attn_mask
    x = qkv.view(I, H, H, 3 * C)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    ws = 7
    xw = x.view(I, H // ws, ws, H // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 49, 3, heads, 32).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * 32 ** -0.5, xw[1], xw[2]
    a = q @ k.transpose(-2, -1) + table[rel_pos_index(7).reshape(-1)].view(49, 49, heads).permute(2, 0, 1)[None]
    if shift:
        m = shift_attn_mask(H, H, 7, shift)
        a = (a.view(I, -1, heads, 49, 49) + m[None, :, None]).view(-1, heads, 49, 49)
    o = (a.softmax(-1) @ v).transpose(1, 2).reshape(-1, 49, C)
    o = o.view(I, H // ws, H // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(I, H, H, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(-1, C)


@pytest.mark.parametrize("H,heads,shift", [(14, 3, 0), (14, 3, 3), (28, 6, 3), (7, 24, 0),
                                           # Swin-B head counts (BASELINE configuration 5: heads 4 / 8 / 16 / 32 at 56 / 28 / 14 / 7)
                                           (56, 4, 3), (28, 8, 0), (28, 8, 3), (14, 16, 3), (7, 32, 0)])
def test_window_attention(dev, H, heads, shift):
    g = torch.Generator().manual_seed(H + heads + shift)
    I, Cd = 2, heads * 32
    qkv = torch.randn(I * H * H, 3 * Cd, generator=g, requires_grad=True)
    table = (0.5 * torch.randn(169, heads, generator=g)).requires_grad_(True)
    ref = _ref_window_attention(qkv, table, I, H, Cd, heads, shift)
    do = torch.randn(ref.shape, generator=g)
    ref.backward(do)
    qd, td = qkv.detach().to(dev), table.detach().to(dev)
    out = ops.empty(I * H * H, Cd, device=dev)
    call("sv_window_attention_fwd", ptr(qd), ptr(td), ptr(out), I, H, H, Cd, heads, shift, hip.MATH_F32)
    assert rel(out, ref) < TOL
    out16 = ops.empty(I * H * H, Cd, device=dev)
    call("sv_window_attention_fwd", ptr(qd), ptr(td), ptr(out16), I, H, H, Cd, heads, shift, hip.MATH_BF16)
    assert rel(out16, ref) < 2e-2
    # fp8 (OCP e4m3, per-tile scales) operands for QK^T and PV - BASELINE configuration 5.  Tolerance: e4m3 keeps 3 mantissa bits
    # (relative step 2^-3 .. 2^-4 per operand); over a 32-long / 49-long dot product the error averages down to a few percent of max|out|
    out8 = ops.empty(I * H * H, Cd, device=dev)
    call("sv_window_attention_fwd", ptr(qd), ptr(td), ptr(out8), I, H, H, Cd, heads, shift, hip.MATH_FP8)
    # (measured on these N(0, 1) operands, whose softmax is far sharper than the model's: 0.10 of max|out| worst element, 0.03 on average)
    assert bool(torch.isfinite(out8).all()) and rel(out8, ref) < 0.15, rel(out8, ref)
    assert float((out8.cpu() - ref.detach()).abs().mean() / ref.detach().abs().mean()) < 8e-2
    dqkv, dt = ops.empty(I * H * H, 3 * Cd, device=dev), ops.zeros(169, heads, device=dev)
    dod = do.to(dev)
    call("sv_window_attention_bwd", ptr(qd), ptr(td), ptr(dod), ptr(dqkv), ptr(dt), None, I, H, H, Cd, heads, shift, hip.MATH_F32)
    assert rel(dqkv, qkv.grad) < TOL and rel(dt, table.grad) < TOL
    dqkv16, dt16 = ops.empty(I * H * H, 3 * Cd, device=dev), ops.zeros(169, heads, device=dev)
    ws16 = ops.fzeros(int(hip.load().sv_window_attention_bwd_workspace_floats(heads)), device=dev)   # slot-spread dtable partial sums
    call("sv_window_attention_bwd", ptr(qd), ptr(td), ptr(dod), ptr(dqkv16), ptr(dt16), ptr(ws16), I, H, H, Cd, heads, shift, hip.MATH_BF16)
    assert rel(dqkv16, qkv.grad) < 3e-2 and rel(dt16, table.grad) < 3e-2


@pytest.mark.parametrize("V,P", [(1, 9), (3, 9), (8, 9), (3, 49), (24, 49), (32, 9)])   # P = 49: ATT_SPATIAL_DOWNSAMPLE_RATIO = 1 (six feature chunks)
def test_cross_view_attention_core(dev, V, P):
    g = torch.Generator().manual_seed(V + P)
    B, R, heads = 2, 128, 4
    qkv = torch.randn(B * V * P, 3 * R, generator=g, requires_grad=True)
    t = qkv.view(B, V, P, 3, heads, 32).permute(3, 0, 1, 4, 2, 5).reshape(3, B, V, heads, P * 32)
    s = torch.einsum("bihf,bjhf->bhij", t[0], t[1]) / math.sqrt(32 * V)
    o = torch.einsum("bhij,bjhf->bihf", s.softmax(-1), t[2]).reshape(B, V, heads, P, 32).permute(0, 1, 3, 2, 4).reshape(B * V * P, R)
    do = torch.randn(o.shape, generator=g)
    o.backward(do)
    qd = qkv.detach().to(dev)
    out = ops.empty(B * V * P, R, device=dev)
    call("sv_cross_view_attention_fwd", ptr(qd), ptr(out), B, V, P, R, heads)
    assert rel(out, o) < TOL
    dq = ops.empty(B * V * P, 3 * R, device=dev)
    call("sv_cross_view_attention_bwd", ptr(qd), ptr(keep(do.to(dev))), ptr(dq), B, V, P, R, heads)
    assert rel(dq, qkv.grad) < TOL


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_batched_weight_packs_equal_the_single_packs(dev, storage):
    """ops.PackCache (every pack of a module from ONE sv_pack_weights launch - the data-gradient packs of Linear / 1x1 layers go through its
    LDS-tiled transpose) against sv_pack_weight, pack by pack, bit for bit: linears with and without 64-multiples, convolutions, transposed
    convolutions, zero-padded channel counts."""
    g = torch.Generator().manual_seed(21)
    specs = [ConvSpec.linear(2048, 8192), ConvSpec.linear(384, 1536), ConvSpec.linear(96, 288), ConvSpec.linear(1024, 256), ConvSpec.linear(40, 72),
             ConvSpec.conv2d(64, 64, 3, 1, 1), ConvSpec.conv2d(256, 128, 1, 2, 0), ConvSpec.conv3d(9, 9, 3, 1, 1, cin_mem=12, cout_mem=12),
             ConvSpec.conv3d(32, 8, 4, 2, 1, transposed=True), ConvSpec.linear(64, 256)]
    ops.set_math("bf16" if storage == "bf16" else "f32")
    if storage == "bf16":
        ops.set_storage("bf16")
    try:
        params = []
        for sp in specs:
            shape = (sp.cin, sp.cout) + sp.k if sp.transposed else (sp.cout, sp.cin) + sp.k
            params.append(torch.nn.Parameter(torch.randn(*shape, generator=g).to(dev)))
        cache = ops.PackCache()
        ops.set_pack_cache(cache)
        try:
            first = [(sp.pack_fwd(w), sp.pack_dgrad(w)) for sp, w in zip(specs, params)]       # registers the requests (single packs)
            cache.refresh()                                                                     # one batched launch
            second = [(sp.pack_fwd(w), sp.pack_dgrad(w)) for sp, w in zip(specs, params)]      # views of the batched buffer
        finally:
            ops.set_pack_cache(None)
        torch.cuda.synchronize()
        for sp, (f1, d1), (f2, d2) in zip(specs, first, second):
            assert torch.equal(f1.reshape(-1).float(), f2.reshape(-1).float()), ("fwd", sp)
            assert torch.equal(d1.reshape(-1).float(), d2.reshape(-1).float()), ("dgrad", sp)
    finally:
        ops.set_math("f32")


# ------------------------------------------------------------------------------------------------ glue kernels
def test_transpose_and_pools(dev):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 70, 45, generator=g)
    out = ops.empty(3, 45, 70, device=dev)
    ops.transpose(x.to(dev), out, 3, 70, 45)
    assert rel(out, x.transpose(1, 2)) == 0
    # maxpool 3x3 s2 p1
    a = torch.randn(2, 8, 12, 12, generator=g, requires_grad=True)
    y = F.max_pool2d(a, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ad = cl(a.detach()).to(dev)
    yo, idx = ops.empty(2 * 36, 8, device=dev), torch.empty(2 * 36 * 8, dtype=torch.uint8, device=dev)
    call("sv_maxpool2d_fwd", ptr(ad), ptr(yo), ptr(idx), 2, 12, 12, 8)
    assert rel(yo, cl(y.detach()).reshape(-1, 8)) == 0
    dx = ops.zeros(2 * 144, 8, device=dev)
    call("sv_maxpool2d_bwd", ptr(keep(cl(dy).reshape(-1, 8).to(dev))), ptr(idx), ptr(dx), 2, 12, 12, 8)
    assert rel(dx, cl(a.grad).reshape(-1, 8)) < 1e-6
    # avgpool 2x2 into a column slice
    b = torch.randn(2, 8, 14, 14, generator=g, requires_grad=True)
    yb = F.avg_pool2d(b, 2, 2)
    dyb = torch.randn(yb.shape, generator=g)
    yb.backward(dyb)
    wide = ops.zeros(2 * 49, 16, device=dev)
    call("sv_avgpool2_fwd", ptr(cl(b.detach()).to(dev)), ptr(wide), 2, 14, 14, 8, 16, 8)
    assert rel(wide[:, 8:], cl(yb.detach()).reshape(-1, 8)) < 1e-6
    dwide = torch.zeros(2 * 49, 16); dwide[:, 8:] = cl(dyb).reshape(-1, 8)
    dxb = ops.empty(2 * 196, 8, device=dev)
    call("sv_avgpool2_bwd", ptr(keep(dwide.to(dev))), ptr(dxb), 2, 14, 14, 8, 16, 8)
    assert rel(dxb, cl(b.grad).reshape(-1, 8)) < 1e-6
    # maxpool3d floor(33/2)
    c = torch.randn(2, 4, 9, 9, 9, generator=g, requires_grad=True)
    yc = F.max_pool3d(c, 2)
    dyc = torch.randn(yc.shape, generator=g)
    yc.backward(dyc)
    yo3, idx3 = ops.empty(2 * 64, 4, device=dev), torch.empty(2 * 64 * 4, dtype=torch.uint8, device=dev)
    call("sv_maxpool3d_fwd", ptr(cl(c.detach()).to(dev)), ptr(yo3), ptr(idx3), 2, 9, 9, 9, 4)
    assert rel(yo3, cl(yc.detach()).reshape(-1, 4)) == 0
    dx3 = ops.empty(2 * 729, 4, device=dev)
    call("sv_maxpool3d_bwd", ptr(keep(cl(dyc).reshape(-1, 4).to(dev))), ptr(idx3), ptr(dx3), 2, 9, 9, 9, 4)
    assert rel(dx3, cl(c.grad).reshape(-1, 4)) < 1e-6


def test_decoder_seed_head_cva_spatial(dev):
    g = torch.Generator().manual_seed(10)
    I, Cc = 3, 16
    f = torch.randn(I, Cc, 7, 7, generator=g, requires_grad=True)
    s = F.adaptive_avg_pool2d(f, (2, 2))[:, :, None].expand(-1, -1, 2, -1, -1).contiguous()
    ds = torch.randn(s.shape, generator=g)
    s.backward(ds)
    out = ops.empty(I * 8, Cc, device=dev)
    call("sv_decoder_seed_fwd", ptr(cl(f.detach()).to(dev)), ptr(out), I, Cc)
    assert rel(out, cl(s.detach()).reshape(-1, Cc)) < 1e-6
    df = ops.empty(I * 49, Cc, device=dev)
    call("sv_decoder_seed_bwd", ptr(keep(cl(ds).reshape(-1, Cc).to(dev))), ptr(df), I, Cc)
    assert rel(df, cl(f.grad).reshape(-1, Cc)) < 1e-6
    # depthwise 2x2/2 + bilinear 3->7 + residual
    x = torch.randn(I, Cc, 7, 7, generator=g, requires_grad=True)
    w = torch.randn(Cc, 1, 2, 2, generator=g, requires_grad=True)
    b = torch.randn(Cc, generator=g, requires_grad=True)
    small = F.conv2d(x, w, b, stride=2, groups=Cc)
    up = F.interpolate(small, size=(7, 7), mode="bilinear", align_corners=False) + x
    dup = torch.randn(up.shape, generator=g)
    up.backward(dup)
    xd = cl(x.detach()).to(dev)
    sm = ops.empty(I * 9, Cc, device=dev)
    call("sv_dwconv2x2_fwd", ptr(xd), ptr(keep(w.detach().to(dev))), ptr(keep(b.detach().to(dev))), ptr(sm), I, Cc)
    assert rel(sm, cl(small.detach()).reshape(-1, Cc)) < 1e-6
    upo = ops.empty(I * 49, Cc, device=dev)
    call("sv_upsample3to7_add_fwd", ptr(sm), ptr(xd), Cc, ptr(upo), I, Cc)
    assert rel(upo, cl(up.detach()).reshape(-1, Cc)) < 1e-5
    dupd = cl(dup).reshape(-1, Cc).to(dev)
    dsm = ops.empty(I * 9, Cc, device=dev)
    call("sv_upsample3to7_bwd", ptr(dupd), ptr(dsm), I, Cc)
    dx, dw, db = ops.empty(I * 49, Cc, device=dev), ops.zeros(Cc, 1, 2, 2, device=dev), ops.zeros(Cc, device=dev)
    call("sv_dwconv2x2_bwd", ptr(dsm), ptr(xd), ptr(keep(w.detach().to(dev))), ptr(dx), ptr(dw), ptr(db), I, Cc)
    assert rel(dx + dupd, cl(x.grad).reshape(-1, Cc)) < 1e-5 and rel(dw, w.grad) < 1e-5 and rel(db, b.grad) < 1e-5


def test_merge_bce_iou_dropout(dev):
    g = torch.Generator().manual_seed(12)
    B, V, S = 2, 5, 4096
    wl = torch.randn(B, V, S, generator=g, requires_grad=True)
    vol = torch.randn(B, V, S, generator=g, requires_grad=True)
    out = (vol * wl.softmax(1)).sum(1)
    do = torch.randn(out.shape, generator=g)
    out.backward(do)
    wd, vd = wl.detach().to(dev), vol.detach().to(dev)
    od = ops.empty(B, S, device=dev)
    call("sv_merge_views_fwd", ptr(wd), ptr(vd), ptr(od), B, V, S)
    assert rel(od, out) < 1e-5
    dw, dv = ops.empty(B, V, S, device=dev), ops.empty(B, V, S, device=dev)
    call("sv_merge_views_bwd", ptr(wd), ptr(vd), ptr(od), ptr(keep(do.to(dev))), ptr(dw), ptr(dv), B, V, S)
    assert rel(dw, wl.grad) < 1e-5 and rel(dv, vol.grad) < 1e-5
    # BCE with logits
    x = (3 * torch.randn(B, S, generator=g)).requires_grad_(True)
    t = (torch.rand(B, S, generator=g) < 0.1).float()
    loss = F.binary_cross_entropy_with_logits(x, t)
    (loss * 1.7).backward()
    ld, dx = ops.zeros(1, device=dev), ops.empty(B, S, device=dev)
    gs = torch.tensor([1.7], device=dev)
    call("sv_bce_logits", ptr(keep(x.detach().to(dev))), ptr(keep(t.to(dev))), B * S, ptr(ld), ptr(dx), ptr(gs))
    assert abs(float(ld) - float(loss)) < 1e-5 * max(1, abs(float(loss))) and rel(dx, x.grad) < 1e-5
    # IoU counters vs the reference definition
    import oracle as O
    ths = torch.tensor([0.2, 0.3, 0.4, 0.5])
    cnt = ops.empty(B, 4, 4, device=dev)
    call("sv_iou_counts", ptr(keep(x.detach().to(dev))), ptr(keep(t.to(dev))), ptr(keep(ths.to(dev))), 4, B, S, ptr(cnt))
    ref = O.iou_at_thresholds(x.detach().view(B, 16, 16, 16), t.view(B, 16, 16, 16))
    reff = O.fscore_at_thresholds(x.detach().view(B, 16, 16, 16), t.view(B, 16, 16, 16))
    c = cnt.cpu()
    for b in range(B):
        for k in range(4):
            tp, un, fp, fn = (float(v) for v in c[b, k])
            assert abs(tp / un - ref[b][k]) < 1e-3 and abs(un - (tp + fp + fn)) < 0.5
            pr, rc = tp / (tp + fp + 1e-8), tp / (tp + fn + 1e-8)
            assert abs(2 * pr * rc / (pr + rc + 1e-8) - reff[b][k]) < 1e-3
    with pytest.raises(RuntimeError, match="iou_counts"):
        call("sv_iou_counts", ptr(x.detach().to(dev)), ptr(t.to(dev)), ptr(ths.to(dev)), 9, B, S, ptr(cnt))
    # dropout: mask statistics and fwd/bwd consistency
    n = 1 << 20
    ones = torch.ones(n, device=dev)
    y1, y2 = ops.empty(n, device=dev), ops.empty(n, device=dev)
    call("sv_dropout", ptr(ones), ptr(y1), n, 0.1, 1234, None)
    call("sv_dropout", ptr(ones), ptr(y2), n, 0.1, 1234, None)
    assert torch.equal(y1, y2)
    # device-side seed epoch (hipGraph replays freeze the scalar seed): same word -> same mask, another word -> another mask
    ep = torch.tensor([3], dtype=torch.int32, device=dev)
    y3, y4 = ops.empty(n, device=dev), ops.empty(n, device=dev)
    call("sv_dropout", ptr(ones), ptr(y3), n, 0.1, 1234, ptr(ep))
    call("sv_dropout", ptr(ones), ptr(y4), n, 0.1, 1234, ptr(ep))
    assert torch.equal(y3, y4) and not torch.equal(y3, y1)
    ep.add_(1)
    call("sv_dropout", ptr(ones), ptr(y4), n, 0.1, 1234, ptr(ep))
    assert not torch.equal(y3, y4) and abs(float((y4 == 0).float().mean()) - 0.1) < 5e-3
    frac = float((y1 == 0).float().mean())
    assert abs(frac - 0.1) < 5e-3 and abs(float(y1.max()) - 1 / 0.9) < 1e-6


def test_error_convention(dev):
    """Bad arguments come back as RuntimeError with a message, never a crash (SURVEY 8b error convention)."""
    x = ops.empty(10, 10, device=dev)
    with pytest.raises(RuntimeError, match="layernorm_fwd"):
        ops.layernorm_fwd(x, x, x, 10, 10)          # C=10 not a multiple of 4
    with pytest.raises(RuntimeError, match="window_attention"):
        call("sv_window_attention_fwd", ptr(x), ptr(x), ptr(x), 1, 10, 10, 96, 3, 0, 0)


# ------------------------------------------------------------------------------------------------ LDS-halo MFMA stencils
@pytest.mark.parametrize("cin,cout,groups", [(9, 9, 1), (9, 1, 1), (36, 9, 3)])
def test_stencil3_fwd_dgrad_wgrad(dev, cin, cout, groups):
    """csrc/stencil.hip vs torch conv3d on bf16-rounded operands (fp32 accumulate)."""
    g = torch.Generator().manual_seed(cin + cout)
    n, D = 2, 32
    bf = lambda t: t.bfloat16().float()
    x = bf(torch.randn(n, cin, D, D, D, generator=g)).requires_grad_(True)
    w = bf(torch.randn(cout, cin, 3, 3, 3, generator=g) / math.sqrt(cin * 27)).requires_grad_(True)
    b = torch.randn(cout, generator=g)
    y = F.conv3d(x, w, b, padding=1)
    dy = bf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    M = n * D ** 3
    cmem = 16 * groups if groups == 1 else 48
    ld = 12 if groups == 1 else 48
    cols = torch.arange(cin) if groups == 1 else torch.tensor([12 * k + j for k in range(4) for j in range(9)])
    xm = torch.zeros(M, ld); xm[:, cols] = cl(x.detach()).reshape(M, cin)
    wf = torch.zeros(16, 27, cmem); wf[:cout][:, :, cols] = w.detach().reshape(cout, cin, 27).permute(0, 2, 1)
    xd, wfd, bd = xm.to(dev), wf.to(dev).bfloat16().contiguous(), b.to(dev)
    out = ops.zeros(M, 12, device=dev)
    stats = torch.zeros(ops.BN_SLOTS, 2 * cout, dtype=torch.float64, device=dev)
    call("sv_stencil3_fwd", ptr(xd), ld, ld, groups, ptr(wfd), 1, ptr(bd), ptr(out), 12, 0, cout, None, 0, ptr(stats), n, D, D, D, 0, 0)
    ref = cl(y.detach()).reshape(M, cout)
    assert rel(out[:, :cout], ref) < 2e-5
    st = stats.sum(0)
    assert rel(st[:cout], ref.double().sum(0)) < 1e-5 and rel(st[cout:], (ref.double() ** 2).sum(0)) < 1e-5
    # data gradient = the same kernel with flipped / transposed weights (rows = memory channels of the conv input)
    nt = 1 if groups == 1 else 3
    wd = torch.zeros(16 * nt, 27, 16); wd[cols, :, :cout] = w.detach().reshape(cout, cin, 27).flip(2).permute(1, 2, 0)
    dym = torch.zeros(M, 12); dym[:, :cout] = cl(dy).reshape(M, cout)
    dyd, wdd = dym.to(dev), wd.to(dev).bfloat16().contiguous()
    base = torch.randn(M, ld, generator=g)
    dx = base.clone().to(dev)
    call("sv_stencil3_fwd", ptr(dyd), 12, 12, 1, ptr(wdd), nt, None, ptr(dx), ld, 0, ld if groups == 3 else 9, ptr(dx), ld, None, n, D, D, D, 0, 0)
    assert rel((dx.cpu() - base)[:, cols], cl(x.grad).reshape(M, cin)) < 2e-5
    # weight gradient (bf16 operands, fp32 atomics)
    dw = ops.zeros(cout, cin, 3, 3, 3, device=dev)
    call("sv_stencil3_wgrad", ptr(xd), ld, ld, groups, ptr(dyd), 12, 12, ptr(dw), None, None, cout, cin, 16 if groups == 1 else 12, 9, n, D, D, D, 0)
    assert rel(dw, w.grad) < 2e-4
    if groups == 3:
        # planar channel storage (four dense 12-wide planes instead of 48-wide rows): same numbers from / into the planes
        xp = xm.view(M, 4, 12).permute(1, 0, 2).contiguous().to(dev)                      # [4][M][12]
        out2 = ops.zeros(M, 12, device=dev)
        call("sv_stencil3_fwd", ptr(xp), 12, 48, 3, ptr(wfd), 1, ptr(bd), ptr(out2), 12, 0, cout, None, 0, None, n, D, D, D, M * 12, 0)
        assert torch.equal(out2, out)
        dxp = ops.zeros(4, M, 12, device=dev)
        call("sv_stencil3_fwd", ptr(dyd), 12, 12, 1, ptr(wdd), 3, None, ptr(dxp), 12, 0, 48, None, 0, None, n, D, D, D, 0, M * 12)
        assert rel(dxp.permute(1, 0, 2).reshape(M, 48).cpu()[:, cols], cl(x.grad).reshape(M, cin)) < 2e-5
        dw2 = ops.zeros(cout, cin, 3, 3, 3, device=dev)
        call("sv_stencil3_wgrad", ptr(xp), 12, 48, 3, ptr(dyd), 12, 12, ptr(dw2), None, None, cout, cin, 12, 9, n, D, D, D, M * 12)
        assert rel(dw2, w.grad) < 2e-4
        with pytest.raises(RuntimeError, match="planar output"):
            call("sv_stencil3_fwd", ptr(dyd), 12, 12, 1, ptr(wdd), 3, None, ptr(dxp), 12, 4, 48, None, 0, None, n, D, D, D, 0, M * 12)
