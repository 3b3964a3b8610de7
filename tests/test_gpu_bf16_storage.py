"""bf16 ACTIVATION STORAGE (ops.set_storage("bf16"), C ABI act_dtype = SV_BF16): every act-typed entry point is run twice
through the C ABI on the same bf16-representable inputs - once with fp32 tensors (the variant the other GPU tests pin
against torch / the oracle) and once with bf16 tensors - and the two results must agree to bf16 rounding of the outputs.
Tolerance: max|a-b| <= 1.2e-2 * max|a| (one bf16 ulp is 2^-8 relative; sums of rounded terms stay well inside)."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from swinvox_amd import hip, ops  # noqa: E402
from swinvox_amd.hip import call, ptr  # noqa: E402
from swinvox_amd.ops import ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, BatchNormState, ConvSpec  # noqa: E402

TOL = 1.2e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    hip.load()
    return torch.device("cuda", 0)


def q(t):
    """make values exactly representable in bf16 so both storage variants see identical inputs"""
    return t.bfloat16().float()


def rnd(g, *shape, scale=1.0):
    return q(torch.randn(*shape, generator=g) * scale)


_HELD = []   # device temporaries stay referenced until the launches that read them were synchronised


def D(t, dev):
    """fp32 parameter / index tensor on the device, kept alive for the current both() pass"""
    d = t.to(dev)
    _HELD.append(d)
    return d


def both(fn, dev):
    """run fn(A) under fp32 storage and bf16 storage; A(t) places an fp32 CPU activation tensor on the device in the
    current storage dtype.  fn returns a dict name -> tensor (activations in storage dtype or fp32 parameter gradients)."""
    res = []
    for mode in ("f32", "bf16"):
        ops.set_math("bf16")
        ops.set_storage(mode)
        try:
            dt = torch.float32 if mode == "f32" else torch.bfloat16
            _HELD.clear()

            def A(t, dt=dt):
                d = t.to(dev).to(dt).contiguous()
                _HELD.append(d)
                return d

            out = fn(A)
            torch.cuda.synchronize()
            _HELD.clear()
            res.append({k: v.float().cpu() for k, v in out.items()})
        finally:
            ops.set_math("f32")
    a, b = res
    for k in a:
        den = float(a[k].abs().max()) + 1e-20
        err = float((a[k] - b[k]).abs().max()) / den
        assert math.isfinite(err) and err <= TOL, (k, err)


def test_linear_all_epilogues(dev):
    g = torch.Generator().manual_seed(1)
    M, K, N = 400, 192, 160
    x, w, b = rnd(g, M, K), rnd(g, N, K, scale=1 / math.sqrt(K)), rnd(g, N)
    res, dy, src = rnd(g, M, N), rnd(g, M, N), rnd(g, M, K)
    sc = torch.tensor([1.0, 0.0, 1.25, 1.0])
    sp = ConvSpec.linear(K, N)

    def run(A):
        wd, bd = D(w, dev), D(b, dev)
        xd, dyd = A(x), A(dy)
        y, pre, y2 = ops.empty(M, N, device=dev), ops.empty(M, N, device=dev), ops.empty(M, N, device=dev)
        ops.linear_fwd(xd, M, sp, wd, y, bias=bd, act=ACT_GELU, pre_act=pre)
        ops.linear_fwd(xd, M, sp, wd, y2, bias=bd, residual=A(res), ldr=N, row_scale=D(sc, dev), rows_per_scale=100)
        dx = ops.empty(M, K, device=dev)
        ops.linear_dgrad(dyd, M, sp, sp.pack_dgrad(wd), dx, act_grad_src=A(src), act_grad_kind=ACT_GELU)
        dw, db = ops.fzeros(N, K, device=dev), ops.fzeros(N, device=dev)
        ops.linear_wgrad(dyd, xd, M, sp, dw, db)
        return dict(y=y, pre=pre, y2=y2, dx=dx, dw=dw, db=db)

    both(run, dev)


@pytest.mark.parametrize("cin,cout,k,s,p,H", [(64, 64, 3, 1, 1, 14), (256, 256, 3, 2, 1, 14), (3, 64, 7, 2, 3, 32), (64, 128, 1, 2, 0, 14)])
def test_conv2d(dev, cin, cout, k, s, p, H):
    g = torch.Generator().manual_seed(cin + k)
    n = 3
    sp = ConvSpec.conv2d(cin, cout, k, s, p)
    og = sp.out_grid((1, H, H))
    M, Min = n * og[1] * og[2], n * H * H
    x, w, b, dy = rnd(g, Min, cin), rnd(g, cout, cin, k, k, scale=1 / math.sqrt(cin * k * k)), rnd(g, cout), rnd(g, M, cout)

    def run(A):
        wd = D(w, dev)
        xd, dyd = A(x), A(dy)
        y = ops.empty(M, cout, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 2 * cout, dtype=torch.float64, device=dev)
        sp.forward(xd, n, (1, H, H), sp.pack_fwd(wd), y, bias=D(b, dev), stats=stats)
        out = dict(y=y, stats=stats.sum(0).float())
        if cin % 4 == 0:
            dx = ops.empty(Min, cin, device=dev)
            sp.dgrad(dyd, n, (1, H, H), sp.pack_dgrad(wd), dx)
            out["dx"] = dx
        dw, db = ops.fzeros(cout, cin, k, k, device=dev), ops.fzeros(cout, device=dev)
        sp.wgrad(dyd, xd, n, (1, H, H), dw, db=db)
        out.update(dw=dw, db=db)
        return out

    both(run, dev)


def test_tconv3d(dev):
    g = torch.Generator().manual_seed(3)
    cin, cout, n, G = 64, 32, 2, (4, 4, 4)
    sp = ConvSpec.conv3d(cin, cout, 4, 2, 1, transposed=True)
    og = sp.out_grid(G)
    Min, M = n * 64, n * og[0] * og[1] * og[2]
    x, w, dy = rnd(g, Min, cin), rnd(g, cin, cout, 4, 4, 4, scale=1 / math.sqrt(cin * 8)), rnd(g, M, cout)

    def run(A):
        wd, xd, dyd = D(w, dev), A(x), A(dy)
        y = ops.empty(M, cout, device=dev)
        sp.forward(xd, n, G, sp.pack_fwd(wd), y)
        dx = ops.empty(Min, cin, device=dev)
        sp.dgrad(dyd, n, G, sp.pack_dgrad(wd), dx)
        dw, db = ops.fzeros(cin, cout, 4, 4, 4, device=dev), ops.fzeros(cout, device=dev)
        sp.wgrad(dyd, xd, n, G, dw, db=db)
        return dict(y=y, dx=dx, dw=dw, db=db)

    both(run, dev)


@pytest.mark.parametrize("merge", [False, True])
def test_layernorm(dev, merge):
    g = torch.Generator().manual_seed(4)
    I, H, C0 = 3, 8, 48
    if merge:
        rows, Cd, mh = I * 16, 4 * C0, (H, H)
        x = rnd(g, I * H * H, C0)
    else:
        rows, Cd, mh = 500, 96, (0, 0)
        x = rnd(g, rows, Cd)
    gam, bet, dy = rnd(g, Cd), rnd(g, Cd), rnd(g, rows, Cd)

    def run(A):
        xd = A(x)
        y, mean, rstd = ops.layernorm_fwd(xd, D(gam, dev), D(bet, dev), rows, Cd, merge_hw=mh)
        dx = ops.zeros(*x.shape, device=dev)
        dg, db = ops.fzeros(Cd, device=dev), ops.fzeros(Cd, device=dev)
        ops.layernorm_bwd(A(dy), xd, D(gam, dev), mean, rstd, dx, dg, db, rows, Cd, merge_hw=mh)
        return dict(y=y, dx=dx, dg=dg, db=db)

    both(run, dev)


def test_ln_image(dev):
    g = torch.Generator().manual_seed(5)
    I, L = 3, 7 * 7 * 96
    x, w, b, dy = rnd(g, I, L), rnd(g, L), rnd(g, L), rnd(g, I, L)

    def run(A):
        xd = A(x)
        y = ops.empty(I, L, device=dev)
        mr = ops.fempty(2 * I, device=dev)
        nws = int(hip.load().sv_ln_image_workspace_floats(I, L))
        ws = ops.fempty(nws, device=dev)
        call("sv_ln_image_fwd", ptr(xd), ptr(D(w, dev)), ptr(D(b, dev)), ptr(y), ptr(mr), ptr(ws), I, L, 1e-5, 0.05, 77, None)
        dx = ops.empty(I, L, device=dev)
        dw, db = ops.fzeros(L, device=dev), ops.fzeros(L, device=dev)
        sums = torch.empty(2 * I, dtype=torch.float64, device=dev)
        call("sv_ln_image_bwd", ptr(A(dy)), ptr(xd), ptr(D(w, dev)), ptr(mr), ptr(dx), ptr(dw), ptr(db), ptr(sums), I, L, 0.05, 77, None)
        return dict(y=y, dx=dx, dw=dw, db=db)

    both(run, dev)


@pytest.mark.parametrize("Cc,ld", [(64, 64), (9, 12)])
def test_batchnorm_apply_and_backward(dev, Cc, ld):
    g = torch.Generator().manual_seed(6)
    M = 3000
    x, res, dz = rnd(g, M, ld), rnd(g, M, ld), rnd(g, M, ld)
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.copy_(rnd(g, Cc).abs() + 0.5); bn.bias.copy_(rnd(g, Cc))

    def run(A):
        import copy
        b2 = copy.deepcopy(bn).to(dev)
        xd = A(x)
        st = BatchNormState(b2, M, True)
        call("sv_bn_stats", ptr(xd), M, Cc, ld, ptr(st.sums))
        st.finalize()
        z = ops.zeros(M, ld, device=dev)
        st.apply(xd, ld, z, ld, ACT_LRELU, 0.2, A(res), ld)
        dx, dres = ops.zeros(M, ld, device=dev), ops.zeros(M, ld, device=dev)
        dg, db = ops.fzeros(Cc, device=dev), ops.fzeros(Cc, device=dev)
        st.backward(A(dz), ld, z, ld, xd, ld, dx, ld, dg, db, ACT_LRELU, 0.2, dres, ld)
        return dict(z=z, dx=dx, dres=dres, dg=dg, db=db, rm=b2.running_mean, rv=b2.running_var)

    both(run, dev)


@pytest.mark.parametrize("shift", [0, 3])
def test_window_attention(dev, shift):
    g = torch.Generator().manual_seed(7 + shift)
    I, H, heads = 2, 14, 3
    Cd = heads * 32
    qkv, table, dout = rnd(g, I * H * H, 3 * Cd, scale=0.5), rnd(g, 169, heads, scale=0.2), rnd(g, I * H * H, Cd)

    def run(A):
        qd = A(qkv)
        out = ops.empty(I * H * H, Cd, device=dev)
        call("sv_window_attention_fwd", ptr(qd), ptr(D(table, dev)), ptr(out), I, H, H, Cd, heads, shift, hip.MATH_BF16)
        dqkv = ops.empty(I * H * H, 3 * Cd, device=dev)
        dt = ops.fzeros(169, heads, device=dev)
        call("sv_window_attention_bwd", ptr(qd), ptr(D(table, dev)), ptr(A(dout)), ptr(dqkv), ptr(dt), None, I, H, H, Cd, heads, shift, hip.MATH_BF16)
        return dict(out=out, dqkv=dqkv, dtable=dt)

    both(run, dev)


def test_cross_view_attention(dev):
    g = torch.Generator().manual_seed(9)
    B, V, P, R, heads = 2, 3, 9, 128, 4
    qkv, dout = rnd(g, B * V * P, 3 * R), rnd(g, B * V * P, R)

    def run(A):
        qd = A(qkv)
        out, dqkv = ops.empty(B * V * P, R, device=dev), ops.empty(B * V * P, 3 * R, device=dev)
        call("sv_cross_view_attention_fwd", ptr(qd), ptr(out), B, V, P, R, heads)
        call("sv_cross_view_attention_bwd", ptr(qd), ptr(A(dout)), ptr(dqkv), B, V, P, R, heads)
        return dict(out=out, dqkv=dqkv)

    both(run, dev)


def test_elementwise_and_pools(dev):
    g = torch.Generator().manual_seed(10)
    I, Cc = 2, 64
    x112 = rnd(g, I * 16 * 16, Cc)
    dy_mp = rnd(g, I * 8 * 8, Cc)
    a, b, c = rnd(g, 300, Cc), rnd(g, 300, Cc), rnd(g, 300, Cc)
    v3 = rnd(g, 2 * 9 * 9 * 9, 32)
    d3 = rnd(g, 2 * 4 * 4 * 4, 32)
    sc = torch.tensor([0.0, 1.25, 1.0])

    def run(A):
        out = {}
        xd = A(x112)
        mp = ops.empty(I * 8 * 8, Cc, device=dev)
        idx = torch.empty(I * 8 * 8 * Cc, dtype=torch.uint8, device=dev)
        call("sv_maxpool2d_fwd", ptr(xd), ptr(mp), ptr(idx), I, 16, 16, Cc)
        dmp = ops.empty(I * 16 * 16, Cc, device=dev)
        call("sv_maxpool2d_bwd", ptr(A(dy_mp)), ptr(idx), ptr(dmp), I, 16, 16, Cc)
        out.update(mp=mp, dmp=dmp)
        ap = ops.zeros(I * 8 * 8, 2 * Cc, device=dev)
        call("sv_avgpool2_fwd", ptr(xd), ptr(ap), I, 16, 16, Cc, 2 * Cc, Cc)
        dap = ops.empty(I * 16 * 16, Cc, device=dev)
        call("sv_avgpool2_bwd", ptr(ap), ptr(dap), I, 16, 16, Cc, 2 * Cc, Cc)
        out.update(ap=ap, dap=dap)
        ad, bd, cd = A(a), A(b), A(c)
        s3 = ops.empty(300, Cc, device=dev)
        call("sv_add_n", ptr(ad), ptr(bd), ptr(cd), None, ptr(s3), 300, Cc, Cc)
        ax = ops.empty(300, Cc, device=dev)
        call("sv_axpby", ptr(ad), ptr(bd), ptr(ax), 0.5, -2.0, 300 * Cc)
        ax_odd = ops.empty(300 * Cc - 3, device=dev)
        call("sv_axpby", ptr(ad), ptr(bd), ptr(ax_odd), 0.5, -2.0, 300 * Cc - 3)
        rb = ops.empty(300, Cc, device=dev)
        call("sv_relu_bwd", ptr(ad), ptr(bd), ptr(rb), 300 * Cc)
        dr = ops.empty(300, Cc, device=dev)
        call("sv_dropout", ptr(ad), ptr(dr), 300 * Cc, 0.1, 123, None)
        rs = ops.empty(300, Cc, device=dev)
        call("sv_rowscale", ptr(ad), ptr(D(sc, dev)), ptr(rs), 300, Cc, 100)
        tr = ops.empty(Cc, 300, device=dev)
        ops.transpose(ad, tr, 1, 300, Cc)
        cs = ops.fzeros(Cc, device=dev)
        ops.colsum(ad, 300, Cc, Cc, cs)
        out.update(s3=s3, ax=ax, ax_odd=ax_odd, rb=rb, dr=dr, rs=rs, tr=tr, cs=cs)
        vd = A(v3)
        p3 = ops.empty(2 * 64, 32, device=dev)
        i3 = torch.empty(2 * 64 * 32, dtype=torch.uint8, device=dev)
        call("sv_maxpool3d_fwd", ptr(vd), ptr(p3), ptr(i3), 2, 9, 9, 9, 32)
        dv = ops.empty(2 * 729, 32, device=dev)
        call("sv_maxpool3d_bwd", ptr(A(d3)), ptr(i3), ptr(dv), 2, 9, 9, 9, 32)
        out.update(p3=p3, dv=dv)
        return out

    both(run, dev)


def test_cva_spatial_and_decoder_tail(dev):
    g = torch.Generator().manual_seed(11)
    I, Cc = 3, 64
    x, w, b = rnd(g, I * 49, Cc), rnd(g, Cc, 1, 2, 2), rnd(g, Cc)
    dy9, small = rnd(g, I * 9, Cc), rnd(g, I * 9, Cc)
    feat, dseed = rnd(g, I * 49, Cc), rnd(g, I * 8, Cc)
    M = 2048
    x8, w8, b8 = rnd(g, M, 8), rnd(g, 8), rnd(g, 1)
    draw, dvol = rnd(g, M, 12), rnd(g, M)
    B, V, S = 2, 3, 512
    wl, vol, dout = rnd(g, B * V * S), rnd(g, B * V * S), rnd(g, B * S)

    def run(A):
        out = {}
        xd = A(x)
        y = ops.empty(I * 9, Cc, device=dev)
        call("sv_dwconv2x2_fwd", ptr(xd), ptr(D(w, dev)), ptr(D(b, dev)), ptr(y), I, Cc)
        dx = ops.empty(I * 49, Cc, device=dev)
        dw, db = ops.fzeros(Cc, 1, 2, 2, device=dev), ops.fzeros(Cc, device=dev)
        call("sv_dwconv2x2_bwd", ptr(A(dy9)), ptr(xd), ptr(D(w, dev)), ptr(dx), ptr(dw), ptr(db), I, Cc)
        up = ops.empty(I * 49, Cc, device=dev)
        call("sv_upsample3to7_add_fwd", ptr(A(small)), ptr(xd), Cc, ptr(up), I, Cc)
        ds = ops.empty(I * 9, Cc, device=dev)
        call("sv_upsample3to7_bwd", ptr(xd), ptr(ds), I, Cc)
        out.update(y=y, dx=dx, dw=dw, db=db, up=up, ds=ds)
        seed = ops.empty(I * 8, Cc, device=dev)
        call("sv_decoder_seed_fwd", ptr(A(feat)), ptr(seed), I, Cc)
        dfe = ops.empty(I * 49, Cc, device=dev)
        call("sv_decoder_seed_bwd", ptr(A(dseed)), ptr(dfe), I, Cc)
        out.update(seed=seed, dfe=dfe)
        x8d = A(x8)
        raw, vo = ops.empty(M, 12, device=dev), ops.empty(M, device=dev)
        call("sv_decoder_head_fwd", ptr(x8d), ptr(D(w8, dev)), ptr(D(b8, dev)), ptr(raw), ptr(vo), M)
        dx8 = ops.empty(M, 8, device=dev)
        dw8, db8 = ops.fzeros(8, device=dev), ops.fzeros(1, device=dev)
        call("sv_decoder_head_bwd", ptr(A(draw)), ptr(A(dvol)), ptr(x8d), ptr(D(w8, dev)), ptr(dx8), ptr(dw8), ptr(db8), M)
        out.update(raw=raw, vo=vo, dx8=dx8, dw8=dw8, db8=db8)
        wld, vold = A(wl), A(vol)
        mo = ops.empty(B * S, device=dev)
        call("sv_merge_views_fwd", ptr(wld), ptr(vold), ptr(mo), B, V, S)
        dwl, dvo = ops.empty(B * V * S, device=dev), ops.empty(B * V * S, device=dev)
        call("sv_merge_views_bwd", ptr(wld), ptr(vold), ptr(mo), ptr(A(dout)), ptr(dwl), ptr(dvo), B, V, S)
        out.update(mo=mo, dwl=dwl, dvo=dvo)
        return out

    both(run, dev)


@pytest.mark.parametrize("li", [0, 4])
def test_merger_stencils(dev, li):
    g = torch.Generator().manual_seed(12 + li)
    I, Dg = 1, 16
    M = I * Dg * Dg * Dg
    cin_mem, groups = (48, 3) if li == 4 else (12, 1)
    x = rnd(g, M, cin_mem)
    wp = rnd(g, 16, 27, 16 * groups, scale=0.1).bfloat16()
    bias = rnd(g, 9)
    dy = rnd(g, M, 12)

    def run(A):
        xd = A(x)
        y = ops.zeros(M, 12, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 18, dtype=torch.float64, device=dev)
        call("sv_stencil3_fwd", ptr(xd), cin_mem, cin_mem, groups, ptr(D(wp, dev)), 1, ptr(D(bias, dev)), ptr(y), 12, 0, 9, None, 0, ptr(stats),
             I, Dg, Dg, Dg, 0, 0)
        cin = 36 if li == 4 else 9
        dw, db = ops.fzeros(9, cin, 27, device=dev), ops.fzeros(9, device=dev)
        ws = ops.fzeros(int(hip.load().sv_stencil3_wgrad_workspace_floats(9, cin)), device=dev)
        call("sv_stencil3_wgrad", ptr(xd), cin_mem, cin_mem, groups, ptr(A(dy)), 12, 12, ptr(dw), ptr(db), ptr(ws), 9, cin, 12 if li == 4 else 16, 9, I, Dg, Dg, Dg, 0)
        return dict(y=y, stats=stats.sum(0).float(), dw=dw, db=db)

    both(run, dev)


def test_cast_roundtrip(dev):
    g = torch.Generator().manual_seed(13)
    x = torch.randn(1000, generator=g).to(dev)
    b = torch.empty(1000, dtype=torch.bfloat16, device=dev)
    y = torch.empty(1000, device=dev)
    call("sv_cast", ptr(x), hip.F32, ptr(b), hip.BF16, 1000)
    call("sv_cast", ptr(b), hip.BF16, ptr(y), hip.F32, 1000)
    assert torch.equal(b, x.bfloat16()) and torch.equal(y, x.bfloat16().float())
