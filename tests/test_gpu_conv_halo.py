"""Halo-tile convolution kernels (csrc/conv_halo.hip) behind sv_conv_gather / sv_tconv_gather: the 3 x 3 / stride 1 / 64 -> 64 convolution of the
ResNet layer1 bottlenecks (forward with BatchNorm statistics, and data gradient) and the 4 x 4 stem on the space-to-depth image, against torch
fp32 on bf16-representable operands and against the gather engine on the same call (reference models/encoder.py:22-23: torchvision resnet50's
conv1 and layer1.*.conv2).  Shapes: the bench grid, ragged grids (tiles cut by both image edges), one exact tile, a grid smaller than a tile."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from swinvox_amd import hip, ops  # noqa: E402
from swinvox_amd.ops import ConvSpec  # noqa: E402


def cl(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


def rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture()
def bf16_mode():
    ops.set_math("bf16")
    ops.set_storage("bf16")
    yield
    ops.set_conv_halo(1)
    ops.set_math("f32")


def _run(sp, mode, fn):
    """fn() under halo mode `mode`; returns (result, number of calls the halo kernels took)."""
    ops.set_conv_halo(mode)
    n0 = int(hip.load().sv_conv_halo_launches())
    out = fn()
    torch.cuda.synchronize()
    return out, int(hip.load().sv_conv_halo_launches()) - n0


SHAPES = [(20, 56, 56), (3, 45, 70), (2, 8, 32), (1, 5, 9), (2, 17, 33)]


@pytest.mark.parametrize("n,H,W", SHAPES)
def test_conv3x3_64_forward_statistics_and_data_gradient(dev, bf16_mode, n, H, W):
    g = torch.Generator().manual_seed(n * 1000 + H * 10 + W)
    x = torch.randn(n, 64, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)).bfloat16().float().requires_grad_(True)
    y = F.conv2d(x, w, None, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    sp = ConvSpec.conv2d(64, 64, 3, 1, 1)
    M = n * H * W
    xd, dyd, wd = cl(x).to(dev).bfloat16(), cl(dy).to(dev).bfloat16(), w.detach().to(dev)
    wf, wdg = ops.pack_one(sp, wd, "f"), ops.pack_one(sp, wd, "d")

    def fwd():
        out = ops.empty(M, 64, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 128, dtype=torch.float64, device=dev)
        sp.forward(xd, n, (1, H, W), wf, out, stats=stats)
        return out, stats

    def dgrad():
        dx = ops.empty(M, 64, device=dev)
        sp.dgrad(dyd, n, (1, H, W), wdg, dx)
        return dx

    for rep in range(2):                                   # twice: the launch's last workgroup must have zeroed its scheduler slot
        (out, stats), took = _run(sp, 2, fwd)
        assert took == 1
        assert rel(out, cl(y)) < 6e-3                      # bf16 rounding of the stored output
        st, o = stats.sum(0), out.float().cpu().double()
        assert rel(st[:64], o.sum(0)) < 1e-4 and rel(st[64:], (o * o).sum(0)) < 1e-4     # a tile counted twice / a masked column counted: >= 1e-3
        dx, took = _run(sp, 2, dgrad)
        assert took == 1
        assert rel(dx, cl(x.grad)) < 6e-3
    (eout, estats), took = _run(sp, 0, fwd)
    assert took == 0
    edx, _ = _run(sp, 0, dgrad)
    # against the gather engine: the same products in another summation order - single bf16 roundings may flip
    assert rel(out, eout) < 8e-3 and float((out.float() - eout.float()).abs().mean() / eout.float().abs().mean()) < 5e-4
    assert rel(dx, edx) < 8e-3 and float((dx.float() - edx.float()).abs().mean() / edx.float().abs().mean()) < 5e-4
    # the engine's register epilogue counts the fp32 accumulators, the halo kernel what it stored: bf16 rounding noise apart
    hs, es = stats.sum(0), estats.sum(0)
    assert rel(hs[:64], es[:64]) < 2e-2 and rel(hs[64:], es[64:]) < 5e-3


@pytest.mark.parametrize("n,H,W", [(6, 112, 112), (2, 30, 50), (1, 8, 32)])
def test_stem_4x4_on_the_space_to_depth_image(dev, bf16_mode, n, H, W):
    g = torch.Generator().manual_seed(n + H + W)
    x = torch.randn(n, 16, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, 16, 4, 4, generator=g) / 16.0).bfloat16().float()
    y = F.conv2d(F.pad(x, (2, 1, 2, 1)), w, None)          # pads (2, 1): output grid = input grid (Encoder._stem_spec)
    sp = ConvSpec.conv2d(16, 64, 4, 1, 2, og_fixed=(1, H, W))
    M = n * H * W
    xd, wf = cl(x).to(dev).bfloat16(), ops.pack_one(sp, w.to(dev), "f")

    def fwd():
        out = ops.empty(M, 64, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 128, dtype=torch.float64, device=dev)
        sp.forward(xd, n, (1, H, W), wf, out, stats=stats)
        return out, stats

    for rep in range(2):
        (out, stats), took = _run(sp, 2, fwd)
        assert took == 1
        assert rel(out, cl(y)) < 6e-3
        st, o = stats.sum(0), out.float().cpu().double()
        assert rel(st[:64], o.sum(0)) < 1e-4 and rel(st[64:], (o * o).sum(0)) < 1e-4
    (eout, estats), took = _run(sp, 0, fwd)
    assert took == 0
    hs, es = stats.sum(0), estats.sum(0)
    assert rel(out, eout) < 8e-3 and rel(hs[:64], es[:64]) < 2e-2 and rel(hs[64:], es[64:]) < 5e-3


def test_calls_the_kernels_do_not_take_stay_on_the_engine(dev, bf16_mode):
    """bias, an activation, a residual, a channel count that is no multiple of 64, another stride: the gather engine keeps the call (mode 2 = every call the kernels can take)."""
    ops.set_conv_halo(2)
    n, H = 2, 16
    x = torch.randn(n * H * H, 64, device=dev).bfloat16()
    cases = [(ConvSpec.conv2d(64, 64, 3, 1, 1), dict(bias=torch.zeros(64, device=dev))),
             (ConvSpec.conv2d(64, 64, 3, 1, 1), dict(act=ops.ACT_RELU)),
             (ConvSpec.conv2d(64, 64, 3, 1, 1), dict(residual=x, ldr=64)),
             (ConvSpec.conv2d(64, 96, 3, 1, 1), {}),
             (ConvSpec.conv2d(64, 64, 3, 2, 1), {})]
    for sp, epi in cases:
        w = torch.randn(sp.cout, 64, 3, 3, device=dev) / 24.0
        og = sp.out_grid((1, H, H))
        out = ops.empty(n * og[1] * og[2], sp.cout, device=dev)
        n0 = int(hip.load().sv_conv_halo_launches())
        sp.forward(x, n, (1, H, H), ops.pack_one(sp, w, "f"), out, **epi)
        torch.cuda.synchronize()
        assert int(hip.load().sv_conv_halo_launches()) == n0, (sp.cout, sp.s, list(epi))
        assert bool(torch.isfinite(out.float()).all())


@pytest.mark.parametrize("n,H,W,ci,co,stride,bias", [(20, 56, 56, 64, 64, 1, False), (6, 28, 28, 128, 128, 1, False), (4, 14, 14, 256, 256, 1, True),
                                                      (3, 45, 70, 64, 128, 1, False), (2, 8, 32, 128, 64, 1, True), (1, 5, 9, 64, 64, 1, False),
                                                      (6, 56, 56, 64, 64, 2, True), (4, 28, 28, 256, 256, 2, True), (3, 45, 71, 128, 64, 2, False),
                                                      (2, 14, 14, 64, 128, 2, True), (1, 6, 9, 64, 64, 2, False)])
def test_conv3x3_weight_gradient_halo(dev, bf16_mode, n, H, W, ci, co, stride, bias):
    """Halo-tile weight gradient of the 3 x 3 / padding 1 convolutions of stride 1 and 2 (channel counts multiples of 64), with and without the
    bias gradient, against torch autograd on bf16-representable operands and against wgrad_kernel on the same call; twice (the workspace
    is zeroed per call)."""
    g = torch.Generator().manual_seed(n * 1000 + H * 10 + W + ci + stride)
    x = torch.randn(n, ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)).requires_grad_(True)
    b = torch.zeros(co, requires_grad=True)
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    sp = ConvSpec.conv2d(ci, co, 3, stride, 1)
    xd, dyd = cl(x).to(dev).bfloat16(), cl(dy).to(dev).bfloat16()

    def wgrad():
        dw, db = torch.zeros(co, ci, 3, 3, device=dev), torch.zeros(co, device=dev)
        sp.wgrad(dyd, xd, n, (1, H, W), dw, db=db if bias else None)
        return dw, db

    outs = []
    for mode in (2, 2, 0):
        ops.set_conv_halo_wgrad(mode)
        n0 = int(hip.load().sv_conv_halo_launches())
        dw, db = wgrad()
        torch.cuda.synchronize()
        assert int(hip.load().sv_conv_halo_launches()) - n0 == (1 if mode else 0)
        outs.append((dw, db))
    ops.set_conv_halo_wgrad(1)
    for dw, db in outs:
        assert rel(dw, w.grad) < 2e-3, rel(dw, w.grad)          # fp32 sums of products of bf16 operands: only the summation order differs
        if bias:
            assert rel(db, b.grad) < 1e-4
    assert rel(outs[0][0], outs[2][0]) < 1e-3 and rel(outs[0][0], outs[1][0]) < 1e-4


@pytest.mark.parametrize("n,H,W,ci,co", [(40, 28, 28, 128, 128), (24, 14, 14, 256, 256), (3, 45, 70, 128, 64), (2, 17, 33, 64, 192), (1, 5, 9, 256, 128)])
def test_conv3x3_blocked_forward_statistics_and_data_gradient(dev, bf16_mode, n, H, W, ci, co):
    """3 x 3 / stride 1 with more than 64 channels (conv2 of the layer2 / layer3 bottlenecks): items of 256 positions x 64 output channels
    with a K loop over blocks of 64 input channels - forward with statistics and data gradient against torch and against the gather engine."""
    g = torch.Generator().manual_seed(n * 1000 + H * 10 + W + ci * 3 + co)
    x = torch.randn(n, ci, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci)).bfloat16().float().requires_grad_(True)
    y = F.conv2d(x, w, None, stride=1, padding=1)
    dy = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(dy)
    sp = ConvSpec.conv2d(ci, co, 3, 1, 1)
    M = n * H * W
    xd, dyd, wd = cl(x).to(dev).bfloat16(), cl(dy).to(dev).bfloat16(), w.detach().to(dev)
    wf, wdg = ops.pack_one(sp, wd, "f"), ops.pack_one(sp, wd, "d")

    def fwd():
        out = ops.empty(M, co, device=dev)
        stats = torch.zeros(ops.BN_SLOTS, 2 * co, dtype=torch.float64, device=dev)
        sp.forward(xd, n, (1, H, W), wf, out, stats=stats)
        return out, stats

    def dgrad():
        dx = ops.empty(M, ci, device=dev)
        sp.dgrad(dyd, n, (1, H, W), wdg, dx)
        return dx

    for rep in range(2):
        (out, stats), took = _run(sp, 2, fwd)
        assert took == 1
        assert rel(out, cl(y)) < 6e-3
        st, o = stats.sum(0), out.float().cpu().double()
        assert rel(st[:co], o.sum(0)) < 1e-4 and rel(st[co:], (o * o).sum(0)) < 1e-4
        dx, took = _run(sp, 2, dgrad)
        assert took == 1
        assert rel(dx, cl(x.grad)) < 6e-3
    (eout, estats), took = _run(sp, 0, fwd)
    assert took == 0
    edx, _ = _run(sp, 0, dgrad)
    assert rel(out, eout) < 8e-3 and float((out.float() - eout.float()).abs().mean() / eout.float().abs().mean()) < 5e-4
    assert rel(dx, edx) < 8e-3 and float((dx.float() - edx.float()).abs().mean() / edx.float().abs().mean()) < 5e-4
