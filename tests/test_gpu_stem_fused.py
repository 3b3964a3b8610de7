"""ResNet stem passes fused around the max-pool (sv_bn_act_maxpool_fwd / sv_bn_maxpool_bwd; reference models/encoder.py:22-23: torchvision
resnet50's bn1 / relu / maxpool) against the separate passes they replace (sv_scale_shift_act -> sv_maxpool2d_fwd, sv_maxpool2d_bwd ->
sv_bn_bwd) and against torch autograd in fp32.  The pooled map and the arg-max taps must be IDENTICAL to the separate passes (same values,
same tie rule); the gradients agree up to the summation order of the channel sums."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from swinvox_amd import ops  # noqa: E402
from swinvox_amd.hip import call, ptr  # noqa: E402
from swinvox_amd.ops import ACT_RELU, BatchNormState  # noqa: E402


def cl(t):
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


def rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("store", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,C", [(3, 112, 112, 64), (2, 13, 9, 64), (2, 8, 8, 16), (1, 7, 10, 128)])
def test_fused_stem_passes(dev, store, N, H, W, C):
    g = torch.Generator().manual_seed(N * 100 + H + W + C)
    y = torch.randn(N, C, H, W, generator=g)
    if store == "bf16":
        y = y.bfloat16().float()
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    bn.train()
    yr = y.clone().requires_grad_(True)
    ref = F.max_pool2d(F.relu(bn(yr)), 3, 2, 1)
    dmp = torch.randn(ref.shape, generator=g)
    if store == "bf16":
        dmp = dmp.bfloat16().float()
    ref.backward(dmp)
    Ho, Wo, M = (H + 1) // 2, (W + 1) // 2, N * H * W
    ops.set_math("bf16" if store == "bf16" else "f32")
    ops.set_storage(store)
    try:
        bnd = torch.nn.BatchNorm2d(C).to(dev)
        with torch.no_grad():
            bnd.weight.copy_(bn.weight); bnd.bias.copy_(bn.bias)
        yd, dmpd = ops.to_store(cl(y).to(dev)), ops.to_store(cl(dmp).to(dev))
        st = BatchNormState(bnd, M, True)
        call("sv_bn_stats", ptr(yd), M, C, C, ptr(st.sums))
        st.finalize()
        # separate passes
        z = ops.empty(M, C, like=yd)
        st.apply(yd, C, z, C, ACT_RELU, 0.0)
        mp0 = ops.empty(N * Ho * Wo, C, like=yd)
        idx0 = torch.empty(N * Ho * Wo * C, dtype=torch.uint8, device=dev)
        call("sv_maxpool2d_fwd", ptr(z), ptr(mp0), ptr(idx0), N, H, W, C)
        dz = ops.empty(M, C, like=yd)
        call("sv_maxpool2d_bwd", ptr(dmpd), ptr(idx0), ptr(dz), N, H, W, C)
        dy0, dg0, db0 = ops.empty(M, C, like=yd), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        st.backward(dz, C, None, C, yd, C, dy0, C, dg0, db0, ACT_RELU, 0.0)
        # fused passes
        mp1 = ops.empty(N * Ho * Wo, C, like=yd)
        idx1 = torch.empty(N * Ho * Wo * C, dtype=torch.uint8, device=dev)
        call("sv_bn_act_maxpool_fwd", ptr(yd), ptr(st.scale), ptr(st.shift), ptr(mp1), ptr(idx1), N, H, W, C, ACT_RELU, 0.0)
        dy1, dg1, db1 = ops.empty(M, C, like=yd), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = ops.zeros_f64((ops.BN_BWD_SLOTS + 1) * 2 * C + 2, dev)
        call("sv_bn_maxpool_bwd", ptr(dmpd), ptr(idx1), ptr(yd), ptr(bnd.weight), ptr(st.mean), ptr(st.rstd), ptr(st.scale), ptr(st.shift), N, H, W, C,
             ACT_RELU, 0.0, 1, ptr(dy1), ptr(dg1), ptr(db1), ptr(ws))
        torch.cuda.synchronize()
        assert torch.equal(mp0, mp1) and torch.equal(idx0, idx1)
        # bf16: the separate passes store the pool's gradient (a sum of up to four bf16 values) rounded, the fused passes never store it
        tol, tg = (1e-5, 1e-5) if store == "f32" else (1e-2, 5e-3)
        assert rel(dy1, dy0) < tol and rel(dg1, dg0) < tg and rel(db1, db0) < tg
        # against torch (fp32 storage: to rounding; bf16 storage: the stored values are rounded)
        t = 2e-5 if store == "f32" else 2e-2
        # bf16: values that differ in fp32 can tie after rounding, the first of them then takes the window's gradient - a few elements of dy differ
        # by O(1) from torch's (in the separate passes too): L1-relative there
        l1 = float((dy1.float().cpu() - cl(yr.grad)).abs().sum() / cl(yr.grad).abs().sum())
        assert rel(mp1, cl(ref)) < t and (rel(dy1, cl(yr.grad)) < t if store == "f32" else l1 < t)
        assert rel(dg1, bn.weight.grad) < t and rel(db1, bn.bias.grad) < t
    finally:
        ops.set_math("f32")
        ops.set_storage("f32")


@pytest.mark.parametrize("store", ["f32", "bf16"])
@pytest.mark.parametrize("N,D,H,W,C", [(2, 9, 9, 9, 32), (1, 5, 6, 7, 16), (2, 8, 8, 8, 64), (1, 17, 17, 17, 64)])
def test_fused_refiner_down_passes(dev, store, N, D, H, W, C):
    """Conv3d output -> BatchNorm3d -> LeakyReLU -> MaxPool3d(2) (floor: the last plane of an odd grid lies in no window, reference
    models/refiner.py:21-39): sv_bn_act_maxpool3d_fwd / sv_bn_maxpool3d_bwd against the separate passes and torch."""
    from swinvox_amd.ops import ACT_LRELU
    slope = 0.2
    g = torch.Generator().manual_seed(N * 100 + D + H * 3 + W * 7 + C)
    y = torch.randn(N, C, D, H, W, generator=g)
    if store == "bf16":
        y = y.bfloat16().float()
    bn = torch.nn.BatchNorm3d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    bn.train()
    yr = y.clone().requires_grad_(True)
    ref = F.max_pool3d(F.leaky_relu(bn(yr), slope), 2)
    dmp = torch.randn(ref.shape, generator=g)
    if store == "bf16":
        dmp = dmp.bfloat16().float()
    ref.backward(dmp)
    cl3 = lambda t: t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).contiguous()      # noqa: E731
    Do, Ho, Wo, M = D // 2, H // 2, W // 2, N * D * H * W
    Mo = N * Do * Ho * Wo
    ops.set_math("bf16" if store == "bf16" else "f32")
    ops.set_storage(store)
    try:
        bnd = torch.nn.BatchNorm3d(C).to(dev)
        with torch.no_grad():
            bnd.weight.copy_(bn.weight); bnd.bias.copy_(bn.bias)
        yd, dmpd = ops.to_store(cl3(y).to(dev)), ops.to_store(cl3(dmp).to(dev))
        st = BatchNormState(bnd, M, True)
        call("sv_bn_stats", ptr(yd), M, C, C, ptr(st.sums))
        st.finalize()
        z = ops.empty(M, C, like=yd)
        st.apply(yd, C, z, C, ACT_LRELU, slope)
        mp0 = ops.empty(Mo, C, like=yd)
        idx0 = torch.empty(Mo * C, dtype=torch.uint8, device=dev)
        call("sv_maxpool3d_fwd", ptr(z), ptr(mp0), ptr(idx0), N, D, H, W, C)
        dz = ops.empty(M, C, like=yd)
        call("sv_maxpool3d_bwd", ptr(dmpd), ptr(idx0), ptr(dz), N, D, H, W, C)
        dy0, dg0, db0 = ops.empty(M, C, like=yd), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        st.backward(dz, C, None, C, yd, C, dy0, C, dg0, db0, ACT_LRELU, slope)
        mp1 = ops.empty(Mo, C, like=yd)
        idx1 = torch.empty(Mo * C, dtype=torch.uint8, device=dev)
        call("sv_bn_act_maxpool3d_fwd", ptr(yd), ptr(st.scale), ptr(st.shift), ptr(mp1), ptr(idx1), N, D, H, W, C, ACT_LRELU, slope)
        dy1, dg1, db1 = ops.empty(M, C, like=yd), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = ops.zeros_f64((ops.BN_BWD_SLOTS + 1) * 2 * C + 2, dev)
        call("sv_bn_maxpool3d_bwd", ptr(dmpd), ptr(idx1), ptr(yd), ptr(bnd.weight), ptr(st.mean), ptr(st.rstd), ptr(st.scale), ptr(st.shift), N, D, H, W, C,
             ACT_LRELU, slope, 1, ptr(dy1), ptr(dg1), ptr(db1), ptr(ws))
        torch.cuda.synchronize()
        assert torch.equal(mp0, mp1) and torch.equal(idx0, idx1)
        tol, tg = (1e-5, 1e-5) if store == "f32" else (1e-2, 5e-3)
        assert rel(dy1, dy0) < tol and rel(dg1, dg0) < tg and rel(db1, db0) < tg
        t = 2e-5 if store == "f32" else 2e-2
        l1 = float((dy1.float().cpu() - cl3(yr.grad)).abs().sum() / cl3(yr.grad).abs().sum())
        assert rel(mp1, cl3(ref)) < t and (rel(dy1, cl3(yr.grad)) < t if store == "f32" else l1 < t)
        assert rel(dg1, bn.weight.grad) < t and rel(db1, bn.bias.grad) < t
    finally:
        ops.set_math("f32")
        ops.set_storage("f32")
