"""Fused Swin MLP branch (csrc/swin_mlp.hip: sv_swin_mlp_pack / _fwd / _bwd / _wgrad) against a plain PyTorch fp32 reference of
x2 = x1 + s * fc2(GELU(fc1(LayerNorm(x1)))) (timm Mlp + norm2 + DropPath of a SwinTransformerBlock, models/swin_transformer.py:78)
and its autograd gradients.  Inputs are bf16-representable; the kernels round the LayerNorm output, the hidden activation and its
gradient to bf16 exactly where the unfused chain stores them, so the tolerance is the bf16 one of tests/test_gpu_bf16_storage.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from swinvox_amd import hip  # noqa: E402
from swinvox_amd.hip import call, ptr  # noqa: E402


def _bf(t):
    return t.to(torch.bfloat16).float()


def _case(C, M, seed, with_scale):
    g = torch.Generator().manual_seed(seed)
    x1 = _bf(torch.randn(M, C, generator=g))
    dy = _bf(torch.randn(M, C, generator=g) * 0.5)
    w1 = torch.randn(4 * C, C, generator=g) / C ** 0.5
    b1 = 0.1 * torch.randn(4 * C, generator=g)
    w2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5
    b2 = 0.1 * torch.randn(C, generator=g)
    lg = 1 + 0.1 * torch.randn(C, generator=g)
    lb = 0.1 * torch.randn(C, generator=g)
    rps = 250
    sc = None
    if with_scale:
        sc = torch.tensor([0.0 if i % 3 == 1 else 1.0 / 0.9 for i in range((M + rps - 1) // rps)])
    return x1, dy, w1, b1, w2, b2, lg, lb, sc, rps


def _reference(x1, dy, w1, b1, w2, b2, lg, lb, sc, rps):
    ps = [t.clone().double().requires_grad_(True) for t in (x1, w1, b1, w2, b2, lg, lb)]
    x, W1, B1, W2, B2, G, Bt = ps
    xn = torch.nn.functional.layer_norm(x, (x.shape[1],), G, Bt, 1e-5)
    h = torch.nn.functional.gelu(xn @ W1.T + B1)
    y = h @ W2.T + B2
    s = torch.ones(x.shape[0], 1, dtype=torch.float64) if sc is None else sc.double().repeat_interleave(rps)[:x.shape[0], None]
    out = x + s * y
    out.backward(dy.double())
    return out.detach().float(), [p.grad.float() for p in ps]


@pytest.mark.parametrize("C", [96, 128, 192])
@pytest.mark.parametrize("with_scale", [False, True])
def test_fused_swin_mlp_matches_torch(dev, C, with_scale):
    M = 1000 if C != 96 else 1337          # not a multiple of the 32-token wave tile / the 128-token weight-gradient tile
    assert hip.load().sv_swin_mlp_supported(C) == 1 and hip.load().sv_swin_mlp_supported(384) == 0
    x1, dy, w1, b1, w2, b2, lg, lb, sc, rps = _case(C, M, 3 + C, with_scale)
    # the kernels see bf16 weights: give the reference the same rounded values
    w1r, w2r = _bf(w1), _bf(w2)
    want, (gx, gw1, gb1, gw2, gb2, gg, gb) = _reference(x1, dy, w1r, b1, w2r, b2, lg, lb, sc, rps)
    D = lambda t: t.to(dev).contiguous()
    xd, dyd = D(x1.to(torch.bfloat16)), D(dy.to(torch.bfloat16))
    w1d, b1d, w2d, b2d, lgd, lbd = D(w1r), D(b1), D(w2r), D(b2), D(lg), D(lb)
    scd = D(sc) if sc is not None else None
    packs = torch.empty(16 * C * C, dtype=torch.bfloat16, device=dev)
    call("sv_swin_mlp_pack", ptr(w1d), ptr(w2d), ptr(packs), C)
    x2 = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=dev)
    call("sv_swin_mlp_fwd", ptr(xd), ptr(x2), ptr(lgd), ptr(lbd), ptr(packs), ptr(b1d), ptr(b2d), ptr(scd), rps, M, C, 1e-5)
    err = float((x2.float().cpu() - want).abs().max() / want.abs().max())
    assert err < 1.5e-2, ("forward", err)
    # data gradient + LayerNorm parameter gradients
    dx1 = torch.full((M, C), float("nan"), dtype=torch.bfloat16, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    call("sv_swin_mlp_bwd", ptr(xd), ptr(dyd), ptr(dx1), ptr(lgd), ptr(lbd), ptr(packs), ptr(b1d), ptr(scd), rps, ptr(dg), ptr(db), M, C, 1e-5)

    def rel(a, b):
        return float((a.float().cpu() - b).norm() / (b.norm() + 1e-20))

    assert rel(dx1, gx) < 2e-2, ("dx1", rel(dx1, gx))
    assert float((dx1.float().cpu() - gx).abs().max() / gx.abs().max()) < 5e-2
    assert rel(dg, gg) < 2e-2 and rel(db, gb) < 2e-2, ("ln grads", rel(dg, gg), rel(db, gb))
    # weight gradients (accumulate into the buffers: start from a known offset)
    dw1, db1 = torch.full((4 * C, C), 0.5, device=dev), torch.full((4 * C,), 0.5, device=dev)
    dw2, db2 = torch.full((C, 4 * C), 0.5, device=dev), torch.full((C,), 0.5, device=dev)
    w1rows = w1d.to(torch.bfloat16).contiguous()                 # [4C][C]
    w2trows = w2d.t().to(torch.bfloat16).contiguous()            # [4C][C] = fc2.weight transposed
    call("sv_swin_mlp_wgrad", ptr(xd), ptr(dyd), ptr(lgd), ptr(lbd), ptr(w1rows), ptr(w2trows), ptr(b1d), ptr(scd), rps,
         ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), M, C, 1e-5)
    for name, got, ref in (("dw1", dw1 - 0.5, gw1), ("db1", db1 - 0.5, gb1), ("dw2", dw2 - 0.5, gw2), ("db2", db2 - 0.5, gb2)):
        assert rel(got, ref) < 2e-2, (name, rel(got, ref))
        assert float((got.cpu() - ref).abs().max() / ref.abs().max()) < 5e-2, name


def test_fused_swin_mlp_rejects_bad_arguments(dev):
    lib = hip.load()
    z = torch.zeros(64, 96, dtype=torch.bfloat16, device=dev)
    f = torch.zeros(96, device=dev)
    with pytest.raises(RuntimeError, match="swin_mlp"):
        call("sv_swin_mlp_fwd", ptr(z), ptr(z), ptr(f), ptr(f), ptr(z), ptr(f), ptr(f), None, 1, 64, 100, 1e-5)     # unsupported width
    with pytest.raises(RuntimeError, match="swin_mlp"):
        call("sv_swin_mlp_pack", ptr(f), ptr(f), None, 96)
