"""The BASELINE configurations and config.py knobs around the default path: n_views 1 / 5 / 24, odd batches,
single-stage Swin (USE_SWIN_T_MULTI_STAGE=False, encoder.py:77,140), a stage subset, no cross-view attention.  Whole train-mode
step in exact-fp32 mode against the CPU oracle (loss and refined logits); bf16 math + bf16 storage checked for a close loss
and finite gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
import swinvox_amd as S  # noqa: E402
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner  # noqa: E402

from swinvox_amd.losses import bce_with_logits as bce  # noqa: E402  (sv_bce_logits behind an autograd node)
CASES = [dict(B=2, V=1), dict(B=1, V=24), dict(B=3, V=5), dict(B=2, V=2, multi=False), dict(B=2, V=3, cva=False), dict(B=2, V=2, stages=[1, 3]),
         dict(B=2, V=3, ds=1),                  # ATT_SPATIAL_DOWNSAMPLE_RATIO = 1: cross-view attention on the 7x7 grid (cross_view_attention.py:26-34,67-73)
         dict(B=1, V=2, stages=[0, 1, 2]),      # no stage 3: timm's FeatureListNet drops layers_3, so must the state_dict
         dict(B=1, V=2, variant="base")]        # Swin-B encoder (BASELINE config 5): embed 128, depths 2/2/18/2, heads 4/8/16/32


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()).replace(" ", ""))
def test_config_variant_matches_the_oracle(dev, case):
    B, V = case["B"], case["V"]
    ocfg, pcfg = O.default_cfg(), S.default_cfg()
    for c in (ocfg, pcfg):
        if "multi" in case:
            c.NETWORK.USE_SWIN_T_MULTI_STAGE = case["multi"]
        if "cva" in case:
            c.NETWORK.USE_CROSS_VIEW_ATTENTION = case["cva"]
        if "stages" in case:
            c.NETWORK.SWIN_T_STAGES = case["stages"]
        if "ds" in case:
            c.NETWORK.ATT_SPATIAL_DOWNSAMPLE_RATIO = case["ds"]
    torch.manual_seed(0)
    variant = case.get("variant", "tiny")
    onets = [O.Encoder(ocfg, variant=variant), O.Decoder(ocfg), O.Merger(ocfg), O.Refiner(ocfg)]
    for i, n in enumerate(onets):
        O.seeded_weights_(n, seed=50 + i)
        n.train()
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, O.model.SwinBlock):
                m.dp = 0.0
    pnets = [Encoder(pcfg, variant=variant), Decoder(pcfg), Merger(pcfg), Refiner(pcfg)]
    if "stages" in case:     # backbone stages after the last requested one do not exist (timm FeatureListNet / notebook 40,339,770 KAT)
        last = max(case["stages"])
        keys = [k for k in pnets[0].state_dict() if k.startswith("swin_transformer.model.layers_")]
        assert {int(k.split("layers_")[1][0]) for k in keys} == set(range(last + 1))
    for p, o in zip(pnets, onets):
        p.load_state_dict(o.state_dict())       # strict: the variant has exactly the reference's parameters
        p.to(dev).train()
        p.stochastic = False
    g = torch.Generator().manual_seed(1)
    x = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)
    gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.1).float()
    with torch.no_grad():
        total_o, _, _, _, refined_o = O.train_step_loss(onets, ocfg, x, gt)
    out = {}
    for mode in ("f32", "bf16"):
        S.set_math(mode)
        if mode == "bf16":
            S.set_storage("bf16")
        try:
            for p in pnets:
                p.zero_grad(set_to_none=True)
            raw, vol = pnets[1](pnets[0](x.to(dev)))
            merged = pnets[2](raw, vol)
            refined = pnets[3](merged)
            total = bce(merged, gt.to(dev)) + bce(refined, gt.to(dev))
            total.backward()
            finite = all(bool(torch.isfinite(p.grad).all()) for n in pnets for p in n.parameters() if p.grad is not None)
            n_grads = sum(p.grad is not None for n in pnets for p in n.parameters())
            out[mode] = (float(total.detach()), float((refined.detach().cpu() - refined_o).abs().max()), finite, n_grads)
        finally:
            S.set_math("f32")
    l32, e32, f32, n32 = out["f32"]
    l16, _, f16, _ = out["bf16"]
    ref = float(total_o)
    assert n32 == sum(1 for n in pnets for _ in n.parameters())           # every parameter of the variant received a gradient
    assert abs(l32 - ref) < 1e-3 and e32 < 2e-3 * max(1.0, float(refined_o.abs().max())) and f32      # fp32: 1e-3 (north_star)
    assert f16 and abs(l16 - ref) < 3e-2 * max(1.0, abs(ref))


def test_swin_b_encoder_matches_the_golden_vector(dev):
    """Swin-B encoder variant against the committed fp32 golden features (tests/golden/make_swin_b_pin.py: oracle Encoder(variant="base"),
    whose backbone is pinned against transformers' SwinModel(embed_dim=128, depths=[2,2,18,2], num_heads=[4,8,16,32]) to 1.4e-5)."""
    import json
    import os
    import numpy as np
    from swinvox_amd import goldens
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    case = json.load(open(os.path.join(gdir, "manifest.json")))["cases"]["swin_b_B1_V2"]
    gold = torch.from_numpy(np.load(os.path.join(gdir, "case_swin_b_B1_V2.npz"))["features"])
    enc = Encoder(S.default_cfg(), variant="base")
    goldens.seeded_fill_(enc, case["weights_seed"])
    assert sum(p.numel() for p in enc.parameters()) == 104832376
    enc.to(dev).eval()
    x = goldens.synth_images(1, 2, case["seed"]).to(dev)
    S.set_math("f32")
    with torch.no_grad():
        f = enc(x).cpu()
    assert float((f - gold).abs().max()) < 1e-3 * float(gold.abs().max())
    S.set_math("bf16")
    S.set_storage("bf16")
    try:
        with torch.no_grad():
            f16 = enc(x).cpu()
    finally:
        S.set_math("f32")
    assert bool(torch.isfinite(f16).all()) and float((f16 - gold).abs().mean()) < 0.1 * float(gold.abs().mean()) + 1e-2


def test_swin_b_encoder_with_fp8_attention(dev):
    """BASELINE configuration 5 as a whole module: the Swin-B encoder (heads 4 / 8 / 16 / 32) with the window-attention forward's QK^T / PV
    on fp8 (OCP e4m3) MFMA operands (`set_attention_fp8(True)`, bf16 everywhere else), eval forward against the committed fp32 golden
    features of the Swin-B encoder, and one train-mode step (the backward keeps bf16 operands) for finite gradients on every parameter.
    Stated bound: the fp8 features stay within 1.25x the bf16 path's own mean deviation from the golden vector (+ 1e-2 of mean|gold|) -
    fp8 operands of the 49-key attention core must not add more than a quarter to what bf16 storage already costs on this weight set;
    and they must differ from the bf16 result (the fp8 kernel really ran)."""
    import json
    import os
    import numpy as np
    from swinvox_amd import goldens, ops
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    case = json.load(open(os.path.join(gdir, "manifest.json")))["cases"]["swin_b_B1_V2"]
    gold = torch.from_numpy(np.load(os.path.join(gdir, "case_swin_b_B1_V2.npz"))["features"])
    enc = Encoder(S.default_cfg(), variant="base")
    goldens.seeded_fill_(enc, case["weights_seed"])
    enc.to(dev).eval()
    x = goldens.synth_images(1, 2, case["seed"]).to(dev)
    S.set_math("bf16")
    S.set_storage("bf16")
    try:
        with torch.no_grad():
            f16 = enc(x).cpu()
            S.set_attention_fp8(True)
            from swinvox_amd import hip as _hip
            assert ops.attention_math() == _hip.MATH_FP8
            f8 = enc(x).cpu()
        enc.train()
        enc.stochastic = False
        enc.zero_grad(set_to_none=True)
        enc(x).square().mean().backward()
        grads_ok = all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in enc.parameters())
    finally:
        S.set_attention_fp8(False)
        S.set_math("f32")
    gm = float(gold.abs().mean())
    e16, e8 = float((f16 - gold).abs().mean()), float((f8 - gold).abs().mean())
    print(f"Swin-B encoder vs golden: mean|d| bf16 {e16:.4e}, bf16+fp8 attention {e8:.4e}, mean|gold| {gm:.4e}, fp8-vs-bf16 {float((f8 - f16).abs().mean()):.4e}")
    assert bool(torch.isfinite(f8).all()) and grads_ok
    assert float((f8 - f16).abs().max()) > 0.0
    assert e8 <= 1.25 * e16 + 1e-2 * gm, (e8, e16, gm)


def test_documented_refusals(dev):
    """INTEGRATION.md section 3b, row by row: what the reference accepts and this library refuses must raise - never run something else."""
    cfg = S.default_cfg()
    cfg.NETWORK.ATT_SPATIAL_DOWNSAMPLE_RATIO = 3                      # 7 -> 2 -> 7 grid: not built (ratios 1 and 2 are)
    enc = Encoder(cfg).to(dev).eval()
    with pytest.raises(NotImplementedError, match="ATT_SPATIAL_DOWNSAMPLE_RATIO"), torch.no_grad():
        enc(torch.zeros(1, 1, 3, 224, 224, device=dev))
    enc = Encoder(S.default_cfg()).to(dev).eval()
    with pytest.raises(AssertionError, match="224"), torch.no_grad():   # the reference resizes inside the Swin wrapper (swin_transformer.py:74-75)
        enc(torch.zeros(1, 1, 3, 128, 128, device=dev))
    with pytest.raises(RuntimeError, match="n_views"), torch.no_grad():  # cross-view attention keeps all views of a sample in LDS: V <= 32
        enc(torch.zeros(1, 33, 3, 224, 224, device=dev))
    with pytest.raises(RuntimeError, match="GPU"):                       # no CPU path
        enc(torch.zeros(1, 1, 3, 224, 224))
