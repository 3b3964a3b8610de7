"""Module-level parity (through the C ABI) of the HIP Encoder/Decoder/Merger/Refiner against the CPU oracle and the
committed golden vectors.  Tolerances follow BASELINE.json: 1e-3 fp32 (relative to max|ref| per module, absolute
on final logits), thresholded occupancy bit-exact outside a 1e-3 band around logit(th)."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
import swinvox_amd as S  # noqa: E402
from swinvox_amd import ops  # noqa: E402
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def synth_images(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)


def synth_gt(B, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float()


def grad_report(items, factor=4.0, floor_l1=3e-3, floor_max=2.5e-2):
    """items: (name, hip_grad, oracle_fp32_grad, oracle_fp64_grad); the fp64 run of the oracle is the truth.

    Two fp32 forwards (HIP vs CPU) differ by rounding, so a max-pool arg-max or a (Leaky)ReLU mask occasionally flips
    at a near-tie; one flip re-routes ONE gradient element and moves a whole conv-weight gradient by O(1%) of its max
    (measured: scripts/grad_mask_flip_study.py - the contraction kernels themselves are exact to 3e-7 on the same
    tensors).  Train-mode BatchNorm over a handful of images additionally makes some ResNet gradients ill-conditioned:
    the CPU fp32 oracle itself is up to 17% off the fp64 truth there.  Hence: the L1-relative error must be within
    `factor` x the CPU-fp32 oracle's own L1 error + floor_l1, and the max-norm error within factor x e32 + floor_max (round 3: floors
    3e-3 / 2.5e-2, down from 5e-3 / 5e-2 - the largest excess measured over the whole train step is 1.9e-3 / 1.3e-2, one re-routed
    max-pool / ReLU element in `encoder.layer3.0.weight`; the tail modules alone stay below 4e-4 / 8e-4);
    analytically-zero gradients (conv biases in front of a train-mode BatchNorm) are compared absolutely."""
    bad = []
    worst_l1, worst_mx = (0.0, ""), (0.0, "")          # largest excess over factor x the CPU-fp32 error: what the floors have to cover
    for name, gh, g32, g64 in items:
        gh, g32, g64 = gh.detach().cpu().double(), g32.detach().double(), g64.detach()
        scale = float(g64.abs().max())
        if scale < 1e-9:
            if float((gh - g64).abs().max()) >= 1e-6:
                bad.append((name, "zero-grad", float((gh - g64).abs().max())))
            continue
        l1 = float(g64.abs().sum())
        e_hip_l1, e_32_l1 = float((gh - g64).abs().sum()) / l1, float((g32 - g64).abs().sum()) / l1
        e_hip_mx, e_32_mx = float((gh - g64).abs().max()) / scale, float((g32 - g64).abs().max()) / scale
        worst_l1 = max(worst_l1, (e_hip_l1 - factor * e_32_l1, name))
        worst_mx = max(worst_mx, (e_hip_mx - factor * e_32_mx, name))
        if e_hip_l1 > factor * e_32_l1 + floor_l1 or e_hip_mx > factor * e_32_mx + floor_max:
            bad.append((name, e_hip_l1, e_32_l1, e_hip_mx, e_32_mx))
    print(f"grad_report: largest excess over {factor} x CPU-fp32 error: L1 {worst_l1[0]:.2e} ({worst_l1[1]}), max-norm {worst_mx[0]:.2e} ({worst_mx[1]})")
    return bad


def no_stochastic(nets):
    for n in nets:
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, O.model.SwinBlock):
                m.dp = 0.0


@pytest.fixture(scope="module")
def nets(dev):
    """Oracle nets with the golden recipe (seeded weights + calibration) and HIP nets holding the same state."""
    torch.manual_seed(0)
    cfg = O.default_cfg()
    onets = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
    for i, n in enumerate(onets):
        O.seeded_weights_(n, seed=100 + i)
    O.calibrate_(onets, synth_images(2, 2, 1234))
    for n in onets:
        n.eval()
    pcfg = S.default_cfg()
    pnets = [Encoder(pcfg), Decoder(pcfg), Merger(pcfg), Refiner(pcfg)]
    for p, o in zip(pnets, onets):
        p.load_state_dict(o.state_dict(), strict=True)
        p.to(dev).eval()
    ops.set_math("f32")
    return onets, pnets


def test_tail_modules_eval_and_train(dev, nets):
    onets, pnets = nets
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(2, 3, 256, 7, 7, generator=g)
    for mode in ("eval", "train"):
        # fresh copies so BN running stats of the shared fixture stay untouched
        import copy
        o_dec, o_mer, o_ref = (copy.deepcopy(n) for n in onets[1:])
        p_dec, p_mer, p_ref = Decoder(S.default_cfg()), Merger(S.default_cfg()), Refiner(S.default_cfg())
        for p, o in ((p_dec, o_dec), (p_mer, o_mer), (p_ref, o_ref)):
            p.load_state_dict(o.state_dict()); p.to(dev)
            p.train(mode == "train"); o.train(mode == "train")
        with torch.no_grad():
            raw_o, vol_o = o_dec(feat)
            mer_o = o_mer(raw_o, vol_o)
            ref_o = o_ref(mer_o)
            raw_p, vol_p = p_dec(feat.to(dev))
            mer_p = p_mer(raw_p, vol_p)
            ref_p = p_ref(mer_p)
        assert raw_p.shape == raw_o.shape and vol_p.shape == vol_o.shape
        assert rel(raw_p, raw_o) < 1e-3 and rel(vol_p, vol_o) < 1e-3, mode
        assert rel(mer_p, mer_o) < 1e-3 and rel(ref_p, ref_o) < 1e-3, mode
        if mode == "train":   # running statistics were updated identically
            for (k, a), (_, b) in zip(p_ref.state_dict().items(), o_ref.state_dict().items()):
                if "running" in k:
                    assert rel(a, b) < 1e-3, k


def test_tail_backward(dev, nets):
    import copy
    onets, _ = nets
    g = torch.Generator().manual_seed(4)
    B, V = 2, 2
    o_dec, o_mer, o_ref = (copy.deepcopy(n).train() for n in onets[1:])
    p_dec, p_mer, p_ref = Decoder(S.default_cfg()), Merger(S.default_cfg()), Refiner(S.default_cfg())
    for p, o in ((p_dec, o_dec), (p_mer, o_mer), (p_ref, o_ref)):
        p.load_state_dict(o.state_dict()); p.to(dev).train()
    feat = torch.randn(B, V, 256, 7, 7, generator=g)
    gt = synth_gt(B, 7)
    f1 = feat.clone().requires_grad_(True)
    raw, vol = o_dec(f1)
    mer = o_mer(raw, vol)
    loss_o = O.bce_logits(mer, gt) + O.bce_logits(o_ref(mer), gt)
    loss_o.backward()
    f2 = feat.clone().to(dev).requires_grad_(True)
    raw, vol = p_dec(f2)
    mer = p_mer(raw, vol)
    loss_p = torch.nn.functional.binary_cross_entropy_with_logits(mer, gt.to(dev)) + \
        torch.nn.functional.binary_cross_entropy_with_logits(p_ref(mer), gt.to(dev))
    loss_p.backward()
    assert abs(float(loss_p) - float(loss_o)) < 1e-4
    # fp64 run of the same oracle = ground truth; the HIP fp32 gradients must be as close to it as CPU fp32 is
    # (train-mode BatchNorm over a handful of images makes some gradients ill-conditioned in fp32)
    d_dec, d_mer, d_ref = (copy.deepcopy(n).double() for n in (o_dec, o_mer, o_ref))
    for n in (d_dec, d_mer, d_ref):
        n.zero_grad()
    f3 = feat.double().requires_grad_(True)
    raw, vol = d_dec(f3)
    mer = d_mer(raw, vol)
    (O.bce_logits(mer, gt.double()) + O.bce_logits(d_ref(mer), gt.double())).backward()
    bad = grad_report([("feat", f2.grad, f1.grad, f3.grad)] + [
        (f"{type(p).__name__}.{k}", a.grad, b.grad, c.grad)
        for p, o, d in ((p_dec, o_dec, d_dec), (p_mer, o_mer, d_mer), (p_ref, o_ref, d_ref))
        for (k, a), (_, b), (_, c) in zip(p.named_parameters(), o.named_parameters(), d.named_parameters())])
    assert not bad, bad[:10]


@pytest.mark.parametrize("B,V", [(2, 1), (1, 2), (2, 8)])    # (2, 8): n_views of the headline configuration (BASELINE config 3)
def test_full_forward_vs_oracle_and_golden(dev, nets, B, V):
    onets, pnets = nets
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"][f"B{B}_V{V}"]
    x, gt = synth_images(B, V, man["seed"]), synth_gt(B, man["seed"])
    gold = np.load(os.path.join(GOLD, f"case_B{B}_V{V}.npz"))
    enc, dec, mer, ref = pnets
    with torch.no_grad():
        f = enc(x.to(dev))
        raw, vol = dec(f)
        merged = mer(raw, vol)
        refined = ref(merged)
        of = onets[0](x)
        oraw, ovol = onets[1](of)
        omerged = onets[2](oraw, ovol)
        orefined = onets[3](omerged)
    # HIP vs oracle on this machine
    assert rel(f, of) < 1e-3
    assert rel(raw, oraw) < 1e-3 and rel(vol, ovol) < 1e-3 and rel(merged, omerged) < 1e-3
    assert float((refined.cpu() - orefined).abs().max()) < 1e-3 * max(1.0, float(orefined.abs().max()))
    # HIP vs the committed golden vectors (made in the build container from the pinned oracle)
    assert rel(f, torch.from_numpy(gold["features"])) < 1e-3
    assert rel(refined, torch.from_numpy(gold["refined"])) < 2e-3
    # thresholded occupancy: bit-exact outside a 1e-3 band around logit(th); IoU within 1e-3
    excluded = 0
    for th in (0.2, 0.3, 0.4, 0.5):
        lt = math.log(th / (1 - th))
        band = (orefined - lt).abs() <= 1e-3
        excluded += int(band.sum())
        a = torch.sigmoid(refined.cpu()) >= th
        b = torch.sigmoid(orefined) >= th
        assert bool((a == b)[~band].all())
    assert excluded < 0.01 * orefined.numel() * 4
    iou_p = O.iou_at_thresholds(refined.cpu(), gt)
    assert np.abs(np.array(iou_p) - gold["iou"]).max() < 1e-3


def test_bf16_end_to_end_at_the_headline_views(dev, nets):
    """The BENCHMARKED mode (bf16 MFMA operands + bf16 activation storage) through the WHOLE pipeline at n_views = 8 on the golden
    inputs of case_B2_V8 - encoder features included, nothing taken from the oracle in between - against the fp32 oracle and the
    committed golden vectors: max|dlogit|, occupancy flips outside a band around logit(th), |dIoU|.  The measured numbers are
    written to gpurun_out/bf16_e2e_V8.json (quoted in DESIGN.md section 3 and in bench.py's line); the bounds below are what this
    weight set allows: the calibrated golden weights are seeded at default-init scale, not trained, and the CPU oracle under
    torch.autocast(bfloat16) is itself reported beside the HIP result as the yardstick."""
    onets, pnets = nets
    B, V = 2, 8
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"][f"B{B}_V{V}"]
    x, gt = synth_images(B, V, man["seed"]), synth_gt(B, man["seed"])
    gold = np.load(os.path.join(GOLD, f"case_B{B}_V{V}.npz"))
    g_ref, g_feat = torch.from_numpy(gold["refined"]), torch.from_numpy(gold["features"])
    import copy
    with torch.no_grad():
        ocp = [copy.deepcopy(n).eval() for n in onets]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            raw_c, vol_c = ocp[1](ocp[0](x))
            ref_c = ocp[3](ocp[2](raw_c, vol_c)).float()
    ops.set_math("bf16")
    ops.set_storage("bf16")
    try:
        with torch.no_grad():
            f = pnets[0](x.to(dev))
            raw, vol = pnets[1](f)
            merged = pnets[2](raw, vol)
            refined = pnets[3](merged).cpu()
    finally:
        ops.set_math("f32")

    def report(r):
        d = (r - g_ref).abs()
        flips, band_n = 0, 0
        for th in (0.2, 0.3, 0.4, 0.5):
            lt = math.log(th / (1 - th))
            band = (g_ref - lt).abs() <= 5e-2
            band_n += int(band.sum())
            flips += int(((torch.sigmoid(r) >= th) != (torch.sigmoid(g_ref) >= th))[~band].sum())
        iou = np.array(O.iou_at_thresholds(r, gt))
        return {"max_abs_dlogit": float(d.max()), "mean_abs_dlogit": float(d.mean()), "logit_absmax": float(g_ref.abs().max()),
                "flips_outside_5e-2_band": flips, "voxels_in_band": band_n, "voxel_threshold_pairs": int(g_ref.numel() * 4),
                "max_abs_dIoU": float(np.abs(iou - gold["iou"]).max())}

    out = {"case": f"B{B}_V{V}", "feature_rel_err_hip_bf16": rel(f, g_feat), "hip_bf16": report(refined), "cpu_autocast_bf16": report(ref_c)}
    os.makedirs(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", "bf16_e2e_V8.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("bf16 end-to-end V=8:", json.dumps(out))
    h, c = out["hip_bf16"], out["cpu_autocast_bf16"]
    assert bool(torch.isfinite(refined).all())
    # north_star: IoU@32^3 within 1e-3 of the reference on fixed inputs - asserted as such for the benchmarked mode (measured 5-8e-4 over
    # the boxes of rounds 2-3).  Logits: this weight set (seeded at default-init scale and calibrated, not trained) amplifies bf16 rounding
    # through 12 Swin blocks for the HIP path and for CPU autocast alike, so they are bounded by the CPU-autocast deviation (1.2x + floors),
    # and so are the occupancy flips outside the band.
    assert h["max_abs_dIoU"] <= 1e-3, out
    assert h["mean_abs_dlogit"] <= 1.2 * c["mean_abs_dlogit"] + 1e-2, out
    assert h["max_abs_dlogit"] <= 1.2 * c["max_abs_dlogit"] + 5e-2, out
    assert h["flips_outside_5e-2_band"] <= 1.2 * c["flips_outside_5e-2_band"], out


def test_train_step_gradients_vs_oracle(dev, nets):
    """One full training step (core/train.py:226-272 semantics, dropout/drop-path off): loss and every gradient."""
    import copy
    onets, _ = nets
    B, V = 2, 2
    ocp = [copy.deepcopy(n).train() for n in onets]
    no_stochastic(ocp)
    pcfg = S.default_cfg()
    pn = [Encoder(pcfg), Decoder(pcfg), Merger(pcfg), Refiner(pcfg)]
    for p, o in zip(pn, ocp):
        p.load_state_dict(o.state_dict()); p.to(dev).train(); p.stochastic = False
    x, gt = synth_images(B, V, 44), synth_gt(B, 44)
    total_o, el_o, rl_o, _, _ = O.train_step_loss(ocp, O.default_cfg(), x, gt)
    total_o.backward()
    xd, gd = x.to(dev).clamp(-1, 1), gt.to(dev)
    raw, vol = pn[1](pn[0](xd))
    merged = pn[2](raw, vol)
    from swinvox_amd.losses import bce_with_logits       # the product's loss (sv_bce_logits), as bench.py / harness.py use it
    el = bce_with_logits(merged, gd)
    rl = bce_with_logits(pn[3](merged), gd)
    (el + rl).backward()
    assert abs(float(el) - float(el_o)) < 1e-4 and abs(float(rl) - float(rl_o)) < 1e-4
    gold = json.load(open(os.path.join(GOLD, "train_step_B2_V2.json")))
    assert abs(float(el + rl) - gold["total"]) < 1e-3
    # ground truth = the same oracle in fp64 (see test_tail_backward)
    o64 = [copy.deepcopy(n).double() for n in ocp]
    for n in o64:
        n.zero_grad()
    t64, _, _, _, _ = O.train_step_loss(o64, O.default_cfg(), x.double(), gt.double())
    t64.backward()
    bad = grad_report([(f"{tag}.{k}", a.grad, b.grad, c.grad)
                       for tag, p, o, d in zip(("encoder", "decoder", "merger", "refiner"), pn, ocp, o64)
                       for (k, a), (_, b), (_, c) in zip(p.named_parameters(), o.named_parameters(), d.named_parameters())])
    assert not bad, bad[:10]


def test_single_stage_config(dev):
    """USE_SWIN_T_MULTI_STAGE=False / SWIN_T_STAGES=[3] path (models/encoder.py:77,140) and V=1 cross-view attention."""
    cfg_o = O.default_cfg(); cfg_o.NETWORK.USE_SWIN_T_MULTI_STAGE = False; cfg_o.NETWORK.SWIN_T_STAGES = [3]
    cfg_p = S.default_cfg(); cfg_p.NETWORK.USE_SWIN_T_MULTI_STAGE = False; cfg_p.NETWORK.SWIN_T_STAGES = [3]
    o, p = O.Encoder(cfg_o), Encoder(cfg_p)
    O.seeded_weights_(o, seed=9)
    o.eval()
    p.load_state_dict(o.state_dict(), strict=True)
    p.to(dev).eval()
    x = synth_images(1, 1, 5)
    with torch.no_grad():
        assert rel(p(x.to(dev)), o(x)) < 1e-3


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_bf16_math(dev, nets, storage):
    """set_math("bf16") (MFMA bf16 operands, fp32 accumulate; LDS-halo stencils in the merger), with fp32 and with bf16
    activation storage (set_storage).
    Tail modules (well conditioned): forward within 2e-2 of the fp32 oracle on the oracle's own inputs, gradients within
    30 % L1 (train-mode BatchNorm over 4 images amplifies bf16 rounding in the backward).  Encoder: this weight set amplifies bf16 rounding (the CPU oracle under torch.autocast(bfloat16) is itself
    ~40 % off its fp32 result), so the HIP error is bounded by 1.5x that CPU-bf16 deviation; gradients must be finite."""
    import copy
    onets, _ = nets
    ocp = [copy.deepcopy(n).train() for n in onets]
    no_stochastic(ocp)
    pcfg = S.default_cfg()
    pn = [Encoder(pcfg), Decoder(pcfg), Merger(pcfg), Refiner(pcfg)]
    for p, o in zip(pn, ocp):
        p.load_state_dict(o.state_dict()); p.to(dev).train(); p.stochastic = False
    x, gt = synth_images(2, 2, 44), synth_gt(2, 44)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    oc2 = [copy.deepcopy(n) for n in ocp]
    with torch.no_grad():
        f_o = oc2[0](x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            f_ob = copy.deepcopy(ocp[0])(x).float()
    cpu_bf16_dev = rel(f_ob, f_o)
    feat = f_o.clone().requires_grad_(True)
    raw_o, vol_o = ocp[1](feat)
    mer_o = ocp[2](raw_o, vol_o)
    ref_o = ocp[3](mer_o)
    (bce(mer_o, gt) + bce(ref_o, gt)).backward()
    ops.set_math("bf16")
    ops.set_storage(storage)
    try:
        f_p = pn[0](x.to(dev))
        f_p.square().mean().backward()
        featd = f_o.clone().to(dev).requires_grad_(True)
        raw, vol = pn[1](featd)
        merged = pn[2](raw, vol)
        refined = pn[3](merged)
        (bce(merged, gt.to(dev)) + bce(refined, gt.to(dev))).backward()
    finally:
        ops.set_math("f32")
    assert rel(f_p, f_o) < 1.5 * cpu_bf16_dev + 2e-2, (rel(f_p, f_o), cpu_bf16_dev)
    assert rel(vol, vol_o) < 2e-2 and rel(merged, mer_o) < 2e-2 and rel(refined, ref_o) < 2e-2
    iou_p, iou_o = np.array(O.iou_at_thresholds(refined.detach().cpu(), gt)), np.array(O.iou_at_thresholds(ref_o.detach(), gt))
    assert np.abs(iou_p - iou_o).max() < 1e-2
    for k, a in pn[0].named_parameters():
        assert bool(torch.isfinite(a.grad).all()), k
    assert float((featd.grad.cpu() - feat.grad).abs().sum() / feat.grad.abs().sum()) < 0.3
    for p, o in ((pn[1], ocp[1]), (pn[2], ocp[2]), (pn[3], ocp[3])):
        for (k, a), (_, b) in zip(p.named_parameters(), o.named_parameters()):
            if float(b.grad.abs().max()) > 1e-8:
                l1 = float((a.grad.cpu() - b.grad).abs().sum() / b.grad.abs().sum())
                assert l1 < 0.3, (type(p).__name__, k, l1)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("which", ["encoder", "decoder", "merger", "refiner"])
def test_modules_read_no_uninitialised_memory(dev, which, mode):
    """Buffers are allocated with torch.empty wherever their producer writes every element that is read later (padded rows are
    written whole, or their padding is ignored).  Poison the allocator's free memory with NaN first: a read of memory nobody
    wrote turns the outputs / the gradients into NaN (found one in the fp32 path of the merger when this test was added)."""
    import swinvox_amd as S
    from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
    cfg = S.default_cfg()
    g = torch.Generator().manual_seed(5)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev).requires_grad_(True)
    if which == "encoder":
        m, ins = Encoder(cfg), [rnd(1, 2, 3, 224, 224)]
    elif which == "decoder":
        m, ins = Decoder(cfg), [rnd(2, 3, 256, 7, 7)]
    elif which == "merger":
        m, ins = Merger(cfg), [rnd(2, 3, 9, 32, 32, 32), rnd(2, 3, 32, 32, 32)]
    else:
        m, ins = Refiner(cfg), [rnd(3, 32, 32, 32)]
    m = m.to(dev).train()
    S.set_math(mode)
    if mode == "bf16":
        S.set_storage("bf16")
    try:
        for _ in range(2):
            torch.cuda.empty_cache()
            poison = [torch.full((64 << 20,), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(8)]   # 1 GB of NaN
            del poison                                   # back to the caching allocator, contents intact
            m.zero_grad(set_to_none=True)
            for t in ins:
                t.grad = None
            out = m(*ins)
            outs = [out] if isinstance(out, torch.Tensor) else list(out)
            sum(o.sum() for o in outs).backward()
            assert all(bool(torch.isfinite(o).all()) for o in outs)
            assert all(bool(torch.isfinite(t.grad).all()) for t in ins if t.grad is not None)
            assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    finally:
        S.set_math("f32")
