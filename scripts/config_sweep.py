#!/usr/bin/env python3
"""Robustness sweep on the GPU: the four modules, forward + backward, over the BASELINE configurations and config knobs
(n_views 1 / 5 / 24, odd batches, single-stage Swin, no cross-view attention), exact-fp32 mode checked against the CPU
oracle (loss and refined logits), bf16 mode checked for finiteness and closeness of the loss."""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O
import swinvox_amd as S
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0)
bce = torch.nn.functional.binary_cross_entropy_with_logits
CASES = [dict(B=2, V=1), dict(B=1, V=24), dict(B=3, V=5), dict(B=2, V=2, multi=False), dict(B=2, V=3, cva=False), dict(B=2, V=2, stages=[1, 3])]
for case in CASES:
    B, V = case["B"], case["V"]
    ocfg, pcfg = O.default_cfg(), S.default_cfg()
    for c in (ocfg, pcfg):
        if "multi" in case: c.NETWORK.USE_SWIN_T_MULTI_STAGE = case["multi"]
        if "cva" in case: c.NETWORK.USE_CROSS_VIEW_ATTENTION = case["cva"]
        if "stages" in case: c.NETWORK.SWIN_T_STAGES = case["stages"]
    torch.manual_seed(0)
    onets = [O.Encoder(ocfg), O.Decoder(ocfg), O.Merger(ocfg), O.Refiner(ocfg)]
    for i, n in enumerate(onets):
        O.seeded_weights_(n, seed=50 + i); n.train()
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout): m.p = 0.0
            if isinstance(m, O.model.SwinBlock): m.dp = 0.0
    pnets = [Encoder(pcfg), Decoder(pcfg), Merger(pcfg), Refiner(pcfg)]
    for p, o in zip(pnets, onets):
        p.load_state_dict(o.state_dict()); p.to(dev).train(); p.stochastic = False
    g = torch.Generator().manual_seed(1)
    x = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)
    gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.1).float()
    total_o, _, _, _, refined_o = O.train_step_loss(onets, ocfg, x, gt)
    out = {}
    for mode in ("f32", "bf16"):
        S.set_math(mode)
        if mode == "bf16": S.set_storage("bf16")
        try:
            for p in pnets: p.zero_grad(set_to_none=True)
            raw, vol = pnets[1](pnets[0](x.to(dev)))
            merged = pnets[2](raw, vol); refined = pnets[3](merged)
            total = bce(merged, gt.to(dev)) + bce(refined, gt.to(dev))
            total.backward(); torch.cuda.synchronize()
            finite = all(bool(torch.isfinite(p.grad).all()) for n in pnets for p in n.parameters() if p.grad is not None)
            out[mode] = (float(total), float((refined.detach().cpu() - refined_o.detach()).abs().max()), finite)
        finally:
            S.set_math("f32")
    l32, e32, f32 = out["f32"]; l16, e16, f16 = out["bf16"]
    ok = abs(l32 - float(total_o)) < 1e-3 and e32 < 2e-3 * max(1.0, float(refined_o.abs().max())) and f32 and f16 and abs(l16 - float(total_o)) < 3e-2 * max(1.0, abs(float(total_o)))
    print(("OK  " if ok else "FAIL"), case, f"oracle {float(total_o):.5f} f32 {l32:.5f} (dlogit {e32:.1e}) bf16 {l16:.5f} (dlogit {e16:.1e})", flush=True)
