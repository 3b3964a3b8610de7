#!/usr/bin/env python3
"""Pins the oracle's CrossViewAttention at cfg.NETWORK.ATT_SPATIAL_DOWNSAMPLE_RATIO = 1 (no depth-wise down-sampling, attention on the
7x7 grid, no bilinear up-sampling: reference models/cross_view_attention.py:26-34,67-73,110-113) against the reference module itself,
imported read-only from /root/reference in the build container: forward (eval and train) and every gradient.  Results are merged into
tests/golden/manifest.json (`pins.cva_ds1_*`); tests/test_cpu_oracle_and_abi.py checks they are recorded as 0.0.

  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_cva_ratio_pin.py
"""
import importlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import oracle as O  # noqa: E402

sys.path.insert(0, REF)
ref_mod = importlib.import_module("models.cross_view_attention")
torch.manual_seed(0)
pins = {}
for ratio in (1, 2):
    cfg = O.default_cfg()
    cfg.NETWORK.ATT_SPATIAL_DOWNSAMPLE_RATIO = ratio
    for V in (1, 3):
        o, r = O.CrossViewAttention(cfg, 512), ref_mod.CrossViewAttention(cfg, 512)
        O.seeded_weights_(o, seed=40 + ratio)
        r.load_state_dict(o.state_dict(), strict=True)            # same keys: ratio 1 has no downsample_qkv.* in either
        g = torch.Generator().manual_seed(10 * ratio + V)
        x = torch.randn(2, V, 512, 7, 7, generator=g)
        worst = 0.0
        for mode in ("eval", "train"):
            o.train(mode == "train"), r.train(mode == "train")
            for m in (o, r):
                m.dropout.p = 0.0
            xo, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            yo, yr = o(xo), r(xr)
            worst = max(worst, float((yo - yr).abs().max()))
            if mode == "train":
                do = torch.randn(yo.shape, generator=g)
                o.zero_grad(), r.zero_grad()
                yo.backward(do), yr.backward(do)
                worst = max(worst, float((xo.grad - xr.grad).abs().max()))
                for (k, a), (_, b) in zip(o.named_parameters(), r.named_parameters()):
                    worst = max(worst, float((a.grad - b.grad).abs().max()))
        pins[f"cva_ds{ratio}_V{V}_fwd_bwd_maxdiff"] = worst
print(pins)
mp = os.path.join(HERE, "manifest.json")
man = json.load(open(mp))
man["pins"].update(pins)
json.dump(man, open(mp, "w"), indent=1)
