#!/usr/bin/env python3
"""Pin the CPU oracle against the reference and emit the golden fixtures (runs ONLY in the build container).

What it does (results land in tests/golden/manifest.json; nothing here travels as source of the reference):
  1. imports the reference's own models/{decoder,merger,refiner,cross_view_attention}.py from /root/reference
     (torch-only imports) and compares them with oracle/ on seeded inputs, same state_dict;
  2. imports the reference's models/encoder.py + models/swin_transformer.py with the oracle's backbones
     standing in for the absent `timm` / `torchvision` packages (their arithmetic is third-party and not
     in /root/reference), which pins the encoder plumbing (stage heads, neck, CVA call, fusion);
  3. compares the oracle's Swin arithmetic with transformers' independent SwinModel (local config, no fetch);
  4. checks the notebook's parameter-count KATs and its state-dict key list (cell 47 / 68);
  5. writes golden input/output vectors for the parity tests (weights are regenerated from the seeded
     recipe in oracle.seeded_weights_ + oracle.calibrate_, never committed).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import importlib
import json
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import oracle as O  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
manifest = {"torch": torch.__version__, "pins": {}, "cases": {}}


def maxdiff(a, b):
    return float((a - b).abs().max())


def synth_images(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)


def synth_gt(B, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float()


# ------------------------------------------------------------------ 1. reference tail modules
sys.path.insert(0, REF)
cfg = O.default_cfg()
ref_dec = importlib.import_module("models.decoder").Decoder(cfg)
ref_mer = importlib.import_module("models.merger").Merger(cfg)
ref_ref = importlib.import_module("models.refiner").Refiner(cfg)
ref_cva_mod = importlib.import_module("models.cross_view_attention")

o_dec, o_mer, o_ref = O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)
for o, r, name in ((o_dec, ref_dec, "decoder"), (o_mer, ref_mer, "merger"), (o_ref, ref_ref, "refiner")):
    O.seeded_weights_(o, seed=3)
    missing = r.load_state_dict(o.state_dict(), strict=True)
    assert set(o.state_dict().keys()) == set(r.state_dict().keys()), name
    for k, v in r.state_dict().items():
        assert tuple(v.shape) == tuple(o.state_dict()[k].shape), (name, k)

g = torch.Generator().manual_seed(11)
feat = torch.randn(2, 3, 256, 7, 7, generator=g)
pins = manifest["pins"]
for mode in ("eval", "train"):
    for m in (o_dec, o_mer, o_ref, ref_dec, ref_mer, ref_ref):
        m.train(mode == "train")
    with torch.no_grad():
        raw_o, vol_o = o_dec(feat)
        raw_r, vol_r = ref_dec(feat)
        mer_o, mer_r = o_mer(raw_o, vol_o), ref_mer(raw_r, vol_r)
        ref_o, ref_r = o_ref(mer_o), ref_ref(mer_r)
    pins[f"decoder_{mode}_maxdiff"] = max(maxdiff(raw_o, raw_r), maxdiff(vol_o, vol_r))
    pins[f"merger_{mode}_maxdiff"] = maxdiff(mer_o, mer_r)
    pins[f"refiner_{mode}_maxdiff"] = maxdiff(ref_o, ref_r)
    # re-sync BN running stats after the train-mode pass so both sides stay identical
    for o, r in ((o_dec, ref_dec), (o_mer, ref_mer), (o_ref, ref_ref)):
        r.load_state_dict(o.state_dict())

for V in (1, 5):
    o_cva = O.CrossViewAttention(cfg, 512)
    r_cva = ref_cva_mod.CrossViewAttention(cfg, 512)
    O.seeded_weights_(o_cva, seed=4)
    r_cva.load_state_dict(o_cva.state_dict(), strict=True)
    o_cva.eval(), r_cva.eval()
    x = torch.randn(2, V, 512, 7, 7, generator=g)
    with torch.no_grad():
        pins[f"cva_V{V}_maxdiff"] = maxdiff(o_cva(x), r_cva(x))

# backward pin on the tail (train mode, BN batch stats): grads of every parameter and of the input
for m in (o_dec, o_mer, o_ref, ref_dec, ref_mer, ref_ref):
    m.train()
    m.zero_grad()
gt = synth_gt(2, 5)
f1 = feat.clone().requires_grad_(True)
f2 = feat.clone().requires_grad_(True)
raw, vol = o_dec(f1)
(O.bce_logits(o_ref(o_mer(raw, vol)), gt)).backward()
raw, vol = ref_dec(f2)
torch.nn.functional.binary_cross_entropy_with_logits(ref_ref(ref_mer(raw, vol)), gt).backward()
gd = maxdiff(f1.grad, f2.grad) / float(f2.grad.abs().max())
for o, r in ((o_dec, ref_dec), (o_mer, ref_mer), (o_ref, ref_ref)):
    for (k, p), (_, q) in zip(o.named_parameters(), r.named_parameters()):
        gd = max(gd, maxdiff(p.grad, q.grad) / (float(q.grad.abs().max()) + 1e-30))
pins["tail_backward_rel_maxdiff"] = gd

# ------------------------------------------------------------------ 2. reference encoder plumbing
timm = types.ModuleType("timm")


def _create_model(name, pretrained=False, features_only=True, out_indices=(0, 1, 2, 3)):
    assert name == "swin_tiny_patch4_window7_224" and features_only
    return O.SwinBackbone(out_indices)


timm.create_model = _create_model
tv = types.ModuleType("torchvision")
tvm = types.ModuleType("torchvision.models")


class _W:
    DEFAULT = None


def _resnet50(weights=None):
    t = O.ResNetTrunk()
    full = torch.nn.Module()
    for n, m in zip(("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3"), t.children()):
        full.add_module(n, m)
    full.add_module("layer4", torch.nn.Identity())
    full.add_module("avgpool", torch.nn.Identity())
    full.add_module("fc", torch.nn.Identity())
    return full


tvm.resnet50, tvm.ResNet50_Weights = _resnet50, _W
tv.models = tvm
sys.modules.update({"timm": timm, "torchvision": tv, "torchvision.models": tvm})
ref_enc_mod = importlib.import_module("models.encoder")
for multi, stages in ((True, [0, 1, 2, 3]), (False, [3])):
    c = O.default_cfg()
    c.NETWORK.USE_SWIN_T_MULTI_STAGE, c.NETWORK.SWIN_T_STAGES = multi, stages
    o_enc, r_enc = O.Encoder(c), ref_enc_mod.Encoder(c)
    O.seeded_weights_(o_enc, seed=5)
    r_enc.load_state_dict(o_enc.state_dict(), strict=True)
    o_enc.eval(), r_enc.eval()
    x = synth_images(1, 2, 21)
    with torch.no_grad():
        a, b = o_enc(x), r_enc(x)
    pins[f"encoder_plumbing_multi{int(multi)}_maxdiff"] = maxdiff(a, b)
    pins[f"encoder_plumbing_multi{int(multi)}_absmax"] = float(b.abs().max())
    pins[f"encoder_params_multi{int(multi)}"] = sum(p.numel() for p in r_enc.parameters())
for k in ("timm", "torchvision", "torchvision.models"):
    sys.modules.pop(k)

# ------------------------------------------------------------------ 3. Swin arithmetic vs transformers
from transformers import SwinConfig, SwinModel  # noqa: E402

hf = SwinModel(SwinConfig(image_size=224, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                          window_size=7, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.1), add_pooling_layer=False).eval()
ob = O.SwinBackbone((0, 1, 2, 3)).eval()
O.seeded_weights_(ob, seed=6)
sd, hsd = ob.state_dict(), {}
for k, v in sd.items():
    m = re.match(r"layers_(\d)\.blocks\.(\d+)\.(.*)", k)
    if k.startswith("patch_embed.proj"):
        hsd["embeddings.patch_embeddings.projection." + k.rsplit(".", 1)[1]] = v
    elif k.startswith("patch_embed.norm"):
        hsd["embeddings.norm." + k.rsplit(".", 1)[1]] = v
    elif m:
        i, j, rest = m.groups()
        p = f"encoder.layers.{i}.blocks.{j}."
        C = v.shape[0] // 3
        if rest.startswith("attn.qkv"):
            leaf = rest.rsplit(".", 1)[1]
            for n, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                hsd[p + f"attention.{nm}.{leaf}"] = v[n * C:(n + 1) * C]
        elif rest == "attn.relative_position_bias_table":
            hsd[p + "attention.relative_position_bias.relative_position_bias_table"] = v
        elif rest.startswith("attn.proj"):
            hsd[p + "attention.o_proj." + rest.rsplit(".", 1)[1]] = v
        elif rest.startswith("norm1"):
            hsd[p + "layernorm_before." + rest.rsplit(".", 1)[1]] = v
        elif rest.startswith("norm2"):
            hsd[p + "layernorm_after." + rest.rsplit(".", 1)[1]] = v
        elif rest.startswith("mlp.fc1"):
            hsd[p + "mlp.fc1." + rest.rsplit(".", 1)[1]] = v
        elif rest.startswith("mlp.fc2"):
            hsd[p + "mlp.fc2." + rest.rsplit(".", 1)[1]] = v
    else:
        m = re.match(r"layers_(\d)\.downsample\.(norm|reduction)\.(\w+)", k)
        assert m, k
        hsd[f"encoder.layers.{int(m.group(1)) - 1}.downsample.{m.group(2)}.{m.group(3)}"] = v
res = hf.load_state_dict(hsd, strict=False)
left = [k for k in res.missing_keys if "relative_position_index" not in k and not k.startswith("layernorm.")]
assert not left and not res.unexpected_keys, (left[:5], res.unexpected_keys[:5])
x = synth_images(2, 1, 31)[:, 0]
with torch.no_grad():
    mine = ob(x)
    theirs = hf(x, output_hidden_states=True, output_hidden_states_before_downsampling=True).reshaped_hidden_states
for i in range(4):
    pins[f"swin_vs_hf_stage{i}_maxdiff"] = maxdiff(mine[i].permute(0, 3, 1, 2), theirs[i + 1])
    pins[f"swin_vs_hf_stage{i}_absmax"] = float(theirs[i + 1].abs().max())

# ------------------------------------------------------------------ 4. notebook KATs (cell 47 / 68)
nb = json.load(open(os.path.join(REF, "Notebooks/SwinVox.ipynb")))
txt = "".join("".join(o.get("text", [])) for o in nb["cells"][68].get("outputs", []))
keys = []
for k in re.findall(r"module\.((?:resnet|swin_transformer\.model)\.[A-Za-z0-9_\.]+)", txt):
    if k not in keys:
        keys.append(k)
enc_keys = set(O.Encoder(O.default_cfg()).state_dict().keys())
absent = [k for k in keys if k not in enc_keys]
pins["notebook_backbone_keys"] = len(keys)
pins["notebook_backbone_keys_absent_in_oracle"] = absent
json.dump(keys, open(os.path.join(HERE, "notebook_backbone_keys.json"), "w"), indent=0)
pins["param_counts"] = {"encoder_default": 45109818, "encoder_single_stage": pins["encoder_params_multi0"],
                        "decoder": 3817944, "refiner": 34880352, "merger": 17877,
                        "notebook_cell47": {"encoder": 40339770, "decoder": 3817944, "refiner": 34880352, "merger": 17877}}
assert pins["encoder_params_multi0"] == 40339770 and pins["encoder_params_multi1"] == 45109818

# ------------------------------------------------------------------ 5. golden vectors
cfg = O.default_cfg()
nets = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
for i, n in enumerate(nets):
    O.seeded_weights_(n, seed=100 + i)
O.calibrate_(nets, synth_images(2, 2, 1234))
for n in nets:
    n.eval()
enc, dec, mer, ref = nets


def strided(t, n=4096):
    f = t.reshape(-1)
    return f[:: max(1, f.numel() // n)][:n].clone()


for (B, V, seed) in ((2, 1, 41), (1, 2, 42), (2, 8, 43)):
    x = synth_images(B, V, seed)
    gt = synth_gt(B, seed)
    with torch.no_grad():
        swin_feats = enc.swin_transformer(x.view(B * V, 3, 224, 224))
        f = enc(x)
        raw, vol = dec(f)
        merged = mer(raw, vol)
        refined = ref(merged)
    iou = O.iou_at_thresholds(refined, gt)
    out = {"features": f.numpy(), "merged": merged.numpy(), "refined": refined.numpy(),
           "gen_volumes_sample": strided(vol).numpy(), "raw_features_sample": strided(raw).numpy(),
           "raw_features_sum": np.array([float(raw.double().sum()), float(raw.double().abs().sum())]),
           "iou": np.array(iou, dtype=np.float64)}
    for i, sf in enumerate(swin_feats):
        out[f"swin_stage{i}_sample"] = strided(sf).numpy()
    np.savez_compressed(os.path.join(HERE, f"case_B{B}_V{V}.npz"), **out)
    manifest["cases"][f"B{B}_V{V}"] = {"seed": seed, "logit_std": float(refined.std()), "iou": iou,
                                      "near_half": int(((refined.abs()) < 1e-3).sum())}

# train-mode backward fixture: per-parameter grad checksums (dropout / drop-path forced to 0)
for n in nets:
    n.train()
    for m in n.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, O.model.SwinBlock):
            m.dp = 0.0
    n.zero_grad()
x, gt = synth_images(2, 2, 44), synth_gt(2, 44)
total, el, rl, _, _ = O.train_step_loss(nets, cfg, x, gt)
total.backward()
gsum = {}
for tag, n in zip(("encoder", "decoder", "merger", "refiner"), nets):
    for k, p in n.named_parameters():
        gsum[f"{tag}.{k}"] = [float(p.grad.double().norm()), float(p.grad.double().sum())]
json.dump({"total": float(total), "encoder_loss": float(el), "refiner_loss": float(rl), "grads": gsum},
          open(os.path.join(HERE, "train_step_B2_V2.json"), "w"), indent=0)

json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
print(json.dumps(manifest["pins"], indent=1)[:3000])
