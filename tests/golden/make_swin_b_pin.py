"""Pin of the oracle's Swin-B backbone (BASELINE config 5: embed 128, depths 2/2/18/2, heads 4/8/16/32 - not in the reference, whose
model name is hard-coded at models/swin_transformer.py:20) against transformers' independent SwinModel built from a local SwinConfig,
under the same key map as the Swin-T pin of make_golden.py.  Also stores a small golden vector of the Swin-B ENCODER (oracle Encoder with
variant="base") for the GPU test.  Run in the build container:  python tests/golden/make_swin_b_pin.py  (updates manifest.json)."""
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
from transformers import SwinConfig, SwinModel  # noqa: E402

torch.manual_seed(0)


def synth_images(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)


def to_hf(sd):
    hsd = {}
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
            leaf = rest.rsplit(".", 1)[-1]
            if rest.startswith("attn.qkv"):
                for n, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                    hsd[p + f"attention.{nm}.{leaf}"] = v[n * C:(n + 1) * C]
            elif rest == "attn.relative_position_bias_table":
                hsd[p + "attention.relative_position_bias.relative_position_bias_table"] = v
            elif rest.startswith("attn.proj"):
                hsd[p + "attention.o_proj." + leaf] = v
            elif rest.startswith("norm1"):
                hsd[p + "layernorm_before." + leaf] = v
            elif rest.startswith("norm2"):
                hsd[p + "layernorm_after." + leaf] = v
            else:
                hsd[p + rest] = v                      # mlp.fc1 / mlp.fc2
        else:
            m = re.match(r"layers_(\d)\.downsample\.(norm|reduction)\.(\w+)", k)
            assert m, k
            hsd[f"encoder.layers.{int(m.group(1)) - 1}.downsample.{m.group(2)}.{m.group(3)}"] = v
    return hsd


hf = SwinModel(SwinConfig(image_size=224, patch_size=4, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32],
                          window_size=7, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.1), add_pooling_layer=False).eval()
ob = O.SwinBackbone((0, 1, 2, 3), embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32)).eval()
O.seeded_weights_(ob, seed=16)
res = hf.load_state_dict(to_hf(ob.state_dict()), strict=False)
left = [k for k in res.missing_keys if "relative_position_index" not in k and not k.startswith("layernorm.")]
assert not left and not res.unexpected_keys, (left[:5], res.unexpected_keys[:5])
x = synth_images(1, 1, 33)[:, 0]
pins = {}
with torch.no_grad():
    mine = ob(x)
    theirs = hf(x, output_hidden_states=True, output_hidden_states_before_downsampling=True).reshaped_hidden_states
for i in range(4):
    pins[f"swin_b_vs_hf_stage{i}_maxdiff"] = float((mine[i].permute(0, 3, 1, 2) - theirs[i + 1]).abs().max())
    pins[f"swin_b_vs_hf_stage{i}_absmax"] = float(theirs[i + 1].abs().max())
pins["swin_b_backbone_params"] = sum(p.numel() for p in ob.parameters())

# golden vector of the Swin-B encoder (eval, fp32, seeded weights at default-init scale; inputs by seed)
cfg = O.default_cfg()
enc = O.Encoder(cfg, variant="base")
O.seeded_weights_(enc, seed=200)
enc.eval()
xi = synth_images(1, 2, 51)
with torch.no_grad():
    f = enc(xi)
np.savez_compressed(os.path.join(HERE, "case_swin_b_B1_V2.npz"), features=f.numpy().astype(np.float32))
pins["swin_b_encoder_params"] = sum(p.numel() for p in enc.parameters())
pins["swin_b_encoder_feature_absmax"] = float(f.abs().max())

man_path = os.path.join(HERE, "manifest.json")
man = json.load(open(man_path))
man["pins"].update(pins)
man["cases"]["swin_b_B1_V2"] = {"seed": 51, "weights_seed": 200}
json.dump(man, open(man_path, "w"), indent=1)
print(json.dumps(pins, indent=1))
