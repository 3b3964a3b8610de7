import copy, sys, torch
sys.path.insert(0, '.')
import oracle as O, swinvox_amd as S
from swinvox_amd import ops
from swinvox_amd.models import Encoder
from swinvox_amd.models import swin_transformer as ST
dev = torch.device('cuda:0')
torch.manual_seed(0)
g = torch.Generator().manual_seed(44)
x = (0.5 * torch.randn(1, 2, 3, 224, 224, generator=g)).clamp(-1, 1)
cfg = O.default_cfg()
oe = O.Encoder(cfg); O.seeded_weights_(oe, seed=100); oe.eval()
pe = Encoder(S.default_cfg()); pe.load_state_dict(oe.state_dict()); pe.to(dev).eval()
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return "max %.3e mean %.3e" % (float((a - b).abs().max() / (b.abs().max() + 1e-12)), float((a - b).abs().mean() / (b.abs().mean() + 1e-12)))
img = x.view(2, 3, 224, 224)
with torch.no_grad():
    r_o = oe.resnet(img)                                  # [2,1024,14,14]
    sw_o = oe.swin_transformer.model(img)                 # list NHWC
    heads_o = oe.swin_transformer(img)                    # list NCHW after LN
    # stem + layer1 separately
    s0 = oe.resnet[3](oe.resnet[2](oe.resnet[1](oe.resnet[0](img))))
    l1 = oe.resnet[4](s0)
for math in ("f32", "bf16"):
    ops.set_math(math)
    with torch.no_grad():
        out, tape = pe._fwd(x.to(dev), save=True)
    B, V, c_stem, mp_idx, c_blocks, res_feat, rr, swin_tape, neck, c_cva, c_post = tape
    print(math, "resnet trunk:", rel(res_feat.view(2, 14, 14, 1024).permute(0, 3, 1, 2), r_o))
    # layer1 output = input x of the first block of layer2 -> c_blocks[3] is (blk, ctx); ctx[0] = c1 ctx whose [0] is x
    x_l2 = c_blocks[3][1][0][0]
    print(math, "resnet layer1 out:", rel(x_l2.view(2, 56, 56, 256).permute(0, 3, 1, 2), l1))
    x_l1 = c_blocks[0][1][0][0]
    print(math, "stem+maxpool out:", rel(x_l1.view(2, 56, 56, 64).permute(0, 3, 1, 2), s0))
    for k, (f, red, cc) in enumerate(neck):
        hw = oe.swin_transformer.out_spatial[k]; C = oe.swin_transformer.out_channels[k]
        print(math, f"swin head {k}:", rel(f.view(2, hw, hw, C).permute(0, 3, 1, 2), heads_o[k]))
    # raw swin stage outputs: input of each head = tape heads xs
    for (si, hi, xs, wt, mr, p, seed, L) in swin_tape["heads"]:
        hw = oe.swin_transformer.out_spatial[hi]; C = oe.swin_transformer.out_channels[hi]
        print(math, f"swin stage {si} raw:", rel(xs.view(2, hw, hw, C), sw_o[hi]))
    emb = swin_tape["embed"][1]
    print(math, "patch-embed conv:", rel(emb.view(2, 56, 56, 96), oe.swin_transformer.model.patch_embed.proj(img).permute(0, 2, 3, 1)))
ops.set_math("f32")
print("---- neck / CVA / post-fusion")
import torch.nn.functional as F
with torch.no_grad():
    r = F.avg_pool2d(oe.resnet_reduce(r_o), 2, 2)
    reds = [red(f) for f, red in zip(heads_o, oe.swin_stage_reduces)]
    chains = []
    for k, (rd, dn) in enumerate(zip(reds, oe.swin_downsamples)):
        outs = [rd]; y = rd
        if not isinstance(dn, torch.nn.Identity):
            for j in range(0, len(dn), 3):
                y = dn[j + 2](dn[j + 1](dn[j](y))); outs.append(y)
        chains.append(outs)
    s_sum = sum(c[-1] for c in chains)
    cat_o = torch.cat((r, s_sum), 1)
    cva_o = oe.cross_view_attention(cat_o.view(1, 2, 512, 7, 7)).view(2, 512, 7, 7)
    p_o = [cva_o]
    for m in (oe.fusion_layer, oe.layer1, oe.layer2, oe.layer3):
        p_o.append(m(p_o[-1]))
def nhwc(t, hw, C): return t.view(2, hw, hw, C).permute(0, 3, 1, 2)
for math in ("f32", "bf16"):
    ops.set_math(math)
    with torch.no_grad():
        out, tape = pe._fwd(x.to(dev), save=True)
    B, V, c_stem, mp_idx, c_blocks, res_feat, rr, swin_tape, neck, c_cva, c_post = tape
    for k, (f, red, cc) in enumerate(neck):
        hw = oe.swin_transformer.out_spatial[k]
        print(math, f"neck {k} reduce:", rel(nhwc(red, hw, 256), chains[k][0]))
        for j, c in enumerate(cc):
            z = c[3]; hw2 = hw >> (j + 1)
            print(math, f"neck {k} chain {j}:", rel(nhwc(z, hw2, 256), chains[k][j + 1]))
    cat = c_cva[0]
    print(math, "cat:", rel(nhwc(cat, 7, 512), cat_o))
    print(math, "cva out (input of fusion):", rel(nhwc(c_post[0][0], 7, 512), cva_o))
    for j, c in enumerate(c_post):
        print(math, f"post conv {j}:", rel(nhwc(c[3], 7, 256), p_o[j + 1]))
ops.set_math("f32")
