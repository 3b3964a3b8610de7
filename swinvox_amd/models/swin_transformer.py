"""Swin-T / Swin-B backbone + per-stage LayerNorm([C,H,W]) heads on HIP kernels.

Mirrors reference models/swin_transformer.py:10-94 (class SwinTransformer, attributes `model`, `layer_norm`,
`dropout`, `out_channels`, `out_spatial`) and the timm 1.0.15 `swin_tiny_patch4_window7_224` FeatureListNet it
wraps (state-dict keys `model.patch_embed.{proj,norm}`, `model.layers_{i}.downsample.{norm,reduction}`,
`model.layers_{i}.blocks.{j}.{norm1,attn.{qkv,proj,relative_position_bias_table},norm2,mlp.{fc1,fc2}}`;
notebook cell 68).  No pretrained weights are fetched (there is no network); `pretrained` is accepted and ignored
with a log line, exactly the tensors the reference would overwrite with init_weights anyway.

The modules below only HOLD parameters; the arithmetic is the kernel chain in swin_forward / swin_backward.
"""
from __future__ import annotations

import logging
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_GELU, ACT_NONE, ConvSpec, call, empty, fempty, fzeros, ptr, zeros

_VARIANTS = {
    "tiny": dict(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24)),
    "base": dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32)),
}


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: run the parent Encoder / SwinTransformer module instead")


class WindowAttention(_Holder):
    def __init__(self, dim, heads, ws=7):
        super().__init__()
        self.num_heads = heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class SwinBlock(_Holder):
    def __init__(self, dim, res, heads, shift, drop_path):
        super().__init__()
        self.dim, self.res, self.heads = dim, res, heads
        self.shift = 0 if res <= 7 else shift          # timm: window clipped to the map => no shift (stage 3)
        self.drop_path = float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim)
        self.s_qkv, self.s_proj = ConvSpec.linear(dim, 3 * dim), ConvSpec.linear(dim, dim)
        self.s_fc1, self.s_fc2 = ConvSpec.linear(dim, 4 * dim), ConvSpec.linear(4 * dim, dim)


class PatchMerging(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.spec = ConvSpec.linear(4 * dim, 2 * dim)


class SwinStage(_Holder):
    def __init__(self, dim_in, dim, res, depth, heads, dps, merge):
        super().__init__()
        self.dim, self.res = dim, res
        self.downsample = PatchMerging(dim_in) if merge else nn.Identity()
        self.blocks = nn.Sequential(*[SwinBlock(dim, res, heads, 0 if i % 2 == 0 else 3, dps[i]) for i in range(depth)])


class PatchEmbed(_Holder):
    def __init__(self, in_ch, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_ch, dim, 4, 4)
        self.norm = nn.LayerNorm(dim)


class _FeatureInfo:
    def __init__(self, ch):
        self._ch = list(ch)

    def channels(self):
        return list(self._ch)


class SwinBackbone(_Holder):
    def __init__(self, out_indices, embed_dim, depths, heads, img_size=224, drop_path_rate=0.1, in_ch=3):
        super().__init__()
        self.out_indices = list(out_indices)
        self.patch_embed = PatchEmbed(in_ch, embed_dim)
        dps = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        res, dim_in, ofs, chans = img_size // 4, embed_dim, 0, []
        for i, d in enumerate(depths):
            dim = embed_dim * 2 ** i
            if i > 0:
                res //= 2
            if i <= max(self.out_indices):   # timm's FeatureListNet drops every module after the last requested stage
                setattr(self, f"layers_{i}", SwinStage(dim_in, dim, res, d, heads[i], dps[ofs:ofs + d], i > 0))
            ofs, dim_in = ofs + d, dim
            chans.append(dim)
        self.feature_info = _FeatureInfo([chans[i] for i in self.out_indices])
        self.embed_spec = ConvSpec.conv2d(in_ch, embed_dim, 4, 4, 0)
        self.embed_lin = ConvSpec.linear(in_ch * 16, embed_dim)     # the same layer on patch rows [(ky, kx, c)] (sv_encoder_prep)

    def stages(self) -> List[SwinStage]:
        return [getattr(self, f"layers_{i}") for i in range(max(self.out_indices) + 1)]


class SwinTransformer(_Holder):
    def __init__(self, cfg, in_channels=3, img_size=224, pretrained=True, variant="tiny"):
        super().__init__()
        self.cfg, self.img_size = cfg, img_size
        if pretrained:
            logging.info("swinvox_amd: pretrained Swin weights are not fetched (offline); load a state_dict instead")
        stages = list(cfg.NETWORK.SWIN_T_STAGES)
        self.model = SwinBackbone(stages, img_size=img_size, in_ch=in_channels, **_VARIANTS[variant])
        self.out_channels = [self.model.feature_info.channels()[i] for i in range(len(stages))]
        self.out_spatial = [img_size // (4 * 2 ** i) for i in stages]
        self.layer_norm = nn.ModuleList([
            nn.LayerNorm([self.out_channels[i], self.out_spatial[i], self.out_spatial[i]]) for i in range(len(stages))])
        self.dropout = nn.Dropout(0.05)


# ---------------------------------------------------------------------------------------------------
# kernel chains
# ---------------------------------------------------------------------------------------------------
def _drop_scale(I, p, seed, like):
    sc = fempty(I, like=like)
    call("sv_droppath_scale", ptr(sc), I, float(p), int(seed), ops.seed_epoch_ptr())
    return sc


def block_forward(blk: SwinBlock, x, I, training, stochastic, seeds, save=True):
    """x [I*res*res, C] -> same shape; returns (out, ctx).  save=False (no backward will follow): the fused attention branch
    skips the tensors it would store for the backward."""
    H = W = blk.res
    Cd, M = blk.dim, I * H * W
    dp = blk.drop_path if (training and stochastic) else 0.0
    esz = 2.0 if x.dtype == torch.bfloat16 else 4.0
    if ops.fused_attn_block_enabled(Cd, blk.heads):
        # norm1 -> qkv -> window attention -> proj -> drop-path -> +x in ONE kernel; with save it also stores what the unfused backward reads
        sc1 = _drop_scale(I, dp, seeds(), x) if dp > 0 else None
        sc2 = _drop_scale(I, dp, seeds(), x) if dp > 0 else None
        x1 = empty(M, Cd, like=x)
        ln1 = qkv = att = m1 = r1 = None
        if save:
            ln1, qkv, att = empty(M, Cd, like=x), empty(M, 3 * Cd, like=x), empty(M, Cd, like=x)
            m1, r1 = fempty(M, like=x), fempty(M, like=x)
        a = blk.attn
        # LayerNorm-free products of the branch: qkv + QK^T + PV + proj per token; bytes: x in, x1 out (+ the 5 saved rows when training)
        ops.traced_call("sv_swin_attn_block_fwd", 2.0 * M * Cd * 4 * Cd + 4.0 * 49 * 32 * M * blk.heads, esz * M * Cd * (7 if save else 2),
                        ptr(x), ptr(blk.norm1.weight), ptr(blk.norm1.bias), ptr(a.qkv.weight), ptr(a.qkv.bias), ptr(a.relative_position_bias_table),
                        ptr(a.proj.weight), ptr(a.proj.bias), ptr(sc1), ptr(x1), ptr(ln1), ptr(m1), ptr(r1), ptr(qkv), ptr(att),
                        I, H, W, Cd, blk.heads, blk.shift, float(blk.norm1.eps), tag=f"M={M} C={Cd}")
    else:
        ln1, m1, r1 = ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, M, Cd)
        qkv = empty(M, 3 * Cd, like=x)
        ops.linear_fwd(ln1, M, blk.s_qkv, blk.attn.qkv.weight, qkv, bias=blk.attn.qkv.bias)
        att = empty(M, Cd, like=x)
        # algorithmic work of the core (49-token windows, no padding): QK^T + PV = 4 * 49 * 32 flop per (token, head); bytes: qkv in, out
        ops.traced_call("sv_window_attention_fwd", 4.0 * 49 * 32 * M * blk.heads, esz * 4 * M * Cd, ptr(qkv), ptr(blk.attn.relative_position_bias_table),
                        ptr(att), I, H, W, Cd, blk.heads, blk.shift, ops.attention_math(), tag=f"M={M} C={Cd}")
        sc1 = _drop_scale(I, dp, seeds(), x) if dp > 0 else None
        sc2 = _drop_scale(I, dp, seeds(), x) if dp > 0 else None
        x1 = empty(M, Cd, like=x)
        ops.linear_fwd(att, M, blk.s_proj, blk.attn.proj.weight, x1, bias=blk.attn.proj.bias, residual=x, ldr=Cd, row_scale=sc1,
                       rows_per_scale=H * W)
    if ops.fused_mlp_enabled(Cd):
        # norm2 -> fc1 -> GELU -> fc2 -> drop-path -> +x1 in ONE kernel; the 4C-wide hidden activation never reaches HBM
        packs = torch.empty(16 * Cd * Cd, dtype=torch.bfloat16, device=x.device)
        call("sv_swin_mlp_pack", ptr(blk.mlp.fc1.weight), ptr(blk.mlp.fc2.weight), ptr(packs), Cd)
        x2 = empty(M, Cd, like=x)
        unit = 2.0 * M * Cd * 4 * Cd
        ops.traced_call("sv_swin_mlp_fwd", 2 * unit, 4.0 * M * Cd, ptr(x1), ptr(x2), ptr(blk.norm2.weight), ptr(blk.norm2.bias), ptr(packs),
                        ptr(blk.mlp.fc1.bias), ptr(blk.mlp.fc2.bias), ptr(sc2), H * W, M, Cd, float(blk.norm2.eps), tag=f"M={M} C={Cd}")
        return x2, (x, m1, r1, ln1, qkv, att, sc1, sc2, x1, None, None, None, packs, None, I)
    ln2, m2, r2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, M, Cd)
    hpre = empty(M, 4 * Cd, like=x)
    h = empty(M, 4 * Cd, like=x)
    ops.linear_fwd(ln2, M, blk.s_fc1, blk.mlp.fc1.weight, h, bias=blk.mlp.fc1.bias, act=ACT_GELU, pre_act=hpre)
    x2 = empty(M, Cd, like=x)
    ops.linear_fwd(h, M, blk.s_fc2, blk.mlp.fc2.weight, x2, bias=blk.mlp.fc2.bias, residual=x1, ldr=Cd, row_scale=sc2,
                   rows_per_scale=H * W)
    return x2, (x, m1, r1, ln1, qkv, att, sc1, sc2, x1, m2, r2, ln2, hpre, h, I)


def block_backward(blk: SwinBlock, ctx, dx2, grads):
    """dx2 is consumed (used as the accumulator of the residual path); returns dx."""
    x, m1, r1, ln1, qkv, att, sc1, sc2, x1, m2, r2, ln2, hpre, h, I = ctx
    H = W = blk.res
    Cd, M = blk.dim, I * H * W
    # ---- MLP branch: x2 = x1 + s2 * fc2(gelu(fc1(ln2)))
    if ln2 is None:
        dx1 = _fused_mlp_backward(blk, x1, hpre, sc2, dx2, grads, M, Cd, H * W)
        return _attention_backward(blk, x, m1, r1, ln1, qkv, att, sc1, dx1, grads, I, H, W, Cd, M)
    dbr = dx2
    if sc2 is not None:
        dbr = empty(M, Cd, like=x)
        call("sv_rowscale", ptr(dx2), ptr(sc2), ptr(dbr), M, Cd, H * W)
    # without a drop-path copy dbr IS dx2, which the LayerNorm backward below updates in place -> keep this one in order
    ops.linear_wgrad(dbr, h, M, blk.s_fc2, grads[blk.mlp.fc2.weight], grads[blk.mlp.fc2.bias], async_ok=sc2 is not None)
    dh = empty(M, 4 * Cd, like=x)
    ops.linear_dgrad(dbr, M, blk.s_fc2, blk.s_fc2.pack_dgrad(blk.mlp.fc2.weight), dh, act_grad_src=hpre, act_grad_kind=ACT_GELU)
    ops.linear_wgrad(dh, ln2, M, blk.s_fc1, grads[blk.mlp.fc1.weight], grads[blk.mlp.fc1.bias])
    dln2 = empty(M, Cd, like=x)
    ops.linear_dgrad(dh, M, blk.s_fc1, blk.s_fc1.pack_dgrad(blk.mlp.fc1.weight), dln2)
    ops.layernorm_bwd(dln2, x1, blk.norm2.weight, m2, r2, dx2, grads[blk.norm2.weight], grads[blk.norm2.bias], M, Cd, accumulate_dx=True)
    dx1 = dx2
    return _attention_backward(blk, x, m1, r1, ln1, qkv, att, sc1, dx1, grads, I, H, W, Cd, M)


def _fused_mlp_backward(blk, x1, packs, sc2, dx2, grads, M, Cd, HW):
    """Data gradient (dx1 = dx2 + branch gradient, LayerNorm parameter gradients) and weight gradients of the fused MLP branch:
    both recompute norm2 and the pre-activation from x1; the weight gradients run on the weight-gradient stream."""
    unit = 2.0 * M * Cd * 4 * Cd
    eps = float(blk.norm2.eps)
    dx1 = empty(M, Cd, like=x1)
    ops.traced_call("sv_swin_mlp_bwd", 3 * unit, 6.0 * M * Cd, ptr(x1), ptr(dx2), ptr(dx1), ptr(blk.norm2.weight), ptr(blk.norm2.bias), ptr(packs),
                    ptr(blk.mlp.fc1.bias), ptr(sc2), HW, ptr(grads[blk.norm2.weight]), ptr(grads[blk.norm2.bias]), M, Cd, eps, tag=f"M={M} C={Cd}")
    w1r, w2tr = blk.s_fc1.pack_fwd(blk.mlp.fc1.weight), blk.s_fc2.pack_dgrad(blk.mlp.fc2.weight)     # bf16 [4C][C] rows both

    def launch():
        ops.traced_call("sv_swin_mlp_wgrad", 4 * unit, 4.0 * M * Cd, ptr(x1), ptr(dx2), ptr(blk.norm2.weight), ptr(blk.norm2.bias), ptr(w1r), ptr(w2tr),
                        ptr(blk.mlp.fc1.bias), ptr(sc2), HW, ptr(grads[blk.mlp.fc1.weight]), ptr(grads[blk.mlp.fc1.bias]),
                        ptr(grads[blk.mlp.fc2.weight]), ptr(grads[blk.mlp.fc2.bias]), M, Cd, eps, tag=f"M={M} C={Cd}")

    aw = ops._CTX.awg
    if aw is not None:
        aw.stream.wait_stream(torch.cuda.current_stream())
        aw.held.append((x1, dx2, w1r, w2tr))
        with torch.cuda.stream(aw.stream):
            launch()
    else:
        launch()
    return dx1


def _attention_backward(blk, x, m1, r1, ln1, qkv, att, sc1, dx1, grads, I, H, W, Cd, M):
    """x1 = x + s1 * proj(attn(qkv(ln1))): dx1 is consumed (accumulator of the residual path); returns dx."""
    esz = 2.0 if x.dtype == torch.bfloat16 else 4.0
    if ops.fused_attn_block_bwd_enabled(Cd, blk.heads):
        # data path in ONE kernel: dx1 -> projection data gradient -> attention backward (P recomputed) -> qkv data gradient -> LayerNorm
        # backward + residual; dqkv goes to HBM for the weight gradients, which stay on the engine (weight-gradient stream)
        a = blk.attn
        dqkv, dx = empty(M, 3 * Cd, like=x), empty(M, Cd, like=x)
        dbr = empty(M, Cd, like=x) if sc1 is not None else None
        ws = ops.zeros_f64(8 * 169 * blk.heads, x.device)    # sv_window_attention_bwd_workspace_floats(heads) floats, zero on entry
        ops.traced_call("sv_swin_attn_block_bwd", 2.0 * M * Cd * 4 * Cd + 10.0 * 49 * 32 * M * blk.heads, esz * M * Cd * (10 if sc1 is not None else 9),
                        ptr(dx1), ptr(qkv), ptr(x), ptr(m1), ptr(r1), ptr(blk.norm1.weight), ptr(a.qkv.weight), ptr(a.proj.weight),
                        ptr(a.relative_position_bias_table), ptr(sc1), ptr(dqkv), ptr(dx), ptr(dbr), ptr(grads[blk.norm1.weight]),
                        ptr(grads[blk.norm1.bias]), ptr(grads[a.relative_position_bias_table]), ptr(ws), I, H, W, Cd, blk.heads, blk.shift,
                        tag=f"M={M} C={Cd}")
        ops.linear_wgrad(dbr if dbr is not None else dx1, att, M, blk.s_proj, grads[a.proj.weight], grads[a.proj.bias])
        ops.linear_wgrad(dqkv, ln1, M, blk.s_qkv, grads[a.qkv.weight], grads[a.qkv.bias])
        return dx
    # ---- attention branch: x1 = x + s1 * proj(attn(qkv(ln1)))
    dbr = dx1
    if sc1 is not None:
        dbr = empty(M, Cd, like=x)
        call("sv_rowscale", ptr(dx1), ptr(sc1), ptr(dbr), M, Cd, H * W)
    ops.linear_wgrad(dbr, att, M, blk.s_proj, grads[blk.attn.proj.weight], grads[blk.attn.proj.bias], async_ok=sc1 is not None)
    datt = empty(M, Cd, like=x)
    ops.linear_dgrad(dbr, M, blk.s_proj, blk.s_proj.pack_dgrad(blk.attn.proj.weight), datt)
    dqkv = empty(M, 3 * Cd, like=x)
    ws = ops.zeros_f64(8 * 169 * blk.heads, x.device)    # sv_window_attention_bwd_workspace_floats(heads) floats, zero on entry
    # backward = 2.5 x the forward products (recomputed P, dP, dQ, dK, dV); bytes: qkv + dout in, dqkv out
    ops.traced_call("sv_window_attention_bwd", 10.0 * 49 * 32 * M * blk.heads, esz * 7 * M * Cd, ptr(qkv), ptr(blk.attn.relative_position_bias_table),
                    ptr(datt), ptr(dqkv), ptr(grads[blk.attn.relative_position_bias_table]), ptr(ws), I, H, W, Cd, blk.heads, blk.shift,
                    ops._STATE["math"], tag=f"M={M} C={Cd}")
    ops.linear_wgrad(dqkv, ln1, M, blk.s_qkv, grads[blk.attn.qkv.weight], grads[blk.attn.qkv.bias])
    dln1 = empty(M, Cd, like=x)
    ops.linear_dgrad(dqkv, M, blk.s_qkv, blk.s_qkv.pack_dgrad(blk.attn.qkv.weight), dln1)
    ops.layernorm_bwd(dln1, x, blk.norm1.weight, m1, r1, dx1, grads[blk.norm1.weight], grads[blk.norm1.bias], M, Cd, accumulate_dx=True)
    return dx1


def stage_forward(stage: SwinStage, x, I, training, stochastic, seeds, save=True):
    """One backbone stage (timm SwinTransformerStage: PatchMerging at the START of stages 1-3, then the blocks): x [I*Hin*Hin, Cin]
    -> ([I*res*res, dim], stage tape)."""
    sctx = {"merge": None, "blocks": []}
    if not isinstance(stage.downsample, nn.Identity):
        ds = stage.downsample
        Hin = stage.res * 2
        Mo = I * stage.res * stage.res
        lnm, mm, rm = ops.layernorm_fwd(x, ds.norm.weight, ds.norm.bias, Mo, 2 * stage.dim, merge_hw=(Hin, Hin))
        y = empty(Mo, stage.dim, like=x)
        ops.linear_fwd(lnm, Mo, ds.spec, ds.reduction.weight, y)
        sctx["merge"] = (x, lnm, mm, rm, Mo, Hin)
        x = y
    for blk in stage.blocks:
        x, bctx = block_forward(blk, x, I, training, stochastic, seeds, save)
        sctx["blocks"].append(bctx)
    return x, sctx


def stage_backward(stage: SwinStage, sctx, dx, grads, I):
    """dx [I*res*res, dim] (consumed) -> gradient wrt the stage input [I*Hin*Hin, Cin]; parameter gradients accumulate into `grads`."""
    for blk, bctx in zip(reversed(list(stage.blocks)), reversed(sctx["blocks"])):
        dx = block_backward(blk, bctx, dx, grads)
    if sctx["merge"] is not None:
        ds = stage.downsample
        xin, lnm, mm, rm, Mo, Hin = sctx["merge"]
        ops.linear_wgrad(dx, lnm, Mo, ds.spec, grads[ds.reduction.weight], None)
        dln = empty(Mo, 2 * stage.dim, like=dx)
        ops.linear_dgrad(dx, Mo, ds.spec, ds.spec.pack_dgrad(ds.reduction.weight), dln)
        dxin = empty(I * Hin * Hin, stage.dim // 2, like=dx)
        ops.layernorm_bwd(dln, xin, ds.norm.weight, mm, rm, dxin, grads[ds.norm.weight], grads[ds.norm.bias], Mo, 2 * stage.dim,
                          merge_hw=(Hin, Hin))
        dx = dxin
    return dx


def swin_forward(st: SwinTransformer, img_nhwc, I, training, stochastic, seeds, ready=None, save=True, patches=None):
    """img_nhwc [I,224,224,3] - or `patches` [I*56*56, 48], the 4 x 4 x 3 pixel blocks as rows [(ky, kx, c)] (sv_encoder_prep) - -> list of
    stage-head outputs [I*HW, C] (NHWC rows) + tape.  `ready` (optional list) receives one event per head output, recorded on the current
    stream as soon as that output is complete, so that another stream can consume the early stages while the later ones are still being
    computed."""
    bb = st.model
    pe = bb.patch_embed
    S = st.img_size
    Ce = pe.proj.out_channels
    M = I * (S // 4) ** 2
    src = patches if patches is not None else img_nhwc
    emb = empty(M, Ce, like=src)
    if patches is not None:
        # PatchEmbed's Conv2d(3, C, 4, 4) as a Linear(48, C) on the patch rows: weight [co][c][(ky, kx)] -> [co][(ky, kx)][c].  On the image the
        # layer gathers 3-channel (6-byte) taps on the engine's scalar path: 0.31 ms forward + 0.28 ms weight gradient at 512 images
        w48 = torch.empty(Ce, 48, dtype=torch.float32, device=src.device)
        ops.transpose(pe.proj.weight, w48, Ce, 3, 16)
        bb.embed_lin.forward(patches, M, (1, 1, 1), ops.pack_one(bb.embed_lin, w48, "f"), emb, bias=pe.proj.bias)
    else:
        wpe = bb.embed_spec.pack_fwd(pe.proj.weight)
        bb.embed_spec.forward(img_nhwc, I, (1, S, S), wpe, emb, bias=pe.proj.bias)
    x, pm, pr = ops.layernorm_fwd(emb, pe.norm.weight, pe.norm.bias, M, Ce)
    tape = {"embed": (src, emb, pm, pr, patches is not None), "stages": [], "heads": []}
    feats = []
    head_i = 0
    for si, stage in enumerate(bb.stages()):
        x, sctx = stage_forward(stage, x, I, training, stochastic, seeds, save)
        tape["stages"].append(sctx)
        if si in bb.out_indices:
            ln = st.layer_norm[head_i]
            Cs, Hs = stage.dim, stage.res
            L = Cs * Hs * Hs
            wt, bt = fempty(L, like=x), fempty(L, like=x)
            ops.transpose(ln.weight, wt, 1, Cs, Hs * Hs)       # [C,HW] -> [HW,C]
            ops.transpose(ln.bias, bt, 1, Cs, Hs * Hs)
            y = empty(I * Hs * Hs, Cs, like=x)
            mr = fempty(2 * I, like=x)
            ws = fempty(int(hipws(I, L)), like=x)
            p = st.dropout.p if (training and stochastic) else 0.0
            seed = seeds() if p > 0 else 0
            call("sv_ln_image_fwd", ptr(x), ptr(wt), ptr(bt), ptr(y), ptr(mr), ptr(ws), I, L, float(ln.eps), float(p), seed, ops.seed_epoch_ptr())
            tape["heads"].append((si, head_i, x, wt, mr, p, seed, L))
            feats.append(y)
            if ready is not None:
                ev = torch.cuda.Event()
                ev.record()
                ready.append(ev)
            head_i += 1
    return feats, tape


def hipws(I, L):
    from .. import hip
    return hip.load().sv_ln_image_workspace_floats(I, L)


def swin_backward(st: SwinTransformer, tape, dfeats, I, grads, ready=None):
    """dfeats: list of gradients wrt the stage-head outputs ([I*HW, C]).  No gradient wrt the image is produced.
    `ready` (optional): one event per entry of dfeats, awaited right before that gradient is first read (the producer may
    still be working on the earlier stages' gradients on another stream)."""
    bb = st.model
    stages = bb.stages()
    head_of = {si: (hi, xs, wt, mr, p, seed, L) for (si, hi, xs, wt, mr, p, seed, L) in tape["heads"]}
    dx = None
    for si in reversed(range(len(stages))):
        stage = stages[si]
        if si in head_of:
            hi, xs, wt, mr, p, seed, L = head_of[si]
            if ready is not None and ready[hi] is not None:
                torch.cuda.current_stream().wait_event(ready[hi])
            ln = st.layer_norm[hi]
            Cs, Hs = stage.dim, stage.res
            dxe = empty(I * Hs * Hs, Cs, like=xs)
            dwt, dbt = fzeros(L, like=xs), fzeros(L, like=xs)
            sums = torch.empty(2 * I, dtype=torch.float64, device=xs.device)
            call("sv_ln_image_bwd", ptr(dfeats[hi]), ptr(xs), ptr(wt), ptr(mr), ptr(dxe), ptr(dwt), ptr(dbt), ptr(sums), I, L, float(p), seed, ops.seed_epoch_ptr())
            ops.transpose(dwt, grads[ln.weight], 1, Hs * Hs, Cs)   # [HW,C] -> [C,HW]
            ops.transpose(dbt, grads[ln.bias], 1, Hs * Hs, Cs)
            if dx is None:
                dx = dxe
            else:
                call("sv_axpby", ptr(dx), ptr(dxe), ptr(dx), 1.0, 1.0, dx.numel())
        dx = stage_backward(stage, tape["stages"][si], dx, grads, I)
    # patch embed: LN backward, then conv weight/bias gradient (the image itself needs no gradient)
    img, emb, pm, pr, as_patches = tape["embed"]
    pe = bb.patch_embed
    M, Ce = emb.shape
    demb = empty(M, Ce, like=dx)
    ops.layernorm_bwd(dx, emb, pe.norm.weight, pm, pr, demb, grads[pe.norm.weight], grads[pe.norm.bias], M, Ce)
    S = st.img_size
    if as_patches:
        dw48 = torch.zeros(Ce, 48, dtype=torch.float32, device=dx.device)
        bb.embed_lin.wgrad(demb, img, M, (1, 1, 1), dw48, db=grads[pe.proj.bias])
        aw = ops._CTX.awg          # the weight gradient went out on the weight-gradient stream: its way back to [co][c][(ky, kx)] follows it there
        if aw is not None:
            aw.held.append((dw48,))
            with torch.cuda.stream(aw.stream):
                ops.transpose(dw48, grads[pe.proj.weight], Ce, 16, 3)
        else:
            ops.transpose(dw48, grads[pe.proj.weight], Ce, 16, 3)
    else:
        bb.embed_spec.wgrad(demb, img, I, (1, S, S), grads[pe.proj.weight], db=grads[pe.proj.bias])
