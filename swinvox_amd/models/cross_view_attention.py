"""Cross-view attention on HIP kernels; mirrors reference models/cross_view_attention.py:10-134.

Parameter holders keep the reference attribute names (`downsample_qkv`, `qkv_conv`, `proj_conv`, `ffn.{0,2}`,
`batch_norm`, `dropout`).  Data layout inside: rows = (sample, view, y, x), channels last.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_GELU, ACT_NONE, BatchNormState, ConvSpec, call, empty, ptr


class CrossViewAttention(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        n = cfg.NETWORK
        self.cfg, self.in_channels = cfg, in_channels
        self.num_heads = n.CROSS_ATT_NUM_HEADS
        self.reduced_channels = in_channels // n.CROSS_ATT_REDUCTION_RATIO
        self.attention_spatial_downsample_ratio = n.ATT_SPATIAL_DOWNSAMPLE_RATIO
        assert self.reduced_channels % self.num_heads == 0, \
            f"reduced_channels ({self.reduced_channels}) must be divisible by num_heads ({self.num_heads})"
        self.head_dim = self.reduced_channels // self.num_heads
        r = self.attention_spatial_downsample_ratio
        self.downsample_qkv = nn.Conv2d(in_channels, in_channels, kernel_size=r, stride=r, groups=in_channels) if r > 1 else None
        self.qkv_conv = nn.Conv2d(in_channels, 3 * self.reduced_channels, kernel_size=1)
        self.softmax = nn.Softmax(dim=-1)
        self.proj_conv = nn.Conv2d(self.reduced_channels, in_channels, kernel_size=1)
        self.ffn = nn.Sequential(nn.Conv2d(in_channels, in_channels, 1), nn.GELU(), nn.Conv2d(in_channels, in_channels, 1))
        self.batch_norm = nn.BatchNorm2d(in_channels)
        self.dropout = nn.Dropout(0.1)
        C, R = in_channels, self.reduced_channels
        self._s_qkv, self._s_proj = ConvSpec.linear(C, 3 * R), ConvSpec.linear(R, C)
        self._s_f0, self._s_f2 = ConvSpec.linear(C, C), ConvSpec.linear(C, C)

    def forward(self, x):  # pragma: no cover - the Encoder drives the kernel chain
        raise RuntimeError("CrossViewAttention runs inside swinvox_amd.models.Encoder (kernel chain cva_forward)")

    # ------------------------------------------------------------------------------------------------
    def cva_forward(self, x, B, V, training, stochastic, seeds):
        """x [B*V*49, C] (7x7 maps, channels last) -> same shape."""
        r = self.attention_spatial_downsample_ratio
        if r not in (1, 2):
            raise NotImplementedError("swinvox_amd: ATT_SPATIAL_DOWNSAMPLE_RATIO must be 1 (attention on the 7x7 grid) or 2 (7x7 -> 3x3, "
                                      "the reference default); got %r" % (r,))
        C, R, I = self.in_channels, self.reduced_channels, B * V
        P = 49 if r == 1 else 9                      # positions of the grid the attention runs on
        if r == 1:                                   # reference cross_view_attention.py:67-73: no depth-wise down-sampling
            dw = x
        else:
            dw = empty(I * 9, C, like=x)
            call("sv_dwconv2x2_fwd", ptr(x), ptr(self.downsample_qkv.weight), ptr(self.downsample_qkv.bias), ptr(dw), I, C)
        qkv = empty(I * P, 3 * R, like=x)
        ops.linear_fwd(dw, I * P, self._s_qkv, self.qkv_conv.weight, qkv, bias=self.qkv_conv.bias)
        att = empty(I * P, R, like=x)
        call("sv_cross_view_attention_fwd", ptr(qkv), ptr(att), B, V, P, R, self.num_heads)
        up = empty(I * 49, C, like=x)
        if r == 1:                                   # :110-120 without the interpolation: proj + x in the projection's epilogue
            ops.linear_fwd(att, I * 49, self._s_proj, self.proj_conv.weight, up, bias=self.proj_conv.bias, residual=x, ldr=C)
        else:
            pr = empty(I * 9, C, like=x)
            ops.linear_fwd(att, I * 9, self._s_proj, self.proj_conv.weight, pr, bias=self.proj_conv.bias)
            call("sv_upsample3to7_add_fwd", ptr(pr), ptr(x), C, ptr(up), I, C)
        f1pre, f1 = empty(I * 49, C, like=x), empty(I * 49, C, like=x)
        ops.linear_fwd(up, I * 49, self._s_f0, self.ffn[0].weight, f1, bias=self.ffn[0].bias, act=ACT_GELU, pre_act=f1pre)
        st = BatchNormState(self.batch_norm, I * 49, training)
        f2 = empty(I * 49, C, like=x)
        ops.linear_fwd(f1, I * 49, self._s_f2, self.ffn[2].weight, f2, bias=self.ffn[2].bias, stats=st.sums)
        st.finalize()
        z = empty(I * 49, C, like=x)
        st.apply(f2, C, z, C, ACT_NONE)
        p = self.dropout.p if (training and stochastic) else 0.0
        seed = 0
        out = z
        if p > 0:
            seed = seeds()
            out = empty(I * 49, C, like=x)
            call("sv_dropout", ptr(z), ptr(out), z.numel(), float(p), seed, ops.seed_epoch_ptr())
        return out, (x, dw, qkv, att, up, f1pre, f1, f2, st, p, seed, B, V)

    def cva_backward(self, ctx, dout, grads):
        x, dw, qkv, att, up, f1pre, f1, f2, st, p, seed, B, V = ctx
        C, R, I = self.in_channels, self.reduced_channels, B * V
        dz = dout
        if p > 0:
            dz = empty(I * 49, C, like=x)
            call("sv_dropout", ptr(dout), ptr(dz), dz.numel(), float(p), seed, ops.seed_epoch_ptr())
        df2 = empty(I * 49, C, like=x)
        st.backward(dz, C, None, 0, f2, C, df2, C, grads[self.batch_norm.weight], grads[self.batch_norm.bias], ACT_NONE)
        ops.linear_wgrad(df2, f1, I * 49, self._s_f2, grads[self.ffn[2].weight], grads[self.ffn[2].bias])
        df1 = empty(I * 49, C, like=x)
        ops.linear_dgrad(df2, I * 49, self._s_f2, self._s_f2.pack_dgrad(self.ffn[2].weight), df1, act_grad_src=f1pre, act_grad_kind=ACT_GELU)
        ops.linear_wgrad(df1, up, I * 49, self._s_f0, grads[self.ffn[0].weight], grads[self.ffn[0].bias])
        dup = empty(I * 49, C, like=x)
        ops.linear_dgrad(df1, I * 49, self._s_f0, self._s_f0.pack_dgrad(self.ffn[0].weight), dup)
        full = self.downsample_qkv is None             # ATT_SPATIAL_DOWNSAMPLE_RATIO = 1: attention on the 7x7 grid itself
        P = 49 if full else 9
        if full:
            dpr = dup
        else:
            dpr = empty(I * 9, C, like=x)
            call("sv_upsample3to7_bwd", ptr(dup), ptr(dpr), I, C)
        ops.linear_wgrad(dpr, att, I * P, self._s_proj, grads[self.proj_conv.weight], grads[self.proj_conv.bias])
        datt = empty(I * P, R, like=x)
        ops.linear_dgrad(dpr, I * P, self._s_proj, self._s_proj.pack_dgrad(self.proj_conv.weight), datt)
        dqkv = empty(I * P, 3 * R, like=x)
        call("sv_cross_view_attention_bwd", ptr(qkv), ptr(datt), ptr(dqkv), B, V, P, R, self.num_heads)
        ops.linear_wgrad(dqkv, dw, I * P, self._s_qkv, grads[self.qkv_conv.weight], grads[self.qkv_conv.bias])
        if full:                                       # dx = qkv data gradient + the residual path, added in the epilogue
            dx = empty(I * 49, C, like=x)
            ops.linear_dgrad(dqkv, I * 49, self._s_qkv, self._s_qkv.pack_dgrad(self.qkv_conv.weight), dx, residual=dup, ldr=C)
            return dx
        ddw = empty(I * 9, C, like=x)
        ops.linear_dgrad(dqkv, I * 9, self._s_qkv, self._s_qkv.pack_dgrad(self.qkv_conv.weight), ddw)
        dx = empty(I * 49, C, like=x)
        call("sv_dwconv2x2_bwd", ptr(ddw), ptr(x), ptr(self.downsample_qkv.weight), ptr(dx), ptr(grads[self.downsample_qkv.weight]),
             ptr(grads[self.downsample_qkv.bias]), I, C)
        call("sv_axpby", ptr(dx), ptr(dup), ptr(dx), 1.0, 1.0, dx.numel())   # + residual path
        return dx
