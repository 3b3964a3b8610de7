"""Merger on HIP kernels; mirrors reference models/merger.py:9-107.

forward(raw_features [B,V,9,32,32,32], coarse_volumes [B,V,32,32,32]) -> [B,32,32,32].
The six 3x3x3 convolutions run on the implicit-GEMM engine over channels-last data padded 9 -> 12 channels
(16-byte channel vectors); layers 1-4 write straight into the 4x12-wide concat buffer that layer 5 reads.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_LRELU, BatchNormState, ConvSpec, call, empty, ptr, zeros
from ._base import HipModule
from .decoder import VOX, as_channels_last12, raw_view

G = (32, 32, 32)


class Merger(HipModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        lk = cfg.NETWORK.LEAKY_VALUE
        self._slope = float(lk)

        def c3(cin, cout):
            return nn.Sequential(nn.Conv3d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm3d(cout), nn.LeakyReLU(lk))

        self.layer1, self.layer2, self.layer3, self.layer4 = c3(9, 9), c3(9, 9), c3(9, 9), c3(9, 9)
        self.layer5 = c3(36, 9)
        self.layer6 = c3(9, 1)
        self._s14 = ConvSpec.conv3d(9, 9, 3, 1, 1, cin_mem=12, cout_mem=12)
        self._s5 = ConvSpec.conv3d(48, 9, 3, 1, 1, cin_mem=48, cout_mem=12)   # 36 real inputs spread over 4 x 12 columns
        self._s6 = ConvSpec.conv3d(9, 1, 3, 1, 1, cin_mem=12, cout_mem=4)
        idx = torch.tensor([12 * g + j for g in range(4) for j in range(9)], dtype=torch.long)
        self.register_buffer("_cat_cols", idx, persistent=False)

    def forward(self, raw_features, coarse_volumes):
        assert raw_features.dim() == 6 and tuple(raw_features.shape[2:]) == (9, 32, 32, 32), "raw_features must be [B,V,9,32,32,32]"
        assert tuple(coarse_volumes.shape) == (raw_features.shape[0], raw_features.shape[1], 32, 32, 32)
        return self._run(raw_features, coarse_volumes)

    def _w5_padded(self):
        w5 = self.layer5[0].weight                       # [9,36,3,3,3] -> [9,48,27] with the concat column map
        wp = zeros(9, 48, 27, like=w5)
        wp[:, self._cat_cols] = w5.detach().reshape(9, 36, 27)
        return wp

    def _fwd(self, raw, vol, save):
        B, V = raw.shape[:2]
        I, M, tr, sl = B * V, B * V * VOX, self.training, self._slope
        x12 = as_channels_last12(raw)
        vol = vol.contiguous()
        cat = zeros(M, 48, like=vol)
        ctx14, xin, ldi = [], x12, 12
        for k, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            conv, bn = layer[0], layer[1]
            y = empty(M, 9, like=vol)
            st = BatchNormState(bn, M, tr)
            self._s14.forward(xin, I, G, self._s14.pack_fwd(conv.weight), y, ldi=ldi, ldc=9, bias=conv.bias, stats=st.sums)
            st.finalize()
            z = cat[:, 12 * k:]
            st.apply(y, 9, z, 48, ACT_LRELU, sl)
            ctx14.append((xin, ldi, y, z, st))
            xin, ldi = z, 48
        conv5, bn5 = self.layer5[0], self.layer5[1]
        w5p = self._w5_padded()
        y5 = empty(M, 9, like=vol)
        st5 = BatchNormState(bn5, M, tr)
        self._s5.forward(cat, I, G, self._s5.pack_fwd(w5p), y5, ldi=48, ldc=9, bias=conv5.bias, stats=st5.sums)
        st5.finalize()
        z5 = zeros(M, 12, like=vol)
        st5.apply(y5, 9, z5, 12, ACT_LRELU, sl)
        conv6, bn6 = self.layer6[0], self.layer6[1]
        y6 = empty(M, 1, like=vol)
        st6 = BatchNormState(bn6, M, tr)
        self._s6.forward(z5, I, G, self._s6.pack_fwd(conv6.weight), y6, ldi=12, ldc=1, bias=conv6.bias, stats=st6.sums)
        st6.finalize()
        wl = empty(M, 1, like=vol)
        st6.apply(y6, 1, wl, 1, ACT_LRELU, sl)
        out = empty(B, 32, 32, 32, like=vol)
        call("sv_merge_views_fwd", ptr(wl), ptr(vol), ptr(out), B, V, VOX)
        tape = (B, V, x12, vol, cat, ctx14, w5p, y5, st5, z5, y6, st6, wl, out) if save else None
        return out, tape

    def _bwd(self, tape, grads, in_needs, dout):
        B, V, x12, vol, cat, ctx14, w5p, y5, st5, z5, y6, st6, wl, out = tape
        I, M, sl = B * V, B * V * VOX, self._slope
        dout = dout.contiguous()
        dwl = empty(M, 1, like=vol)
        dvol = empty(B, V, 32, 32, 32, like=vol)
        call("sv_merge_views_bwd", ptr(wl), ptr(vol), ptr(out), ptr(dout), ptr(dwl), ptr(dvol), B, V, VOX)
        # ---- layer 6
        conv6, bn6 = self.layer6[0], self.layer6[1]
        dy6 = zeros(M, 4, like=vol)
        st6.backward(dwl, 1, wl, 1, y6, 1, dy6, 4, grads[bn6.weight], grads[bn6.bias], ACT_LRELU, sl)
        ops.colsum(dy6, M, 1, 4, grads[conv6.bias])
        self._s6.wgrad(dy6, z5, I, G, grads[conv6.weight], lddy=4, ldx=12)
        dz5 = zeros(M, 12, like=vol)
        self._s6.dgrad(dy6, I, G, self._s6.pack_dgrad(conv6.weight), dz5, lddy=4, lddx=12)
        # ---- layer 5
        conv5, bn5 = self.layer5[0], self.layer5[1]
        dy5 = zeros(M, 12, like=vol)
        st5.backward(dz5, 12, z5, 12, y5, 9, dy5, 12, grads[bn5.weight], grads[bn5.bias], ACT_LRELU, sl)
        ops.colsum(dy5, M, 9, 12, grads[conv5.bias])
        dw5p = zeros(9, 48, 27, like=vol)
        self._s5.wgrad(dy5, cat, I, G, dw5p, lddy=12, ldx=48)
        grads[conv5.weight].view(9, 36, 27).copy_(dw5p[:, self._cat_cols])
        dcat = zeros(M, 48, like=vol)
        # data-gradient wrt the 48-wide concat buffer: produce all 48 columns (pad columns get zero weights)
        s5d = ConvSpec.conv3d(48, 9, 3, 1, 1, cin_mem=48, cout_mem=12)
        s5d.dgrad(dy5, I, G, s5d.pack_dgrad(w5p), dcat, lddy=12, lddx=48)
        # ---- layers 4..1: z_k feeds layer k+1 and the concat -> gradients add up in dcat[:, 12k:12k+9]
        dx = None
        for k in (3, 2, 1, 0):
            layer = (self.layer1, self.layer2, self.layer3, self.layer4)[k]
            conv, bn = layer[0], layer[1]
            xin, ldi, y, z, st = ctx14[k]
            dzk = dcat[:, 12 * k:]
            dy = zeros(M, 12, like=vol)
            st.backward(dzk, 48, z, 48, y, 9, dy, 12, grads[bn.weight], grads[bn.bias], ACT_LRELU, sl)
            ops.colsum(dy, M, 9, 12, grads[conv.bias])
            self._s14.wgrad(dy, xin, I, G, grads[conv.weight], lddy=12, ldx=ldi)
            wd = self._s14.pack_dgrad(conv.weight)
            if k > 0:   # accumulate into the previous layer's slot of dcat
                prev = dcat[:, 12 * (k - 1):]
                self._s14.dgrad(dy, I, G, wd, prev, lddy=12, lddx=48, residual=prev, ldr=48)
            else:
                dx = zeros(M, 12, like=vol)
                self._s14.dgrad(dy, I, G, wd, dx, lddy=12, lddx=12)
        draw = raw_view(dx, B, V) if in_needs[0] else None
        return (draw, dvol if in_needs[1] else None)
