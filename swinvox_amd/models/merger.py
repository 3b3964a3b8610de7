"""Merger on HIP kernels; mirrors reference models/merger.py:9-107.

forward(raw_features [B,V,9,32,32,32], coarse_volumes [B,V,32,32,32]) -> [B,32,32,32].
The six 3x3x3 convolutions work on channels-last data padded 9 -> 12 channels (16-byte channel vectors); layers 1-4
write straight into the 4x12-wide concat buffer that layer 5 reads.  Two kernel back-ends:
  * set_math("bf16"): the LDS-halo MFMA stencils of csrc/stencil.hip (brick + halo staged once in LDS);
  * set_math("f32") : the generic implicit-GEMM engine in exact fp32 (parity runs).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_LRELU, BatchNormState, ConvSpec, call, empty, fzeros, ptr, zeros
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

    # ---- per-layer contraction dispatch ---------------------------------------------------------------
    def _layer(self, li):
        return (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5, self.layer6)[li]

    def _spec(self, li):
        return self._s14 if li < 4 else (self._s5 if li == 4 else self._s6)

    def _w5_padded(self):
        w5 = self.layer5[0].weight                       # [9,36,3,3,3] -> [9,48,27] with the concat column map
        wp = fzeros(9, 48, 27, like=w5)
        wp[:, self._cat_cols] = w5.detach().reshape(9, 36, 27)
        return wp

    def _stencil_pack(self, li, dgrad):
        """bf16 weights for csrc/stencil.hip: forward [16][27][16G] (memory input channels), data-gradient
        [16*NT][27][16] (rows = memory channels of the conv input, taps flipped).  Tiny tensors: a few torch index ops."""
        w = self._layer(li)[0].weight
        wp = torch.empty(16 * 27 * (48 if li == 4 else 16), dtype=torch.bfloat16, device=w.device)
        call("sv_merger_pack", ptr(w), ptr(wp), w.shape[0], w.shape[1], 1 if dgrad else 0, 1 if li == 4 else 0)
        return wp

    def _conv_fwd(self, li, x, ldi, y, ldc, stats, w5p=None):
        conv, I = self._layer(li)[0], y.shape[0] // VOX
        if ops.get_math() == "bf16":   # layer 5 reads the four dense 12-wide planes of the concat
            call("sv_stencil3_fwd", ptr(x), ldi, 48 if li == 4 else 12, 3 if li == 4 else 1, ptr(self._stencil_pack(li, False)), 1,
                 ptr(conv.bias), ptr(y), ldc, 0, conv.out_channels, None, 0, ptr(stats), I, 32, 32, 32, I * VOX * 12 if li == 4 else 0, 0)
        else:
            sp = self._spec(li)
            sp.forward(x, I, G, sp.pack_fwd(w5p if li == 4 else conv.weight), y, ldi=ldi, ldc=ldc, bias=conv.bias, stats=stats)

    def _conv_dgrad(self, li, dy, lddy, dx, lddx, accumulate, w5p=None):
        conv, I = self._layer(li)[0], dy.shape[0] // VOX
        if ops.get_math() == "bf16":
            call("sv_stencil3_fwd", ptr(dy), lddy, lddy if lddy <= 12 else 12, 1, ptr(self._stencil_pack(li, True)), 3 if li == 4 else 1,
                 None, ptr(dx), lddx, 0, 48 if li == 4 else 9, ptr(dx) if accumulate else None, lddx, None, I, 32, 32, 32,
                 0, I * VOX * 12 if li == 4 else 0)
        else:
            sp = self._spec(li)
            epi = dict(residual=dx, ldr=lddx) if accumulate else {}
            sp.dgrad(dy, I, G, sp.pack_dgrad(w5p if li == 4 else conv.weight), dx, lddy=lddy, lddx=lddx, **epi)

    def _conv_wgrad(self, li, dy, lddy, x, ldx, grads):
        conv, I = self._layer(li)[0], dy.shape[0] // VOX
        if ops.get_math() == "bf16":
            ws = ops.zeros_f64(8 * conv.out_channels * conv.in_channels * 27, dy.device)   # 16 slot images of floats, zero on entry
            call("sv_stencil3_wgrad", ptr(x), ldx, 48 if li == 4 else 12, 3 if li == 4 else 1, ptr(dy), lddy, lddy if lddy <= 12 else 12,
                 ptr(grads[conv.weight]), ptr(grads[conv.bias]), ptr(ws), conv.out_channels, conv.in_channels, 12 if li == 4 else 16, 9, I, 32, 32, 32,
                 I * VOX * 12 if li == 4 else 0)
        elif li == 4:
            dw5p = fzeros(9, 48, 27, like=dy)
            self._s5.wgrad(dy, x, I, G, dw5p, lddy=lddy, ldx=ldx, async_ok=False)   # dw5p is read back right below
            grads[conv.weight].view(9, 36, 27).copy_(dw5p[:, self._cat_cols])
        else:
            self._spec(li).wgrad(dy, x, I, G, grads[conv.weight], lddy=lddy, ldx=ldx)

    def _bias_grad(self, dy, M, C, ld, db):
        """conv bias gradient = column sums of dy: folded into the stencil weight-gradient kernel in bf16 math, a separate
        column-sum kernel next to the generic engine otherwise."""
        if ops.get_math() != "bf16":
            ops.colsum(dy, M, C, ld, db)

    # ---- forward / backward chains -------------------------------------------------------------------------
    def _fwd(self, raw, vol, save):
        B, V = raw.shape[:2]
        M, tr, sl = B * V * VOX, self.training, self._slope
        raw_dtype, vol_dtype = raw.dtype, vol.dtype
        x12 = ops.to_store(as_channels_last12(raw))
        vol = ops.to_store(vol)
        # the reference's torch.cat of the four 9-channel maps (merger.py:84) is never materialised: layers 1-4 write their
        # activations where layer 5 reads them.  Stencil back-end: four DENSE planes [4][M][12] (every per-layer pass streams
        # whole cache lines); generic fp32 engine: 12-column windows of 48-wide rows (one 48-channel input).
        planar = ops.get_math() == "bf16"
        # stencil back-end: every 12-wide row is written whole by its producer (9 channels + 3 zero pads: sv_stencil3_fwd and the
        # narrow BatchNorm kernels store complete 4-channel groups), so these buffers need no zero fill; the generic engine of the
        # fp32 mode writes the 9 real columns only
        buf = empty if planar else zeros
        cat = empty(4, M, 12, like=vol) if planar else zeros(M, 48, like=vol)
        ldz = 12 if planar else 48
        w5p = self._w5_padded() if not planar else None
        ctx14, xin, ldi = [], x12, 12
        for k in range(4):
            bn = self._layer(k)[1]
            y = empty(M, 12, like=vol)              # 9 channels + 3 pad columns: whole-row vector stores / loads
            st = BatchNormState(bn, M, tr)
            self._conv_fwd(k, xin, ldi, y, 12, st.sums)
            st.finalize()
            z = cat[k] if planar else cat[:, 12 * k:]
            st.apply(y, 12, z, ldz, ACT_LRELU, sl)
            ctx14.append((xin, ldi, y, z, st))
            xin, ldi = z, ldz
        y5 = empty(M, 12, like=vol)
        st5 = BatchNormState(self.layer5[1], M, tr)
        self._conv_fwd(4, cat, ldz, y5, 12, st5.sums, w5p)
        st5.finalize()
        z5 = buf(M, 12, like=vol)
        st5.apply(y5, 12, z5, 12, ACT_LRELU, sl)
        y6 = empty(M, 1, like=vol)
        st6 = BatchNormState(self.layer6[1], M, tr)
        self._conv_fwd(5, z5, 12, y6, 1, st6.sums)
        st6.finalize()
        wl = empty(M, 1, like=vol)
        st6.apply(y6, 1, wl, 1, ACT_LRELU, sl)
        out = empty(B, 32, 32, 32, like=vol)
        call("sv_merge_views_fwd", ptr(wl), ptr(vol), ptr(out), B, V, VOX)
        tape = (B, V, x12, vol, cat, ctx14, w5p, y5, st5, z5, y6, st6, wl, out, raw_dtype, vol_dtype) if save else None
        return ops.to_f32(out), tape

    def _bwd(self, tape, grads, in_needs, dout):
        B, V, x12, vol, cat, ctx14, w5p, y5, st5, z5, y6, st6, wl, out, raw_dtype, vol_dtype = tape
        M, sl = B * V * VOX, self._slope
        dout = ops.to_store(dout)
        dwl = empty(M, 1, like=vol)
        dvol = empty(B, V, 32, 32, 32, like=vol)
        call("sv_merge_views_bwd", ptr(wl), ptr(vol), ptr(out), ptr(dout), ptr(dwl), ptr(dvol), B, V, VOX)
        # ---- layer 6
        conv6, bn6 = self.layer6[0], self.layer6[1]
        dy6 = zeros(M, 4, like=vol)
        # z = None everywhere below: the LeakyReLU mask is recomputed from the conv output (scale * y + shift > 0), one tensor read
        # less in each of the two BatchNorm backward passes
        st6.backward(dwl, 1, None, 0, y6, 1, dy6, 4, grads[bn6.weight], grads[bn6.bias], ACT_LRELU, sl)
        self._bias_grad(dy6, M, 1, 4, grads[conv6.bias])
        self._conv_wgrad(5, dy6, 4, z5, 12, grads)
        buf = empty if ops.get_math() == "bf16" else zeros   # see _fwd: whole rows are written by the stencil / narrow kernels
        dz5 = buf(M, 12, like=vol)
        self._conv_dgrad(5, dy6, 4, dz5, 12, False)
        # ---- layer 5
        conv5, bn5 = self.layer5[0], self.layer5[1]
        dy5 = buf(M, 12, like=vol)
        st5.backward(dz5, 12, None, 0, y5, 12, dy5, 12, grads[bn5.weight], grads[bn5.bias], ACT_LRELU, sl)
        self._bias_grad(dy5, M, 9, 12, grads[conv5.bias])
        planar = cat.dim() == 3
        ldz = 12 if planar else 48
        self._conv_wgrad(4, dy5, 12, cat, ldz, grads)
        # data-gradient wrt the concat buffer, same storage scheme as `cat` (pad columns receive zero weights)
        dcat = empty(4, M, 12, like=vol) if planar else zeros(M, 48, like=vol)
        self._conv_dgrad(4, dy5, 12, dcat, ldz, False, w5p)
        # ---- layers 4..1: z_k feeds layer k+1 and the concat -> gradients add up in dcat[:, 12k:12k+9]
        dx = None
        for k in (3, 2, 1, 0):
            conv, bn = self._layer(k)[0], self._layer(k)[1]
            xin, ldi, y, z, st = ctx14[k]
            dzk = dcat[k] if planar else dcat[:, 12 * k:]
            dy = buf(M, 12, like=vol)
            st.backward(dzk, ldz, None, 0, y, 12, dy, 12, grads[bn.weight], grads[bn.bias], ACT_LRELU, sl)
            self._bias_grad(dy, M, 9, 12, grads[conv.bias])
            self._conv_wgrad(k, dy, 12, xin, ldi, grads)
            if k > 0:   # accumulate into the previous layer's slot of dcat
                self._conv_dgrad(k, dy, 12, dcat[k - 1] if planar else dcat[:, 12 * (k - 1):], ldz, True)
            else:
                dx = buf(M, 12, like=vol)
                self._conv_dgrad(k, dy, 12, dx, 12, False)
        # gradients in the dtypes the inputs came in (raw_features arrives in the storage dtype from Decoder, fp32 from anyone else)
        draw = raw_view(dx if raw_dtype == dx.dtype else ops.to_f32(dx), B, V) if in_needs[0] else None
        return (draw, (dvol if vol_dtype == dvol.dtype else ops.to_f32(dvol)) if in_needs[1] else None)
