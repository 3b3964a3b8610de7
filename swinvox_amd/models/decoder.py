"""Decoder on HIP kernels; mirrors reference models/decoder.py:10-99.

forward([B,V,256,7,7]) -> (raw_features [B,V,9,32,32,32], gen_volumes [B,V,32,32,32]) (logits, no sigmoid).
raw_features is returned as a logical-NCDHW VIEW of a channels-last buffer [B,V,32,32,32,12] (9 channels used,
3 zero pads) so that the Merger's implicit-GEMM stencils read 16-byte channel vectors; any consumer that
needs planar memory can call .contiguous().
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_RELU, call, empty, ptr, zeros
from ._base import ConvBnAct, HipModule, conv_spec_of

VOX = 32 * 32 * 32


def raw_view(buf12: torch.Tensor, B: int, V: int) -> torch.Tensor:
    """[B*V*32768, 12] channels-last storage -> logical [B,V,9,32,32,32]."""
    return buf12.view(B, V, 32, 32, 32, 12)[..., :9].permute(0, 1, 5, 2, 3, 4)


def as_channels_last12(t: torch.Tensor) -> torch.Tensor:
    """Inverse of raw_view for any [B,V,9,32,32,32] tensor: returns [B*V*32768, 12] storage (zero pads).
    Zero-copy when `t` already is such a view, otherwise one transpose kernel."""
    B, V = t.shape[:2]
    S = VOX
    want = (V * S * 12, S * 12, 1, 1024 * 12, 32 * 12, 12)
    if t.is_cuda and tuple(t.stride()) == want and t.storage_offset() % 4 == 0:
        base = t.as_strided((B * V * S, 12), (12, 1), t.storage_offset())
        return base
    ops.hip.check_cuda(t)
    src = t.contiguous()
    dst = torch.zeros(B * V * S, 12, dtype=src.dtype, device=src.device)
    ops.transpose(src, dst, B * V, 9, S, lds=S, ldd=12, sb=9 * S, db=S * 12)   # [i][c][s] -> [i][s][c]
    return dst


class Decoder(HipModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        b = cfg.NETWORK.TCONV_USE_BIAS
        self.spatial_reduce = nn.AdaptiveAvgPool2d((2, 2))

        def tbr(cin, cout, k, p):
            return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=2, bias=b, padding=p), nn.BatchNorm3d(cout), nn.ReLU())

        self.layer1 = tbr(256, 128, (6, 4, 4), (2, 1, 1))
        self.layer2 = tbr(128, 64, 4, 1)
        self.layer3 = tbr(64, 32, 4, 1)
        self.layer4 = tbr(32, 8, 4, 1)
        self.layer5 = nn.Sequential(nn.ConvTranspose3d(8, 1, kernel_size=1, bias=b))
        self._cbas = [ConvBnAct(m[0], m[1], conv_spec_of(m[0]), ACT_RELU) for m in (self.layer1, self.layer2, self.layer3, self.layer4)]

    def forward(self, image_features):
        assert image_features.dim() == 5 and tuple(image_features.shape[2:]) == (256, 7, 7), "expected [B, V, 256, 7, 7]"
        return self._run(image_features)

    def _fwd(self, feats, save):
        B, V = feats.shape[:2]
        I = B * V
        feats = ops.to_store(feats)
        f = empty(I * 49, 256, like=feats)
        ops.transpose(feats, f, I, 256, 49)                    # NCHW -> NHWC
        seed = empty(I * 8, 256, like=f)
        call("sv_decoder_seed_fwd", ptr(f), ptr(seed), I, 256)
        x, g, ctxs = seed, (2, 2, 2), []
        for cba in self._cbas:
            x, g, c = cba.forward(x, I, g, self.training)
            ctxs.append(c)
        assert g == (32, 32, 32)
        raw12 = empty(I * VOX, 12, like=f)
        vol = empty(B, V, 32, 32, 32, like=f)
        w5 = self.layer5[0]
        call("sv_decoder_head_fwd", ptr(x), ptr(w5.weight), ptr(w5.bias), ptr(raw12), ptr(vol), I * VOX)
        tape = (B, V, ctxs, x) if save else None
        # raw_features [B,V,9,32,32,32] is by far the largest tensor that crosses a module boundary (9 x 32^3 values per view) and its only
        # consumer is Merger: it is handed over in the activation storage dtype - under bf16 storage a bf16 tensor, as the reference's
        # autocast region (core/train.py:235) returns half-precision module outputs too - which saves a bf16 -> fp32 -> bf16 round trip of
        # 2.4 GB per step at B = 64 x V = 8 in the forward and the same for its gradient.  gen_volumes stays fp32.
        return (raw_view(raw12, B, V), ops.to_f32(vol)), tape

    def _bwd(self, tape, grads, in_needs, draw, dvol):
        B, V, ctxs, x8 = tape
        I = B * V
        draw12 = ops.to_store(as_channels_last12(draw)) if draw is not None else zeros(I * VOX, 12, like=x8)
        dvol = ops.to_store(dvol) if dvol is not None else None
        w5 = self.layer5[0]
        dx = empty(I * VOX, 8, like=x8)
        call("sv_decoder_head_bwd", ptr(draw12), ptr(dvol), ptr(x8), ptr(w5.weight), ptr(dx), ptr(grads[w5.weight]),
             ptr(grads[w5.bias]) if w5.bias is not None else None, I * VOX)
        ld = 8
        for cba, c in zip(reversed(self._cbas), reversed(ctxs)):
            dx = cba.backward(c, dx, ld, grads)
            ld = cba.spec.cin_mem
        if not in_needs[0]:
            return (None,)
        df = empty(I * 49, 256, like=x8)
        call("sv_decoder_seed_bwd", ptr(dx), ptr(df), I, 256)
        dfeat = empty(B, V, 256, 7, 7, like=x8)
        ops.transpose(df, dfeat, I, 49, 256)
        return (ops.to_f32(dfeat),)
