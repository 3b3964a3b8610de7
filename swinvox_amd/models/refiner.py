"""Refiner on HIP kernels; mirrors reference models/refiner.py:9-106.  forward([B,32,32,32]) -> [B,32,32,32]."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_LRELU, ACT_NONE, ACT_RELU, ConvSpec, call, empty, ptr, zeros
from ._base import ConvBnAct, HipModule, conv_spec_of


class HeadConvBnAct(ConvBnAct):
    """layer1's Conv3d(1, 32, k = 4, p = 2) + BatchNorm + LeakyReLU (reference models/refiner.py:21-26) as a (4, 1, 1)-tap convolution over the 16
    channels xc[n, z, Y, X, 4 cy + cx] = x[n, z, Y + cy - 2, X + cx - 2] (sv_head_pack_x; Y, X over the 33 output positions) - the ResNet
    stem's trick.  With one input channel the engine gathers 2-byte scalars (K = 64 from 64 separate loads) and the data gradient fills ONE
    column of a 16-wide tile: 0.38 + 0.78 + 0.41 ms per step at 64 samples.  Here the gathers are 32-byte channel vectors, the data gradient
    produces 16 real columns (dxc) that sv_head_unpack_dx folds back to the voxels, and the weight is the native one re-indexed
    [co][kz][(ky, kx)] -> [co][(ky, kx)][kz] (its gradient takes the way back)."""

    def __init__(self, conv, bn, act, slope):
        assert conv.in_channels == 1 and conv.kernel_size == (4, 4, 4) and conv.padding == (2, 2, 2) and conv.stride == (1, 1, 1)
        super().__init__(conv, bn, ConvSpec.conv3d(16, conv.out_channels, (4, 1, 1), 1, (2, 0, 0)), act, slope)
        self._w16 = self._dw16 = None

    def _packed(self, x, n, in_grid):
        D = in_grid[0]
        assert tuple(in_grid) == (D, D, D)
        co = self.spec.cout
        xc = empty(n * D * (D + 1) * (D + 1), 16, like=x)
        call("sv_head_pack_x", ptr(x), ptr(xc), n, D)
        self._w16 = torch.empty(co, 16, 4, dtype=torch.float32, device=x.device)
        ops.transpose(self.conv.weight, self._w16, co, 4, 16)
        return xc, D

    def forward(self, x, n, in_grid, training):
        xc, D = self._packed(x, n, in_grid)
        z, og, c = super().forward(xc, n, (D, D + 1, D + 1), training)
        return z, og, (c, self._w16, D, n)

    def forward_pool3d(self, x, n, in_grid, training):
        xc, D = self._packed(x, n, in_grid)
        p, og, c = super().forward_pool3d(xc, n, (D, D + 1, D + 1), training)
        return p, og, (c, self._w16, D, n)

    def backward(self, ctx, dz, lddz, grads, *, need_dx=True):
        return self._backward(ctx, grads, need_dx, lambda c: ConvBnAct.backward(self, c, dz, lddz, grads, need_dx=need_dx))

    def backward_pool3d(self, ctx, dp, grads, *, need_dx=True):
        return self._backward(ctx, grads, need_dx, lambda c: ConvBnAct.backward_pool3d(self, c, dp, grads, need_dx=need_dx))

    def _backward(self, ctx, grads, need_dx, inner):
        c, self._w16, D, n = ctx
        co = self.spec.cout
        dev = self._w16.device
        self._dw16 = torch.zeros(co, 16, 4, dtype=torch.float32, device=dev)
        dxc = inner(c)
        # the weight gradient went out on the weight-gradient stream when a module backward runs one: its way back follows it there
        aw = ops._CTX.awg
        if aw is not None:
            aw.held.append((self._dw16,))
            with torch.cuda.stream(aw.stream):
                ops.transpose(self._dw16, grads[self.conv.weight], co, 16, 4)
        else:
            ops.transpose(self._dw16, grads[self.conv.weight], co, 16, 4)
        if not need_dx:
            return None
        dx = empty(n * D * D * D, 1, like=dxc)
        call("sv_head_unpack_dx", ptr(dxc), ptr(dx), n, D)
        return dx

    def _pack(self, kind):
        return ops.pack_one(self.spec, self._w16, kind)

    def _dw(self, grads):
        return self._dw16


class Refiner(HipModule):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        lk, b = cfg.NETWORK.LEAKY_VALUE, cfg.NETWORK.TCONV_USE_BIAS

        def down(cin, cout):
            return nn.Sequential(nn.Conv3d(cin, cout, kernel_size=4, padding=2), nn.BatchNorm3d(cout), nn.LeakyReLU(lk), nn.MaxPool3d(kernel_size=2))

        def up(cin, cout):
            return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=4, stride=2, bias=b, padding=1), nn.BatchNorm3d(cout), nn.ReLU())

        self.layer1, self.layer2, self.layer3 = down(1, 32), down(32, 64), down(64, 128)
        self.layer4 = nn.Sequential(nn.Linear(8192, 2048), nn.ReLU())
        self.layer5 = nn.Sequential(nn.Linear(2048, 8192), nn.ReLU())
        self.layer6, self.layer7 = up(128, 64), up(64, 32)
        self.layer8 = nn.Sequential(nn.ConvTranspose3d(32, 1, kernel_size=4, stride=2, bias=b, padding=1))
        self._down = [HeadConvBnAct(self.layer1[0], self.layer1[1], ACT_LRELU, float(lk))]
        self._down += [ConvBnAct(m[0], m[1], conv_spec_of(m[0]), ACT_LRELU, float(lk)) for m in (self.layer2, self.layer3)]
        self._up = [ConvBnAct(m[0], m[1], conv_spec_of(m[0]), ACT_RELU) for m in (self.layer6, self.layer7)]
        self._s4, self._s5 = ConvSpec.linear(8192, 2048), ConvSpec.linear(2048, 8192)
        self._s8 = conv_spec_of(self.layer8[0], cout_mem=4)

    def forward(self, coarse_volumes):
        assert coarse_volumes.dim() == 4 and tuple(coarse_volumes.shape[1:]) == (32, 32, 32), "expected [B, 32, 32, 32]"
        return self._run(coarse_volumes)

    def _fwd(self, vol, save):
        B, tr = vol.shape[0], self.training
        v32 = ops.to_store(vol)                                            # [B,32,32,32,1] channels-last == planar
        x, g, dctx, skips = v32, (32, 32, 32), [], []
        fused = ops.bn_pool_fused_enabled()
        for cba in self._down:
            C = cba.spec.cout
            if fused:                                                      # conv k4 p2 -> 33/17/9 grid, BN over all of it, pool in the same pass
                p, og, c = cba.forward_pool3d(x, B, g, tr)
                idx = None
            else:
                z, og, c = cba.forward(x, B, g, tr)
            pg = (og[0] // 2, og[1] // 2, og[2] // 2)
            if not fused:
                p = empty(B * pg[0] * pg[1] * pg[2], C, like=vol)
                idx = torch.empty(p.numel(), dtype=torch.uint8, device=vol.device)
                call("sv_maxpool3d_fwd", ptr(z), ptr(p), ptr(idx), B, og[0], og[1], og[2], C)
            dctx.append((c, og, idx))
            skips.append(p)
            x, g = p, pg
        v16, v8, v4 = skips
        flat = empty(B, 8192, like=vol)
        ops.transpose(v4, flat, B, 64, 128)                                # [b][pos][c] -> [b][c][pos] = NCDHW flatten order
        h1 = empty(B, 2048, like=vol)
        ops.linear_fwd(flat, B, self._s4, self.layer4[0].weight, h1, bias=self.layer4[0].bias, act=ACT_RELU)
        h2 = empty(B, 8192, like=vol)
        ops.linear_fwd(h1, B, self._s5, self.layer5[0].weight, h2, bias=self.layer5[0].bias, act=ACT_RELU)
        h2t = empty(B * 64, 128, like=vol)
        ops.transpose(h2, h2t, B, 128, 64)                                 # back to channels-last
        r4 = empty(B * 64, 128, like=vol)
        call("sv_axpby", ptr(v4), ptr(h2t), ptr(r4), 1.0, 1.0, r4.numel())
        u8, _, c6 = self._up[0].forward(r4, B, (4, 4, 4), tr)
        r8 = empty(B * 512, 64, like=vol)
        call("sv_axpby", ptr(v8), ptr(u8), ptr(r8), 1.0, 1.0, r8.numel())
        u16, _, c7 = self._up[1].forward(r8, B, (8, 8, 8), tr)
        r16 = empty(B * 4096, 32, like=vol)
        call("sv_axpby", ptr(v16), ptr(u16), ptr(r16), 1.0, 1.0, r16.numel())
        c8 = self.layer8[0]
        t8 = empty(B * 32768, 1, like=vol)
        self._s8.forward(r16, B, (16, 16, 16), self._s8.pack_fwd(c8.weight), t8, ldc=1, bias=c8.bias)
        out = empty(B, 32, 32, 32, like=vol)
        call("sv_axpby", ptr(v32), ptr(t8), ptr(out), 0.5, 0.5, out.numel())
        tape = (B, v32, dctx, flat, h1, h2, r4, c6, u8, r8, c7, u16, r16) if save else None
        return ops.to_f32(out), tape

    def _bwd(self, tape, grads, in_needs, dout):
        B, v32, dctx, flat, h1, h2, r4, c6, u8, r8, c7, u16, r16 = tape
        dout = ops.to_store(dout)
        c8 = self.layer8[0]
        dt8 = zeros(B * 32768, 4, like=dout)                                # 1 real column, padded to 4 for 16-byte gathers
        ops.transpose(dout, dt8, 1, 1, B * 32768, lds=B * 32768, ldd=4)     # column 0 <- dout
        call("sv_axpby", ptr(dt8), None, ptr(dt8), 0.5, 0.0, dt8.numel())
        if c8.bias is not None:
            ops.colsum(dt8, B * 32768, 1, 4, grads[c8.bias])
        self._s8.wgrad(dt8, r16, B, (16, 16, 16), grads[c8.weight], lddy=4, ldx=32)
        dr16 = empty(B * 4096, 32, like=dout)
        self._s8.dgrad(dt8, B, (16, 16, 16), self._s8.pack_dgrad(c8.weight), dr16, lddy=4, lddx=32)
        # r16 = v16 + u16
        dr8 = self._up[1].backward(c7, dr16, 32, grads)                     # through layer7 -> d r8
        dr4 = self._up[0].backward(c6, dr8, 64, grads)                      # through layer6 -> d r4   (dr8 also feeds v8)
        # FC bottleneck: r4 = v4 + T(relu(fc5(relu(fc4(T(v4))))))
        dh2 = empty(B, 8192, like=dout)
        ops.transpose(dr4, dh2, B, 64, 128)
        # ReLU backward of layer5: dpre5 = dh2 * (h2 > 0)
        dpre5 = empty(B, 8192, like=dout)
        _relu_mask(dh2, h2, dpre5)
        ops.linear_wgrad(dpre5, h1, B, self._s5, grads[self.layer5[0].weight], grads[self.layer5[0].bias])
        dh1 = empty(B, 2048, like=dout)
        ops.linear_dgrad(dpre5, B, self._s5, self._s5.pack_dgrad(self.layer5[0].weight), dh1, act_grad_src=h1, act_grad_kind=ACT_RELU)
        ops.linear_wgrad(dh1, flat, B, self._s4, grads[self.layer4[0].weight], grads[self.layer4[0].bias])
        dflat = empty(B, 8192, like=dout)
        ops.linear_dgrad(dh1, B, self._s4, self._s4.pack_dgrad(self.layer4[0].weight), dflat)
        dv4 = empty(B * 64, 128, like=dout)
        ops.transpose(dflat, dv4, B, 128, 64)
        call("sv_axpby", ptr(dv4), ptr(dr4), ptr(dv4), 1.0, 1.0, dv4.numel())
        # down path, deepest first; skip gradients: v8 <- dr8, v16 <- dr16, v32 <- 0.5*dout
        d_skip = [dr16, dr8, dv4]
        dx = None
        for li in (2, 1, 0):
            cba = self._down[li]
            c, og, idx = dctx[li]
            C = cba.spec.cout
            dp = d_skip[li]
            if dx is not None:
                call("sv_axpby", ptr(dp), ptr(dx), ptr(dp), 1.0, 1.0, dp.numel())
            if idx is None:                                                # fused forward: the pool's backward runs inside the BatchNorm backward
                dx = cba.backward_pool3d(c, dp, grads, need_dx=(li > 0 or in_needs[0]))
                continue
            dz = empty(B * og[0] * og[1] * og[2], C, like=dout)
            call("sv_maxpool3d_bwd", ptr(dp), ptr(idx), ptr(dz), B, og[0], og[1], og[2], C)
            dx = cba.backward(c, dz, C, grads, need_dx=(li > 0 or in_needs[0]))
        if not in_needs[0]:
            return (None,)
        dvol = empty(B, 32, 32, 32, like=dout)
        call("sv_axpby", ptr(dx), ptr(dout), ptr(dvol), 1.0, 0.5, dvol.numel())
        return (ops.to_f32(dvol),)


def _relu_mask(dy, y, out):
    """out = dy * (y > 0): ReLU backward of layer5's output, which no contraction epilogue can absorb."""
    call("sv_relu_bwd", ptr(dy), ptr(y), ptr(out), dy.numel())
