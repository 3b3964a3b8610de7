"""Encoder on HIP kernels; mirrors reference models/encoder.py:14-164 (same constructor, attribute names,
state_dict keys and forward signature: [B,V,3,224,224] -> [B,V,256,7,7]).

The ResNet-50 trunk restates torchvision 0.21 resnet50 children[:7] (keys resnet.{0,1,4,5,6}.*); pretrained weights
are not fetched (offline) - load a checkpoint with load_state_dict instead.
"""
from __future__ import annotations

import logging
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_NONE, ACT_RELU, BatchNormState, ConvSpec, call, empty, ptr, zeros
from ._base import ConvBnAct, HipModule, conv_spec_of
from .cross_view_attention import CrossViewAttention
from .swin_transformer import SwinTransformer, swin_backward, swin_forward


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, with_downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        self.stride = stride
        self._c1 = ConvBnAct(self.conv1, self.bn1, conv_spec_of(self.conv1), ACT_RELU)
        self._c2 = ConvBnAct(self.conv2, self.bn2, conv_spec_of(self.conv2), ACT_RELU)
        self._c3 = ConvBnAct(self.conv3, self.bn3, conv_spec_of(self.conv3), ACT_RELU)
        self._cd = ConvBnAct(self.downsample[0], self.downsample[1], conv_spec_of(self.downsample[0]), ACT_NONE) if with_downsample else None

    def fwd(self, x, n, grid, training):
        a, g1, c1 = self._c1.forward(x, n, grid, training)
        b, g2, c2 = self._c2.forward(a, n, g1, training)
        idt, cd = x, None
        if self._cd is not None:
            idt, _, cd = self._cd.forward(x, n, grid, training)
        out, g3, c3 = self._c3.forward(b, n, g2, training, residual=idt, ldr=self.conv3.out_channels)
        return out, g3, (c1, c2, c3, cd)

    def bwd(self, ctx, dout, grads):
        c1, c2, c3, cd = ctx
        Co = self.conv3.out_channels
        didt = empty(dout.shape[0], Co, like=dout)
        db = self._c3.backward(c3, dout, Co, grads, dres=didt, lddres=Co)
        da = self._c2.backward(c2, db, self.conv2.out_channels, grads)
        # the skip-path gradient joins in the epilogue of conv1's data-gradient (no separate add pass)
        skip = self._cd.backward(cd, didt, Co, grads) if cd is not None else didt
        return self._c1.backward(c1, da, self.conv1.out_channels, grads, dx_epi=dict(residual=skip, ldr=self.conv1.in_channels))


def _res_layer(inplanes, planes, blocks, stride):
    return nn.Sequential(Bottleneck(inplanes, planes, stride, True), *[Bottleneck(planes * 4, planes, 1, False) for _ in range(blocks - 1)])


class Encoder(HipModule):
    def __init__(self, cfg, variant: str = "tiny"):
        super().__init__()
        self.cfg = cfg
        n = cfg.NETWORK
        logging.info("swinvox_amd: ResNet-50 / Swin weights are randomly initialised (no network); load a checkpoint to restore them")
        self.resnet = nn.Sequential(
            nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, stride=2, padding=1), _res_layer(64, 64, 3, 1), _res_layer(256, 128, 4, 2), _res_layer(512, 256, 6, 2))
        self.swin_transformer = SwinTransformer(cfg, in_channels=3, img_size=224, pretrained=True, variant=variant)
        self.resnet_reduce = nn.Conv2d(1024, 256, kernel_size=1)
        if n.USE_SWIN_T_MULTI_STAGE:
            self.swin_stage_reduces = nn.ModuleList([nn.Conv2d(ch, 256, kernel_size=1) for ch in self.swin_transformer.out_channels])
            chains = []
            for i in n.SWIN_T_STAGES:
                nblk = 3 - i if i <= 2 else 0
                mods = []
                for _ in range(nblk):
                    mods += [nn.Conv2d(256, 256, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(256), nn.ReLU()]
                chains.append(nn.Sequential(*mods) if nblk else nn.Identity())
            self.swin_downsamples = nn.ModuleList(chains)
        else:
            self.swin_reduce = nn.Conv2d(self.swin_transformer.out_channels[-1], 256, kernel_size=1)
        self.cross_view_attention = CrossViewAttention(cfg, in_channels=512) if n.USE_CROSS_VIEW_ATTENTION else None

        def cbr(cin):
            return nn.Sequential(nn.Conv2d(cin, 256, kernel_size=3, padding=1), nn.BatchNorm2d(256), nn.ReLU())

        self.fusion_layer, self.layer1, self.layer2, self.layer3 = cbr(512), cbr(256), cbr(256), cbr(256)
        # kernel-chain helpers (no parameters of their own)
        # ResNet stem (7x7 / stride 2 / pad 3 on 3 channels) as a 4x4 / stride 1 convolution on the space-to-depth image
        # [112][112][(sy, sx, c) = 16]: input row 2*oy - 3 + ky = 2*(oy - 2 + ty) + sy with ky = 2*ty + sy - 1, i.e. pads (2, 1) and
        # an explicit 112 x 112 output grid.  Same products, but every tap is a 32-byte channel vector instead of 3 scalars.
        self._stem_spec = ConvSpec.conv2d(16, 64, 4, 1, 2, og_fixed=(1, 112, 112))
        self._s_rr = ConvSpec.linear(1024, 256)
        self._post = [ConvBnAct(m[0], m[1], conv_spec_of(m[0]), ACT_RELU) for m in (self.fusion_layer, self.layer1, self.layer2, self.layer3)]
        if n.USE_SWIN_T_MULTI_STAGE:
            self._s_red = [ConvSpec.linear(c.in_channels, 256) for c in self.swin_stage_reduces]
            self._chains = [[ConvBnAct(ch[3 * j], ch[3 * j + 1], conv_spec_of(ch[3 * j]), ACT_RELU) for j in range(len(ch) // 3)]
                            if not isinstance(ch, nn.Identity) else [] for ch in self.swin_downsamples]
        else:
            self._s_red1 = ConvSpec.linear(self.swin_reduce.in_channels, 256)

    def grad_groups(self):
        """Parameter groups in the order their gradients complete inside _bwd (each a contiguous run of the registration order):
        0 = everything behind the two backbones (resnet_reduce, Swin neck, cross-view attention, fusion + conv blocks),
        1 = the Swin backbone + stage heads, 2 = the ResNet trunk."""
        res = list(self.resnet.parameters())
        swin = list(self.swin_transformer.parameters())
        skip = {id(p) for p in res} | {id(p) for p in swin}
        tail = [p for p in self.parameters() if id(p) not in skip]
        return [tail, swin, res]

    def forward(self, rendering_images):
        assert rendering_images.dim() == 5 and rendering_images.shape[2] == 3, "expected [B, V, 3, H, W]"
        assert tuple(rendering_images.shape[-2:]) == (224, 224), "swinvox_amd: images must be 224x224 (reference cfg.CONST.IMG_H/W)"
        if self.cross_view_attention is not None and rendering_images.shape[1] > 32:
            raise RuntimeError("swinvox_amd: cross-view attention keeps all views of a sample in LDS: n_views must be <= 32 "
                               f"(got {rendering_images.shape[1]}; the reference's data sets render 24)")
        return self._run(rendering_images)

    # ---- ResNet stem on the space-to-depth image --------------------------------------------------------
    def _stem_pack(self):
        """[64,3,7,7] -> forward pack [cout][tap (ty,tx)][16 = (sy,sx,c)] of the 4x4 formulation (ky = 2ty+sy-1, kx = 2tx+sx-1)."""
        wp = empty(64 * 16 * 16, like=self.resnet[0].weight)
        call("sv_stem_pack", ptr(self.resnet[0].weight), ptr(wp), ops.hip.ACT)
        return wp

    def _stem_fwd(self, images, I, tr, x16=None):
        """stem convolution + BatchNorm + ReLU + max-pool -> the 56 x 56 x 64 map.  Fused form (default, ops.set_bn_pool_fused): the 112 x 112 x 64
        activation between BatchNorm and pool is never stored (sv_bn_act_maxpool_fwd), nor is its gradient (sv_bn_maxpool_bwd)."""
        if x16 is None:
            x16 = empty(I * 112 * 112, 16, like=images)                   # [.., (sy, sx, c)], channel 3 of every (sy, sx) group = 0
            call("sv_stem_space_to_depth", ptr(images), ptr(x16), I)
        sp, bn, M = self._stem_spec, self.resnet[1], I * 112 * 112
        y = empty(M, 64, like=x16)
        st = BatchNormState(bn, M, tr)
        sp.forward(x16, I, (1, 112, 112), self._stem_pack(), y, stats=st.sums)
        st.finalize()
        mp = empty(I * 56 * 56, 64, like=x16)
        idx = torch.empty(I * 56 * 56 * 64, dtype=torch.uint8, device=x16.device)
        fused = ops.bn_pool_fused_enabled()
        if fused:
            st.probe(y, 64)
            call("sv_bn_act_maxpool_fwd", ptr(y), ptr(st.scale), ptr(st.shift), ptr(mp), ptr(idx), I, 112, 112, 64, ACT_RELU, 0.0)
        else:
            z = empty(M, 64, like=x16)
            st.apply(y, 64, z, 64, ACT_RELU, 0.0)
            call("sv_maxpool2d_fwd", ptr(z), ptr(mp), ptr(idx), I, 112, 112, 64)
        return mp, (1, 56, 56), (x16, y, st, M, I, idx, fused)

    def _stem_bwd(self, ctx, dmp, grads):
        """dmp: gradient of the pooled 56 x 56 x 64 map"""
        x16, y, st, M, I, idx, fused = ctx
        conv, bn = self.resnet[0], self.resnet[1]
        dy = empty(M, 64, like=dmp)
        if fused:
            ws = ops.zeros_f64((ops.BN_BWD_SLOTS + 1) * 2 * 64 + 2, dmp.device)
            call("sv_bn_maxpool_bwd", ptr(dmp), ptr(idx), ptr(y), ptr(bn.weight), ptr(st.mean), ptr(st.rstd), ptr(st.scale), ptr(st.shift), I, 112, 112, 64,
                 ACT_RELU, 0.0, 1 if st.training else 0, ptr(dy), ptr(grads[bn.weight]), ptr(grads[bn.bias]), ptr(ws))
        else:
            dz = empty(M, 64, like=dmp)                                      # the gather-form max-pool backward writes every element
            call("sv_maxpool2d_bwd", ptr(dmp), ptr(idx), ptr(dz), I, 112, 112, 64)
            st.backward(dz, 64, None, 64, y, 64, dy, 64, grads[bn.weight], grads[bn.bias], ACT_RELU, 0.0)
        dw16 = ops.fzeros(64, 16, 4, 4, like=dy)                         # native layout of the 4x4 formulation: [co][(sy,sx,c)][ty][tx]
        self._stem_spec.wgrad(dy, x16, I, (1, 112, 112), dw16, async_ok=False)   # read back right below
        call("sv_stem_unpack_grad", ptr(dw16), ptr(grads[conv.weight]))  # dw[co][c][ky][kx] += dw16[co][(sy,sx,c)][ty][tx]

    # ------------------------------------------------------------------------------------------------
    def _fwd(self, images, save):
        B, V = images.shape[:2]
        I = B * V
        prep = ops.input_prep_enabled() and self.swin_transformer.model.embed_lin.cin == 48
        if not prep:
            images = ops.to_store(images)                                  # fp32 module input -> storage dtype
        tr, sto = self.training, self.stochastic
        seeds = self._seed
        multi = self.cfg.NETWORK.USE_SWIN_T_MULTI_STAGE
        x16 = xp = img = None
        if prep:    # one pass over the renderings as they arrive: the stem's space-to-depth image and the Swin patch rows (no cast, no transpose)
            images = images.contiguous()
            x16, xp = empty(I * 112 * 112, 16, like=images), empty(I * 56 * 56, 48, like=images)
            call("sv_encoder_prep", ptr(images), 1 if images.dtype == torch.float32 else 0, ptr(x16), ptr(xp), I, 224)
        else:
            img = empty(I, 224 * 224, 3, like=images)
            ops.transpose(images, img, I, 3, 224 * 224)                   # NCHW -> NHWC (coalesced both ways)
        cat = empty(I * 49, 512, like=images)                              # [resnet 256 | swin 256] concat buffer
        main = torch.cuda.current_stream()
        side = ops.side_stream(images.device) if ops.overlap_enabled() else main
        side.wait_stream(main)                                             # fork: the Swin backbone runs beside the ResNet trunk
        ready = []
        with torch.cuda.stream(side):
            feats, swin_tape = swin_forward(self.swin_transformer, img, I, tr, sto, seeds, ready, save, patches=xp)
        # ---- ResNet trunk
        x, g, c_stem = self._stem_fwd(images, I, tr, x16)                  # stem conv + BatchNorm + ReLU + max-pool
        c_blocks = []
        for li in (4, 5, 6):
            for blk in self.resnet[li]:
                x, g, c = blk.fwd(x, I, g, tr)
                c_blocks.append((blk, c))
        res_feat = x                                                       # [I*196, 1024]
        rr = empty(I * 196, 256, like=x)
        ops.linear_fwd(res_feat, I * 196, self._s_rr, self.resnet_reduce.weight, rr, bias=self.resnet_reduce.bias)
        call("sv_avgpool2_fwd", ptr(rr), ptr(cat), I, 14, 14, 256, 512, 0)
        # the multi-stage neck (stage reduces + stride-2 conv chains) belongs to the MAIN stream: the ResNet trunk is the
        # shorter branch, and each stage output is consumed as soon as its event fires while the backbone computes the next stage
        neck = self._swin_neck_fwd(feats, ready, cat, I, tr, multi, main)
        main.wait_stream(side)                                             # join
        # ---- cross-view attention
        c_cva = None
        y = cat
        if self.cross_view_attention is not None:
            y, c_cva = self.cross_view_attention.cva_forward(cat, B, V, tr, sto, seeds)
        # ---- fusion + 3 conv blocks @7x7
        c_post = []
        g = (1, 7, 7)
        for cba in self._post:
            y, g, c = cba.forward(y, I, g, tr)
            c_post.append(c)
        out = empty(B, V, 256, 7, 7, like=x)
        ops.transpose(y, out, I, 49, 256)                                  # [I][49][256] -> [I][256][49]
        tape = (B, V, c_stem, c_blocks, res_feat, rr, swin_tape, neck, c_cva, c_post) if save else None
        return ops.to_f32(out), tape

    def _swin_neck_fwd(self, feats, ready, cat, I, tr, multi, stream):
        """stage-head outputs of the Swin backbone -> cat[:, 256:] (runs on `stream`, waiting for each output's event);
        returns the neck tape."""
        img = feats[0] if isinstance(feats, list) else feats
        neck = []
        if multi:
            outs = []
            for k, f in enumerate(feats):
                stream.wait_event(ready[k])
                hw = self.swin_transformer.out_spatial[k]
                red = empty(I * hw * hw, 256, like=img)
                ops.linear_fwd(f, I * hw * hw, self._s_red[k], self.swin_stage_reduces[k].weight, red, bias=self.swin_stage_reduces[k].bias)
                y, gg, cc = red, (1, hw, hw), []
                for cba in self._chains[k]:
                    y, gg, c = cba.forward(y, I, gg, tr)
                    cc.append(c)
                assert gg == (1, 7, 7), "multi-stage neck must end at 7x7"
                outs.append(y)
                neck.append((f, red, cc))
            cat_s = cat[:, 256:]
            if len(outs) == 1:
                outs = outs + [zeros(I * 49, 256, like=img)]
            call("sv_add_n", ptr(outs[0]), ptr(outs[1]), ptr(outs[2]) if len(outs) > 2 else None, ptr(outs[3]) if len(outs) > 3 else None,
                 ptr(cat_s), I * 49, 256, 512)
        else:
            f = feats if not isinstance(feats, list) else feats[-1]
            stream.wait_event(ready[-1])
            ops.linear_fwd(f, I * 49, self._s_red1, self.swin_reduce.weight, cat, ldc=512, col_off=256, bias=self.swin_reduce.bias)
            neck.append((f, None, None))
        return neck

    def _bwd(self, tape, grads, in_needs, dout):
        B, V, c_stem, c_blocks, res_feat, rr, swin_tape, neck, c_cva, c_post = tape
        I = B * V
        multi = self.cfg.NETWORK.USE_SWIN_T_MULTI_STAGE
        dout = ops.to_store(dout)
        dy = empty(I * 49, 256, like=dout)
        ops.transpose(dout, dy, I, 256, 49)                   # [I][256][49] -> [I][49][256]
        for cba, c in zip(reversed(self._post), reversed(c_post)):
            dy = cba.backward(c, dy, 256, grads)
        dcat = dy                                                          # [I*49, 512]
        if c_cva is not None:
            dcat = self.cross_view_attention.cva_backward(c_cva, dcat, grads)
        # ---- Swin neck + backbone on the side stream, ResNet branch on the main stream
        main = torch.cuda.current_stream()
        side = ops.side_stream(dcat.device) if ops.overlap_enabled() else main
        side.wait_stream(main)                                             # fork (dcat and the zeroed gradient store are ready)
        dcat_s = dcat[:, 256:]
        dready = None
        if multi:
            # neck backward on the MAIN stream, last stage first (the order the backbone backward consumes the gradients in);
            # the side stream waits for each stage's event and meanwhile works on the later stages
            dfeats, dready = [None] * len(neck), [None] * len(neck)
            for k in reversed(range(len(neck))):
                f, red, cc = neck[k]
                hw = self.swin_transformer.out_spatial[k]
                d, ld = dcat_s, 512
                for cba, c in zip(reversed(self._chains[k]), reversed(cc)):
                    d = cba.backward(c, d, ld, grads)
                    ld = 256
                conv = self.swin_stage_reduces[k]
                sp = self._s_red[k]
                rows = I * hw * hw
                sp.wgrad(d, f, rows, (1, 1, 1), grads[conv.weight], lddy=ld, db=grads[conv.bias])
                df = empty(rows, sp.cin, like=dout)
                sp.dgrad(d, rows, (1, 1, 1), sp.pack_dgrad(conv.weight), df, lddy=ld)
                dfeats[k] = df
                dready[k] = torch.cuda.Event()
                dready[k].record(main)
        with torch.cuda.stream(side):
            if multi:
                pass
            else:
                f = neck[0][0]
                sp = self._s_red1
                sp.wgrad(dcat_s, f, I * 49, (1, 1, 1), grads[self.swin_reduce.weight], lddy=512, db=grads[self.swin_reduce.bias])
                df = empty(I * 49, sp.cin, like=dout)
                sp.dgrad(dcat_s, I * 49, (1, 1, 1), sp.pack_dgrad(self.swin_reduce.weight), df, lddy=512)
                dfeats = [None] * (len(self.swin_transformer.layer_norm) - 1) + [df]
                for i in range(len(dfeats) - 1):   # unused heads of the single-stage path receive zero gradient
                    hw, ch = self.swin_transformer.out_spatial[i], self.swin_transformer.out_channels[i]
                    dfeats[i] = zeros(I * hw * hw, ch, like=dout)
            swin_backward(self.swin_transformer, swin_tape, dfeats, I, grads, dready)
            self._announce(grads, 1)                                       # Swin backbone + stage heads are enqueued
        # ---- ResNet branch
        drr = empty(I * 196, 256, like=dout)
        call("sv_avgpool2_bwd", ptr(dcat), ptr(drr), I, 14, 14, 256, 512, 0)
        self._s_rr.wgrad(drr, res_feat, I * 196, (1, 1, 1), grads[self.resnet_reduce.weight], db=grads[self.resnet_reduce.bias])
        tail_on_main = multi or side is main                               # single-stage: swin_reduce's gradient is written on the side stream
        if tail_on_main:
            self._announce(grads, 0)                                       # neck + cross-view attention + fusion head are enqueued
        d = empty(I * 196, 1024, like=dout)
        self._s_rr.dgrad(drr, I * 196, (1, 1, 1), self._s_rr.pack_dgrad(self.resnet_reduce.weight), d)
        for blk, c in reversed(c_blocks):
            d = blk.bwd(c, d, grads)
        self._stem_bwd(c_stem, d, grads)                                   # max-pool + BatchNorm + stem conv
        self._announce(grads, 2)                                           # ResNet trunk
        main.wait_stream(side)                                             # join: every parameter gradient is complete
        if not tail_on_main:
            self._announce(grads, 0)
        return (None,)
