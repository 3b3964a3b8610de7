"""Plain-PyTorch fp32 CPU restatement of the SwinVox hot path (TEST INFRASTRUCTURE ONLY).

Every class cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  The Swin-T and ResNet-50 arithmetic lives in
third-party packages that are not vendored by the reference (timm 1.0.15,
torchvision 0.21.0, versions recorded in Notebooks/SwinVox.ipynb cell 43/45);
their published algorithms are restated here and pinned as described in
oracle/__init__.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
"""
from __future__ import annotations

import math
import zlib
from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# config mirror (reference config.py:83-94 NETWORK knobs, :62-65 CONST, :109-110 TRAIN gates)
# --------------------------------------------------------------------------------------------
class Cfg(dict):
    """Attribute-access dict (easydict is not installed in this image)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg() -> Cfg:
    c = Cfg()
    c.CONST = Cfg(IMG_W=224, IMG_H=224, BATCH_SIZE=32, N_VIEWS_RENDERING=1, RNG_SEED=0)
    c.NETWORK = Cfg(
        LEAKY_VALUE=0.2, TCONV_USE_BIAS=False, USE_REFINER=True, USE_MERGER=True,
        USE_SWIN_T_MULTI_STAGE=True, SWIN_T_STAGES=[0, 1, 2, 3], USE_CROSS_VIEW_ATTENTION=True,
        CROSS_ATT_REDUCTION_RATIO=4, ATT_SPATIAL_DOWNSAMPLE_RATIO=2, CROSS_ATT_NUM_HEADS=4,
    )
    c.TRAIN = Cfg(EPOCH_START_USE_REFINER=0, EPOCH_START_USE_MERGER=0)
    c.TEST = Cfg(VOXEL_THRESH=[0.2, 0.3, 0.4, 0.5])
    return c


# --------------------------------------------------------------------------------------------
# ResNet-50 trunk (torchvision 0.21 resnet50 children[:7]; call site models/encoder.py:22-23,119)
# --------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck: 1x1 -> 3x3 (stride here) -> 1x1(x4), BN after each, residual, ReLU."""

    def __init__(self, inplanes: int, planes: int, stride: int, with_downsample: bool):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


def _res_layer(inplanes, planes, blocks, stride):
    mods = [Bottleneck(inplanes, planes, stride, True)]
    mods += [Bottleneck(planes * 4, planes, 1, False) for _ in range(blocks - 1)]
    return nn.Sequential(*mods)


class ResNetTrunk(nn.Sequential):
    """children()[:7] of resnet50 = conv1, bn1, relu, maxpool, layer1, layer2, layer3 (keys resnet.{0,1,4,5,6})."""

    def __init__(self):
        super().__init__(
            nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=False),
            nn.MaxPool2d(3, stride=2, padding=1),
            _res_layer(64, 64, 3, 1), _res_layer(256, 128, 4, 2), _res_layer(512, 256, 6, 2))


# --------------------------------------------------------------------------------------------
# Swin backbone (timm 1.0.15 swin_tiny_patch4_window7_224, features_only; models/swin_transformer.py:19-24,78)
# --------------------------------------------------------------------------------------------
def rel_pos_index(ws: int) -> torch.Tensor:
    """idx[p,q] = (y_p-y_q+ws-1)*(2ws-1) + (x_p-x_q+ws-1), tokens row-major in the window."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    y = ys.reshape(-1)
    x = xs.reshape(-1)
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


def shift_attn_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """[nW, ws*ws, ws*ws] of {0,-100}: 9 regions of the cyclically shifted map (timm SwinTransformerBlock)."""
    img = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, ws: int):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.scale = (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", rel_pos_index(ws), persistent=False)
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, xw, mask):
        Bn, N, C = xw.shape
        qkv = self.qkv(xw).view(Bn, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        a = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.reshape(-1)]
        a = a + bias.view(N, N, self.heads).permute(2, 0, 1).unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            a = (a.view(-1, nW, self.heads, N, N) + mask[None, :, None]).view(-1, self.heads, N, N)
        a = a.softmax(-1)
        return self.proj((a @ v).transpose(1, 2).reshape(Bn, N, C))


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


def drop_path(x, p: float, training: bool):
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    m = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
    return x * m / keep


class SwinBlock(nn.Module):
    def __init__(self, dim, res, heads, ws, shift, drop_path_p):
        super().__init__()
        if res <= ws:  # timm: window clipped to the map, shift disabled (stage 3 of Swin-T)
            ws, shift = res, 0
        self.res, self.ws, self.shift, self.dp = res, ws, shift, drop_path_p
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, heads, ws)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, 4 * dim)
        self.register_buffer("attn_mask", shift_attn_mask(res, res, ws, shift) if shift > 0 else None,
                             persistent=False)

    def forward(self, x):  # x [B,H,W,C]
        B, H, W, C = x.shape
        ws = self.ws
        y = self.norm1(x)
        if self.shift:
            y = torch.roll(y, shifts=(-self.shift, -self.shift), dims=(1, 2))
        yw = y.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
        yw = self.attn(yw, self.attn_mask)
        y = yw.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
        if self.shift:
            y = torch.roll(y, shifts=(self.shift, self.shift), dims=(1, 2))
        x = x + drop_path(y, self.dp, self.training)
        x = x + drop_path(self.mlp(self.norm2(x)), self.dp, self.training)
        return x


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        B, H, W, C = x.shape
        x = x.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)  # h0w0,h1w0,h0w1,h1w1
        return self.reduction(self.norm(x))


class SwinStage(nn.Module):
    def __init__(self, dim_in, dim, res, depth, heads, ws, dps, merge):
        super().__init__()
        self.downsample = PatchMerging(dim_in) if merge else nn.Identity()
        self.blocks = nn.Sequential(*[
            SwinBlock(dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2, dps[i]) for i in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class PatchEmbed(nn.Module):
    def __init__(self, in_ch, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(in_ch, dim, patch, patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(self.proj(x).permute(0, 2, 3, 1))


class _FeatureInfo:
    def __init__(self, ch):
        self._ch = list(ch)

    def channels(self):
        return list(self._ch)


class SwinBackbone(nn.Module):
    """timm FeatureListNet view of swin_tiny (or swin_base for config C5): patch_embed, layers_0..3.

    Returns the NHWC outputs of the stages listed in out_indices; final norm/head are absent
    (notebook cell 68 key list; 40,339,770-parameter KAT of cell 47).
    """

    def __init__(self, out_indices: Sequence[int] = (0, 1, 2, 3), embed_dim=96, depths=(2, 2, 6, 2),
                 heads=(3, 6, 12, 24), window=7, img_size=224, drop_path_rate=0.1, in_ch=3):
        super().__init__()
        self.out_indices = list(out_indices)
        self.patch_embed = PatchEmbed(in_ch, embed_dim, 4)
        dps = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        res, dim_in, ofs = img_size // 4, embed_dim, 0
        chans = []
        for i, d in enumerate(depths):
            dim = embed_dim * 2 ** i
            if i > 0:
                res //= 2
            if i <= max(self.out_indices):   # timm FeatureListNet (timm/models/_features.py) prunes the modules after the last out index
                setattr(self, f"layers_{i}", SwinStage(dim_in, dim, res, d, heads[i], window, dps[ofs:ofs + d], i > 0))
            ofs += d
            dim_in = dim
            chans.append(dim)
        self.n_stages = len(depths)
        self.feature_info = _FeatureInfo([chans[i] for i in self.out_indices])

    def forward(self, x):
        x = self.patch_embed(x)
        outs = []
        for i in range(max(self.out_indices) + 1):
            x = getattr(self, f"layers_{i}")(x)
            if i in self.out_indices:
                outs.append(x)
        return outs


class SwinTransformer(nn.Module):
    """models/swin_transformer.py:10-94: backbone + per-stage LayerNorm([C,H,W]) + Dropout(0.05), NHWC->NCHW."""

    def __init__(self, cfg, in_channels=3, img_size=224, pretrained=False, variant="tiny"):
        super().__init__()
        self.cfg, self.img_size = cfg, img_size
        stages = list(cfg.NETWORK.SWIN_T_STAGES)
        kw = dict(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24)) if variant == "tiny" else \
            dict(embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32))
        self.model = SwinBackbone(stages, img_size=img_size, in_ch=in_channels, **kw)
        self.out_channels = [self.model.feature_info.channels()[i] for i in range(len(stages))]
        self.out_spatial = [img_size // (4 * 2 ** i) for i in stages]
        self.layer_norm = nn.ModuleList([
            nn.LayerNorm([self.out_channels[i], self.out_spatial[i], self.out_spatial[i]]) for i in range(len(stages))])
        self.dropout = nn.Dropout(0.05)

    def forward(self, x):
        if tuple(x.shape[-2:]) != (self.img_size, self.img_size):
            x = F.interpolate(x, size=(self.img_size, self.img_size), mode="bilinear", align_corners=False)
        feats = [self.dropout(ln(f.permute(0, 3, 1, 2))) for ln, f in zip(self.layer_norm, self.model(x))]
        return feats if self.cfg.NETWORK.USE_SWIN_T_MULTI_STAGE else feats[-1]


# --------------------------------------------------------------------------------------------
# Cross-view attention (models/cross_view_attention.py:10-134)
# --------------------------------------------------------------------------------------------
class CrossViewAttention(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        n = cfg.NETWORK
        self.in_channels, self.num_heads = in_channels, n.CROSS_ATT_NUM_HEADS
        self.reduced_channels = in_channels // n.CROSS_ATT_REDUCTION_RATIO
        self.ds = n.ATT_SPATIAL_DOWNSAMPLE_RATIO
        assert self.reduced_channels % self.num_heads == 0, "reduced_channels must be divisible by num_heads"
        self.head_dim = self.reduced_channels // self.num_heads
        self.downsample_qkv = nn.Conv2d(in_channels, in_channels, self.ds, self.ds, groups=in_channels) \
            if self.ds > 1 else None
        self.qkv_conv = nn.Conv2d(in_channels, 3 * self.reduced_channels, 1)
        self.proj_conv = nn.Conv2d(self.reduced_channels, in_channels, 1)
        self.ffn = nn.Sequential(nn.Conv2d(in_channels, in_channels, 1), nn.GELU(), nn.Conv2d(in_channels, in_channels, 1))
        self.batch_norm = nn.BatchNorm2d(in_channels)
        self.dropout = nn.Dropout(0.1)

    def forward(self, x):  # [B,V,C,H,W]
        B, V, C, H, W = x.shape
        xf = x.reshape(B * V, C, H, W)
        xq = self.downsample_qkv(xf) if self.downsample_qkv is not None else xf
        h, w = xq.shape[-2:]
        R, nh = self.reduced_channels, self.num_heads
        q, k, v = self.qkv_conv(xq).split(R, dim=1)
        feat = self.head_dim * h * w                       # head = channels [32h,32h+32), feature = (c,y,x)
        q, k, v = (t.reshape(B, V, nh, feat) for t in (q, k, v))
        s = torch.einsum("bihf,bjhf->bhij", q, k) / math.sqrt(self.head_dim * V)   # :89, NOT /sqrt(feat)
        o = torch.einsum("bhij,bjhf->bihf", s.softmax(-1), v).reshape(B * V, R, h, w)
        o = self.proj_conv(o)
        if self.downsample_qkv is not None:
            o = F.interpolate(o, size=(H, W), mode="bilinear", align_corners=False)
        o = o + xf                                         # single residual (:120); FFN has none (:125)
        o = self.dropout(self.batch_norm(self.ffn(o)))
        return o.view(B, V, C, H, W)


# --------------------------------------------------------------------------------------------
# Encoder (models/encoder.py:14-164)
# --------------------------------------------------------------------------------------------
def _cbr(cin, cout, stride):
    return [nn.Conv2d(cin, cout, 3, stride=stride, padding=1), nn.BatchNorm2d(cout), nn.ReLU()]


class Encoder(nn.Module):
    def __init__(self, cfg, variant="tiny"):
        super().__init__()
        self.cfg = cfg
        n = cfg.NETWORK
        self.resnet = ResNetTrunk()
        self.swin_transformer = SwinTransformer(cfg, in_channels=3, img_size=224, variant=variant)
        self.resnet_reduce = nn.Conv2d(1024, 256, 1)
        if n.USE_SWIN_T_MULTI_STAGE:
            self.swin_stage_reduces = nn.ModuleList([nn.Conv2d(c, 256, 1) for c in self.swin_transformer.out_channels])
            chains = []
            for i in n.SWIN_T_STAGES:                     # 3/2/1/0 stride-2 blocks bring 56/28/14/7 to 7
                nblk = 3 - i if i <= 2 else 0
                chains.append(nn.Sequential(*sum([_cbr(256, 256, 2) for _ in range(nblk)], [])) if nblk else nn.Identity())
            self.swin_downsamples = nn.ModuleList(chains)
        else:
            self.swin_reduce = nn.Conv2d(self.swin_transformer.out_channels[-1], 256, 1)
        self.cross_view_attention = CrossViewAttention(cfg, 512) if n.USE_CROSS_VIEW_ATTENTION else None
        self.fusion_layer = nn.Sequential(*_cbr(512, 256, 1))
        self.layer1 = nn.Sequential(*_cbr(256, 256, 1))
        self.layer2 = nn.Sequential(*_cbr(256, 256, 1))
        self.layer3 = nn.Sequential(*_cbr(256, 256, 1))

    def forward(self, rendering_images):
        B, V, Ci, H, W = rendering_images.shape
        img = rendering_images.reshape(B * V, Ci, H, W)
        r = F.avg_pool2d(self.resnet_reduce(self.resnet(img)), 2, 2)
        s = self.swin_transformer(img)
        if self.cfg.NETWORK.USE_SWIN_T_MULTI_STAGE:
            s = sum(dn(red(f)) for f, red, dn in zip(s, self.swin_stage_reduces, self.swin_downsamples))
        else:
            s = self.swin_reduce(s)
        f = torch.cat((r, s), dim=1).view(B, V, 512, 7, 7)
        if self.cross_view_attention is not None:
            f = self.cross_view_attention(f)
        f = f.reshape(B * V, 512, 7, 7)
        f = self.layer3(self.layer2(self.layer1(self.fusion_layer(f))))
        return f.view(B, V, 256, 7, 7)


# --------------------------------------------------------------------------------------------
# Decoder (models/decoder.py:10-99)
# --------------------------------------------------------------------------------------------
def _tbr(cin, cout, k, p, bias):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, k, stride=2, padding=p, bias=bias), nn.BatchNorm3d(cout), nn.ReLU())


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        b = cfg.NETWORK.TCONV_USE_BIAS
        self.spatial_reduce = nn.AdaptiveAvgPool2d((2, 2))
        self.layer1 = _tbr(256, 128, (6, 4, 4), (2, 1, 1), b)
        self.layer2 = _tbr(128, 64, 4, 1, b)
        self.layer3 = _tbr(64, 32, 4, 1, b)
        self.layer4 = _tbr(32, 8, 4, 1, b)
        self.layer5 = nn.Sequential(nn.ConvTranspose3d(8, 1, 1, bias=b))

    def forward(self, image_features):
        B, V, C, H, W = image_features.shape
        g = self.spatial_reduce(image_features.reshape(B * V, C, H, W))         # bins [0:4],[3:7]
        g = g[:, :, None].expand(-1, -1, 2, -1, -1).contiguous()                # replicate along depth
        raw = self.layer4(self.layer3(self.layer2(self.layer1(g))))
        vol = self.layer5(raw)
        return torch.cat((raw, vol), 1).view(B, V, 9, 32, 32, 32), vol.view(B, V, 32, 32, 32)


# --------------------------------------------------------------------------------------------
# Merger (models/merger.py:9-107)
# --------------------------------------------------------------------------------------------
def _c3(cin, cout, leak):
    return nn.Sequential(nn.Conv3d(cin, cout, 3, padding=1), nn.BatchNorm3d(cout), nn.LeakyReLU(leak))


class Merger(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        lk = cfg.NETWORK.LEAKY_VALUE
        self.layer1, self.layer2, self.layer3, self.layer4 = (_c3(9, 9, lk) for _ in range(4))
        self.layer5 = _c3(36, 9, lk)
        self.layer6 = _c3(9, 1, lk)

    def forward(self, raw_features, coarse_volumes):
        B, V = raw_features.shape[:2]
        x = raw_features.reshape(B * V, 9, 32, 32, 32)
        w1 = self.layer1(x)
        w2 = self.layer2(w1)
        w3 = self.layer3(w2)
        w4 = self.layer4(w3)
        w = self.layer6(self.layer5(torch.cat((w1, w2, w3, w4), 1))).view(B, V, 32, 32, 32)
        return (coarse_volumes * w.softmax(dim=1)).sum(dim=1)


# --------------------------------------------------------------------------------------------
# Refiner (models/refiner.py:9-106)
# --------------------------------------------------------------------------------------------
class Refiner(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        lk, b = cfg.NETWORK.LEAKY_VALUE, cfg.NETWORK.TCONV_USE_BIAS

        def down(cin, cout):
            return nn.Sequential(nn.Conv3d(cin, cout, 4, padding=2), nn.BatchNorm3d(cout), nn.LeakyReLU(lk), nn.MaxPool3d(2))

        self.layer1, self.layer2, self.layer3 = down(1, 32), down(32, 64), down(64, 128)
        self.layer4 = nn.Sequential(nn.Linear(8192, 2048), nn.ReLU())
        self.layer5 = nn.Sequential(nn.Linear(2048, 8192), nn.ReLU())
        self.layer6 = _tbr(128, 64, 4, 1, b)
        self.layer7 = _tbr(64, 32, 4, 1, b)
        self.layer8 = nn.Sequential(nn.ConvTranspose3d(32, 1, 4, stride=2, padding=1, bias=b))

    def forward(self, coarse_volumes):
        v32 = coarse_volumes[:, None]
        v16 = self.layer1(v32)
        v8 = self.layer2(v16)
        v4 = self.layer3(v8)
        fc = self.layer5(self.layer4(v4.reshape(-1, 8192)))
        r4 = v4 + fc.view(-1, 128, 4, 4, 4)
        r8 = v8 + self.layer6(r4)
        r16 = v16 + self.layer7(r8)
        return ((v32 + self.layer8(r16)) * 0.5)[:, 0]


# --------------------------------------------------------------------------------------------
# weights: reference init (utils/helpers.py:20-44), calibrated recipe, seeded fill
# --------------------------------------------------------------------------------------------
def init_weights(m):
    if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu", a=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
        m.weight.data *= 0.1
    elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, 0, 0.01)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
        m.weight.data *= 0.1


def seeded_weights_(module: nn.Module, seed: int = 0, bn_jitter: bool = True) -> None:
    """Deterministic per-tensor-name fill at default-init scale (the weights are never committed).

    Conv/Linear weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch default scale); biases small;
    norm gains around 1; BN running stats perturbed so eval-mode BN is not an identity.
    """
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if not t.dtype.is_floating_point:
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_mean":
            t.copy_(0.1 * torch.randn(t.shape, generator=g) if bn_jitter else torch.zeros_like(t))
        elif leaf == "running_var":
            t.copy_(0.5 + torch.rand(t.shape, generator=g) if bn_jitter else torch.ones_like(t))
        elif leaf == "relative_position_bias_table":
            t.copy_(0.2 * torch.randn(t.shape, generator=g))
        elif t.dim() <= 1 or (leaf in ("weight", "bias") and ".layer_norm." in name):
            if leaf == "weight":     # norm gains
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            else:
                t.copy_(0.05 * torch.randn(t.shape, generator=g))
        else:
            fan_in = t[0].numel()
            bound = 1.0 / math.sqrt(max(fan_in, 1))
            t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * bound * math.sqrt(3.0))


@torch.no_grad()
def calibrate_(nets: Sequence[nn.Module], images: torch.Tensor, logit_std: float = 2.0) -> None:
    """SURVEY 8c calibration: one train-mode pass sets BN running stats to the batch statistics, then the
    last conv of decoder / refiner is rescaled so eval-mode logits have std ~= logit_std."""
    enc, dec, mer, ref = nets
    for n in nets:
        n.train()
        for m in n.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.momentum = 1.0
            if isinstance(m, nn.Dropout):
                m.p_saved, m.p = m.p, 0.0
            if isinstance(m, SwinBlock):
                m.dp_saved, m.dp = m.dp, 0.0
    raw, vol = dec(enc(images))
    ref(mer(raw, vol))
    for n in nets:
        n.eval()
        for m in n.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.momentum = 0.1
            if isinstance(m, nn.Dropout):
                m.p = m.p_saved
            if isinstance(m, SwinBlock):
                m.dp = m.dp_saved
    raw, vol = dec(enc(images))
    dec.layer5[0].weight.mul_(logit_std / float(vol.std().clamp_min(1e-12)))
    raw, vol = dec(enc(images))
    merged = mer(raw, vol)
    out = ref(merged)
    delta = out * 2 - merged                               # layer8 output
    ref.layer8[0].weight.mul_(logit_std / float(delta.std().clamp_min(1e-12)))


# --------------------------------------------------------------------------------------------
# loss / metric / step (core/train.py:165,226-261; core/test.py:141-153)
# --------------------------------------------------------------------------------------------
def bce_logits(x, t):
    return F.binary_cross_entropy_with_logits(x, t)


def iou_at_thresholds(logits: torch.Tensor, gt: torch.Tensor, ths=(0.2, 0.3, 0.4, 0.5)) -> List[List[float]]:
    """Per-sample IoU list per threshold: v = sigmoid(x) >= th, inter = sum(v*gt), union = sum((v+gt)>=1);
    1.0 when both empty (core/test.py:141-153)."""
    p = torch.sigmoid(logits)
    out = []
    for b in range(p.shape[0]):
        row = []
        for th in ths:
            v = (p[b] >= th).float()
            inter = float((v * gt[b]).sum())
            union = float(((v + gt[b]) >= 1).sum())
            row.append(1.0 if union == 0 and inter == 0 else (inter / union if union > 0 else 0.0))
        out.append(row)
    return out


def fscore_at_thresholds(logits: torch.Tensor, gt: torch.Tensor, ths=(0.2, 0.3, 0.4, 0.5)) -> List[List[float]]:
    """Per-sample F1 per threshold from TP / FP / FN of the thresholded occupancy, with the reference's 1e-8 epsilons
    (core/test.py:155-163)."""
    p = torch.sigmoid(logits)
    out = []
    for b in range(p.shape[0]):
        row = []
        for th in ths:
            v = (p[b] >= th).float()
            tp = (v * gt[b]).sum().float()
            fp = (v * (1 - gt[b])).sum().float()
            fn = ((1 - v) * gt[b]).sum().float()
            precision = tp / (tp + fp + 1e-8)
            recall = tp / (tp + fn + 1e-8)
            row.append(float(2 * precision * recall / (precision + recall + 1e-8)))
        out.append(row)
    return out


def train_step_loss(nets, cfg, images, gt, epoch_idx: int = 0):
    """Forward of one training step (core/train.py:226-261, without autocast): returns
    (total_loss, encoder_loss, refiner_loss, merged_volume, refined_volume)."""
    enc, dec, mer, ref = nets
    images = images.clamp(-1, 1)
    gt = gt.clamp(0, 1)
    raw, vol = dec(enc(images))
    if cfg.NETWORK.USE_MERGER and epoch_idx >= cfg.TRAIN.EPOCH_START_USE_MERGER:
        merged = mer(raw, vol)
    else:
        merged = vol.mean(dim=1)
    el = bce_logits(merged, gt)
    if cfg.NETWORK.USE_REFINER and epoch_idx >= cfg.TRAIN.EPOCH_START_USE_REFINER:
        refined = ref(merged)
        rl = bce_logits(refined, gt)
        total = el + rl
    else:
        refined, rl, total = merged, el, el
    return total, el, rl, merged, refined
