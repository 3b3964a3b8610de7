"""Input preparation on the device: counterparts of utils/data_loaders.py:52-88 (get_datum),
utils/binvox_rw.py:105-149 and utils/data_transforms.py as composed at core/train.py:44-65.

The reference expands every sample on the CPU inside DataLoader workers (np.repeat for the voxels; cv2.resize and five numpy
passes per rendering).  Here the workers only read bytes: run-length pairs and 8-bit renderings go to the GPU as they are and
two entry points expand a whole batch - sv_binvox_decode and sv_augment_views.  Random augmentation parameters are drawn on
the host in the reference's call order (draw_train_params), so a seeded run sees the same crops / colours / flips.
"""
from __future__ import annotations

import ctypes as C
import random
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip
from .hip import call, ptr

EIGVALS = (0.2175, 0.0188, 0.0045)                                        # utils/data_transforms.py:362-364
EIGVECS = ((-0.5675, 0.7192, 0.4009), (-0.5808, -0.0045, -0.8140), (-0.5836, -0.6948, 0.4203))


# ---- binvox ----------------------------------------------------------------------------------------------------------------
def parse_binvox_header(raw: bytes) -> Tuple[List[int], List[float], float, int]:
    """utils/binvox_rw.py:105-116: returns (dims, translate, scale, offset of the run-length payload)."""
    pos, lines = 0, []
    for _ in range(5):
        end = raw.find(b"\n", pos)
        if end < 0:
            raise IOError("[ERROR] Not a binvox file")
        lines.append(raw[pos:end].strip())
        pos = end + 1
    if not lines[0].startswith(b"#binvox"):
        raise IOError("[ERROR] Not a binvox file")
    dims = list(map(int, lines[1].split(b" ")[1:]))
    translate = list(map(float, lines[2].split(b" ")[1:]))
    scale = list(map(float, lines[3].split(b" ")[1:]))[0]
    return dims, translate, scale, pos


def decode_binvox_batch(files: Sequence[bytes], device, fix_coords: bool = True, check: bool = True) -> torch.Tensor:
    """B binvox files (whole file contents) with equal dims -> float32 occupancy [B, d0, d2, d1] (xyz order; [B, d0, d1, d2]
    with fix_coords=False), i.e. read_as_3d_array(f).data.astype(np.float32) of utils/data_loaders.py:83-86 for a batch.
    check=True synchronises once to verify that every stream describes exactly d0*d1*d2 voxels."""
    if not files:
        raise ValueError("decode_binvox_batch: no files")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("swinvox_amd: tensors must live on the GPU (no CPU path exists in the product)")
    dims0, offs, chunks = None, [0], []
    for raw in files:
        dims, _, _, pos = parse_binvox_header(raw)
        if dims0 is None:
            dims0 = dims
        elif dims != dims0:
            raise ValueError(f"decode_binvox_batch: mixed volume sizes {dims0} and {dims}")
        payload = raw[pos:]
        if len(payload) % 2:
            raise ValueError("decode_binvox_batch: odd run-length payload")
        chunks.append(payload)
        offs.append(offs[-1] + len(payload) // 2)
    rle = torch.frombuffer(bytearray(b"".join(chunks)) or bytearray(2), dtype=torch.uint8).to(dev)
    off = torch.tensor(offs, dtype=torch.int64).to(dev)
    B, (d0, d1, d2) = len(files), dims0
    out = torch.zeros((B, d0, d2, d1) if fix_coords else (B, d0, d1, d2), dtype=torch.float32, device=dev)
    decoded = torch.empty(B, dtype=torch.int32, device=dev)
    call("sv_binvox_decode", ptr(rle), ptr(off), B, d0, d1, d2, 1 if fix_coords else 0, ptr(out), ptr(decoded))
    if check:
        bad = (decoded != d0 * d1 * d2).nonzero().flatten().tolist()
        if bad:
            raise ValueError(f"decode_binvox_batch: run lengths of volume(s) {bad} do not add up to {d0 * d1 * d2} voxels")
    return out


# ---- augmentation -----------------------------------------------------------------------------------------------------------
@dataclass
class AugParams:
    """Random state of one sample's transforms (shared by its V views, as in the reference) + per-view flips."""
    bg: Sequence[float]
    jitter_value: Sequence[float] = (1.0, 1.0, 1.0)       # brightness, contrast, saturation blend factors
    jitter_order: Sequence[int] = (0, 1, 2)
    noise_alpha: Sequence[float] = (0.0, 0.0, 0.0)        # PCA coefficients of RandomNoise
    flips: Sequence[bool] = field(default_factory=list)
    perm: Sequence[int] = (0, 1, 2)

    def noise_rgb(self) -> np.ndarray:
        a = np.asarray(self.noise_alpha, dtype=np.float64)   # utils/data_transforms.py:373-383
        return np.sum(np.multiply(np.multiply(np.array(EIGVECS), np.tile(a, (3, 1))), np.tile(np.array(EIGVALS), (3, 1))), axis=1)


def draw_train_params(n_views: int, cfg, rng_np=np.random, rng_py=random, n_channels: int = 4) -> AugParams:
    """One __getitem__ worth of random draws in the reference's call order: RandomBackground (data_transforms.py:425-428, plus
    the per-image random.randint of :440), ColorJitter (:276-284), RandomNoise (:372), RandomFlip (:252-255),
    RandomPermuteRGB (:67).  RandomCrop draws nothing without a bounding box (:222-231).  RandomBackground returns before it draws
    anything when the renderings have no alpha channel (:428-430): pass n_channels=3 for RGB inputs."""
    t = cfg.TRAIN
    rg = t.RANDOM_BG_COLOR_RANGE
    if n_channels == 4:
        bg = np.array([rng_np.randint(rg[i][0], rg[i][1] + 1) for i in range(3)]) / 255.0
        for _ in range(n_views):
            rng_py.randint(0, 1)
    else:
        bg = np.ones(3)                       # unused: the composite step is skipped for 3-channel images
    jv = [1 + rng_np.uniform(low=-t.BRIGHTNESS, high=t.BRIGHTNESS), 1 + rng_np.uniform(low=-t.CONTRAST, high=t.CONTRAST),
          1 + rng_np.uniform(low=-t.SATURATION, high=t.SATURATION)]
    order = np.array(range(3))
    rng_np.shuffle(order)
    alpha = rng_np.normal(loc=0, scale=t.NOISE_STD, size=3)
    flips = [bool(rng_py.randint(0, 1)) for _ in range(n_views)]
    perm = rng_np.permutation(3)
    return AugParams(bg=bg.tolist(), jitter_value=jv, jitter_order=[int(i) for i in order], noise_alpha=alpha.tolist(), flips=flips,
                     perm=[int(i) for i in perm])


def val_params(n_views: int, cfg, n_channels: int = 4) -> AugParams:
    """core/train.py:60-65: CenterCrop, RandomBackground(cfg.TEST.RANDOM_BG_COLOR_RANGE), Normalize, ToTensor (no draws for RGB inputs)."""
    rg = cfg.TEST.RANDOM_BG_COLOR_RANGE
    bg = np.array([np.random.randint(rg[i][0], rg[i][1] + 1) for i in range(3)]) / 255.0 if n_channels == 4 else np.ones(3)
    return AugParams(bg=bg.tolist(), flips=[False] * n_views)


def augment_views(images_u8: torch.Tensor, params: Sequence[AugParams], cfg) -> torch.Tensor:
    """images_u8 [B, V, Hs, Ws, C] uint8 on the GPU (C = 4 with alpha or 3, channel order as stored) -> float32
    [B, V, 3, IMG_H, IMG_W]: the tensor the reference's DataLoader hands to core/train.py:222."""
    hip.check_cuda(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 5:
        raise RuntimeError("augment_views expects uint8 [B, V, Hs, Ws, C]")
    B, V, Hs, Ws, Cc = images_u8.shape
    if len(params) != B:
        raise ValueError("augment_views: one AugParams per sample")
    images_u8 = images_u8.contiguous()
    arr = (hip.AugSample * B)()
    flips = np.zeros(B * V, dtype=np.uint8)
    mean, std = cfg.DATASET.MEAN, cfg.DATASET.STD
    for b, p in enumerate(params):
        if sorted(p.jitter_order) != [0, 1, 2] or sorted(p.perm) != [0, 1, 2] or len(p.flips) != V:
            raise ValueError("augment_views: malformed AugParams")
        nz = p.noise_rgb()
        s = arr[b]
        for c in range(3):
            s.bg[c], s.jitter_value[c], s.jitter_order[c] = float(p.bg[c]), float(p.jitter_value[c]), int(p.jitter_order[c])
            s.noise[c] = float(nz[2 - c])                     # the reference adds noise_rgb[i] to channel 2 - i (:386-390)
            s.perm[c], s.mean[c], s.std[c] = int(p.perm[c]), float(mean[c]), float(std[c])
        flips[b * V:(b + 1) * V] = np.asarray(p.flips, dtype=np.uint8)
    dev = images_u8.device
    prm = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    fl = torch.from_numpy(flips).to(dev)
    ws = torch.empty(B * V, dtype=torch.float64, device=dev)
    H, W = cfg.CONST.IMG_H, cfg.CONST.IMG_W
    out = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=dev)
    call("sv_augment_views", ptr(images_u8), B * V, V, Hs, Ws, Cc, cfg.CONST.CROP_IMG_H, cfg.CONST.CROP_IMG_W, H, W, ptr(prm), ptr(fl),
         ptr(ws), ptr(out))
    return out
