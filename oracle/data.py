"""CPU restatement of the reference's input preparation (TEST INFRASTRUCTURE ONLY - never imported by swinvox_amd/).

  read_binvox(bytes)      utils/binvox_rw.py:105-149 (read_header + read_as_3d_array) -> (bool array, dims, translate, scale).
                          PINNED: tests/golden/binvox_cases.npz was written and read back by the reference's own binvox_rw
                          (imported in the build container by tests/golden/make_data_golden.py).
  write_binvox(array)     utils/binvox_rw.py:238-283 run-length writer (runs capped at 255), used to make test streams.
  resize_linear(img, ...) cv2.resize(..., interpolation=INTER_LINEAR) for float32 images as OpenCV's resize.cpp computes it
                          (half-pixel centres, float coefficients, horizontal then vertical pass).  PARITY UNPINNED: cv2 is
                          not installed here and the reference holds no image fixtures; restated from the published algorithm.
  transform_views(...)    utils/data_transforms.py in the order of core/train.py:44-59 (RandomCrop without bounding box ->
                          RandomBackground -> ColorJitter -> RandomNoise -> Normalize -> RandomFlip -> RandomPermuteRGB ->
                          ToTensor), float64 after the resize exactly like the reference's np.append accumulation.  The
                          module imports cv2 at its top, so it cannot be imported here: PARITY UNPINNED beyond code reading.
  draw_train_params(...)  the random draws of those transforms in the reference's call order (np.random / random).
"""
from __future__ import annotations

import io
import random
from typing import Dict, List, Sequence

import numpy as np


# ---- binvox --------------------------------------------------------------------------------------------------------------
def read_binvox(raw: bytes, fix_coords: bool = True):
    fp = io.BytesIO(raw)
    line = fp.readline().strip()
    if not line.startswith(b"#binvox"):
        raise IOError("[ERROR] Not a binvox file")                      # binvox_rw.py:109-110
    dims = list(map(int, fp.readline().strip().split(b" ")[1:]))
    translate = list(map(float, fp.readline().strip().split(b" ")[1:]))
    scale = list(map(float, fp.readline().strip().split(b" ")[1:]))[0]
    fp.readline()
    rle = np.frombuffer(fp.read(), dtype=np.uint8)
    values, counts = rle[::2], rle[1::2]
    data = np.repeat(values, counts).astype(bool).reshape(dims)           # :139-141
    if fix_coords:
        data = np.transpose(data, (0, 2, 1))                              # :142-145
    return data, dims, translate, scale


def write_binvox(data: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> bytes:
    """`data` is an xyz-ordered boolean volume (what read_binvox returns)."""
    dims = list(data.shape)
    out = io.BytesIO()
    out.write(b"#binvox 1\n")
    out.write(("dim %s\n" % " ".join(map(str, dims))).encode("latin-1"))
    out.write(("translate %s\n" % " ".join(map(str, translate))).encode("latin-1"))
    out.write(("scale %s\ndata\n" % str(scale)).encode("latin-1"))
    flat = np.transpose(data.astype(np.uint8), (0, 2, 1)).flatten()      # :262-263
    # boundaries of equal-value runs, each run cut into pieces of at most 255
    edges = np.flatnonzero(np.diff(flat)) + 1
    starts = np.concatenate([[0], edges])
    ends = np.concatenate([edges, [flat.size]])
    buf = bytearray()
    for s, e in zip(starts, ends):
        n, v = int(e - s), int(flat[s])
        while n > 0:
            c = min(n, 255)
            buf += bytes((v, c))
            n -= c
    out.write(bytes(buf))
    return out.getvalue()


# ---- cv2.resize(INTER_LINEAR), float32 ---------------------------------------------------------------------------------------
def _lin_coords(dsize: int, ssize: int):
    scale = float(ssize) / float(dsize)                                   # double, like OpenCV's scale_x = 1. / inv_scale_x
    d = np.arange(dsize, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    fx = fx - s.astype(np.float32)
    lo, hi = s < 0, s >= ssize - 1
    fx[lo | hi] = 0.0
    s[lo] = 0
    s[hi] = ssize - 1
    return s, fx.astype(np.float32)


def resize_linear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img float32 [H, W, C] -> float32 [out_h, out_w, C]."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    H, W, _ = img.shape
    sx, fx = _lin_coords(out_w, W)
    sy, fy = _lin_coords(out_h, H)
    sx1, sy1 = np.minimum(sx + 1, W - 1), np.minimum(sy + 1, H - 1)
    a0, a1 = (np.float32(1) - fx)[None, :, None], fx[None, :, None]
    rows = img[:, sx] * a0 + img[:, sx1] * a1                            # horizontal pass, float32
    b0, b1 = (np.float32(1) - fy)[:, None, None], fy[:, None, None]
    return (rows[sy] * b0 + rows[sy1] * b1).astype(np.float32)            # vertical pass


# ---- transforms ------------------------------------------------------------------------------------------------------------------
def draw_train_params(n_views: int, cfg_train: Dict, rng_np=np.random, rng_py=random, n_channels: int = 4) -> Dict:
    """Random draws of one __getitem__ in the reference's call order (data_transforms.py): RandomBackground (:425-428 three
    randint, then one random.randint per image at :440), ColorJitter (:276-284), RandomNoise (:372), RandomFlip (:252-255),
    RandomPermuteRGB (:67)."""
    rg = cfg_train["RANDOM_BG_COLOR_RANGE"]
    bg = np.ones(3)
    if n_channels == 4:                                                   # :428-430: RGB renderings return before any draw
        bg = np.array([rng_np.randint(rg[i][0], rg[i][1] + 1) for i in range(3)]) / 255.0
        for _ in range(n_views):
            rng_py.randint(0, 1)                                          # drawn although no background folder is configured
    brightness = 1 + rng_np.uniform(low=-cfg_train["BRIGHTNESS"], high=cfg_train["BRIGHTNESS"])
    contrast = 1 + rng_np.uniform(low=-cfg_train["CONTRAST"], high=cfg_train["CONTRAST"])
    saturation = 1 + rng_np.uniform(low=-cfg_train["SATURATION"], high=cfg_train["SATURATION"])
    order = np.array(range(3))
    rng_np.shuffle(order)
    alpha = rng_np.normal(loc=0, scale=cfg_train["NOISE_STD"], size=3)
    flips = [bool(rng_py.randint(0, 1)) for _ in range(n_views)]
    perm = rng_np.permutation(3)
    return dict(bg=bg, jitter_value=[brightness, contrast, saturation], jitter_order=[int(i) for i in order], noise_alpha=alpha,
                flips=flips, perm=[int(i) for i in perm])


EIGVALS = np.array((0.2175, 0.0188, 0.0045))
EIGVECS = np.array(((-0.5675, 0.7192, 0.4009), (-0.5808, -0.0045, -0.8140), (-0.5836, -0.6948, 0.4203)))


def noise_rgb_of(alpha: np.ndarray) -> np.ndarray:
    """RandomNoise: data_transforms.py:373-383."""
    return np.sum(np.multiply(np.multiply(EIGVECS, np.tile(alpha, (3, 1))), np.tile(EIGVALS, (3, 1))), axis=1)


def _grey(img):
    ch = 0.114 * img[:, :, 0] + 0.587 * img[:, :, 1] + 0.299 * img[:, :, 2]
    return np.dstack((ch, ch, ch))


def transform_views(images_u8: np.ndarray, prm: Dict, img_size=(224, 224), crop_size=(128, 128), mean=(0.5, 0.5, 0.5),
                    std=(0.5, 0.5, 0.5)) -> np.ndarray:
    """images_u8 [V, Hs, Ws, C] as cv2.imread(..., IMREAD_UNCHANGED) returns them -> float32 [V, 3, H, W]."""
    out = []
    names = ["brightness", "contrast", "saturation"]
    for v, u8 in enumerate(images_u8):
        img = u8.astype(np.float32) / 255.0                                # data_loaders.py:69
        H, W, C = img.shape
        if H > crop_size[0] and W > crop_size[1]:                         # :222-231 (RandomCrop) == :135-144 (CenterCrop)
            x0, y0 = int(W - crop_size[1]) // 2, int(H - crop_size[0]) // 2
            img = img[y0:y0 + crop_size[0], x0:x0 + crop_size[1]]
        img = resize_linear(img, img_size[0], img_size[1]).astype(np.float64)
        if C == 4:                                                         # RandomBackground :437-441
            alpha = (np.expand_dims(img[:, :, 3], axis=2) == 0).astype(np.float32)
            img = alpha * np.array([[prm["bg"]]]).reshape(1, 1, 3) + (1 - alpha) * img[:, :, :3]
        for idx in prm["jitter_order"]:                                    # ColorJitter :286-328
            a = prm["jitter_value"][idx]
            gs = _grey(img)
            if names[idx] == "contrast":
                img = a * img + (1 - a) * np.mean(gs[:, :, 0])
            elif names[idx] == "saturation":
                img = a * img + (1 - a) * gs
            else:
                img = a * img + (1 - a) * 0
        nz = noise_rgb_of(np.asarray(prm["noise_alpha"], dtype=np.float64))
        img = img.copy()
        rev = img[:, :, ::-1]                                              # RandomNoise :386-390 (adds to the reversed view)
        for i in range(3):
            rev[:, :, i] += nz[i]
        img = (img - np.asarray(mean)) / np.asarray(std)                   # Normalize
        if prm["flips"][v]:
            img = np.fliplr(img)
        img = img[..., prm["perm"]]
        out.append(np.transpose(img, (2, 0, 1)))
    return np.stack(out).astype(np.float32)
