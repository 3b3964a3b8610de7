"""Device-side input preparation (swinvox_amd/data.py -> sv_binvox_decode, sv_augment_views) against the fixtures written by
the reference's utils/binvox_rw.py and the numpy restatement of utils/data_transforms.py (oracle/data.py)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import swinvox_amd as S  # noqa: E402
from oracle import data as OD  # noqa: E402
from swinvox_amd import data as D  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "binvox_cases.npz"))


def golden(name):
    dims = [int(d) for d in G[name + "_dims"]]
    n = int(np.prod(dims))
    xyz = np.unpackbits(G[name + "_xyz"])[:n].astype(bool).reshape(dims[0], dims[2], dims[1])
    xzy = np.unpackbits(G[name + "_xzy"])[:n].astype(bool).reshape(dims)
    return G[name + "_file"].tobytes(), xyz, xzy


def test_binvox_batch_decode_is_bit_exact(dev):
    names = ["empty32", "full32", "sparse32", "dense32", "slab32", "asym32"]
    files, xyz, xzy = zip(*(golden(n) for n in names))
    out = D.decode_binvox_batch(files, dev)
    assert out.dtype == torch.float32 and tuple(out.shape) == (6, 32, 32, 32)
    assert np.array_equal(out.cpu().numpy(), np.stack(xyz).astype(np.float32))          # == volume.data.astype(np.float32)
    raw = D.decode_binvox_batch(files, dev, fix_coords=False)
    assert np.array_equal(raw.cpu().numpy(), np.stack(xzy).astype(np.float32))
    f16, x16, _ = golden("cube16")
    assert np.array_equal(D.decode_binvox_batch([f16], dev).cpu().numpy()[0], x16.astype(np.float32))
    # non-cubic dims and streams with thousands of short runs (more pairs than one 256-thread pass), against the oracle
    rng = np.random.default_rng(5)
    vols = [rng.random((8, 24, 40)) < p for p in (0.5, 0.5, 0.02, 0.98)]
    files = [OD.write_binvox(v) for v in vols]
    got = D.decode_binvox_batch(files, dev).cpu().numpy()
    for g, f, v in zip(got, files, vols):
        # (for non-cubic dims the reference's reshape-then-transpose is not the inverse of its writer; the reader is the contract)
        assert np.array_equal(g, OD.read_binvox(f)[0].astype(np.float32)) and g.sum() == v.sum()


def test_binvox_malformed_streams_are_rejected(dev):
    f, _, _ = golden("sparse32")
    with pytest.raises(ValueError, match="do not add up"):
        D.decode_binvox_batch([f, f[:-2]], dev)                      # one run missing
    with pytest.raises(ValueError, match="do not add up"):
        D.decode_binvox_batch([f + bytes((1, 9))], dev)              # one run too many (the reference's reshape raises)
    with pytest.raises(ValueError, match="mixed volume sizes"):
        D.decode_binvox_batch([f, golden("cube16")[0]], dev)
    with pytest.raises(ValueError, match="odd"):
        D.decode_binvox_batch([f[:-1]], dev)


def _renderings(rng, B, V, H, W, C):
    """Synthetic renderings in the style of ShapeNet's: an opaque object on a transparent background, soft alpha at its rim."""
    img = rng.integers(0, 256, size=(B, V, H, W, C), dtype=np.uint8)
    if C == 4:
        yy, xx = np.mgrid[:H, :W]
        for b in range(B):
            for v in range(V):
                cy, cx, r = rng.integers(H // 3, 2 * H // 3), rng.integers(W // 3, 2 * W // 3), rng.integers(H // 6, H // 3)
                d = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
                img[b, v, :, :, 3] = np.where(d < r, 255, np.where(d < r + 2, 120, 0))
    return img


@pytest.mark.parametrize("shape", [(2, 3, 137, 137, 4), (1, 2, 137, 137, 3), (1, 2, 100, 90, 4), (2, 1, 224, 224, 4)])
def test_augment_views_matches_the_transform_restatement(dev, shape):
    B, V, H, W, C = shape
    cfg = S.default_cfg()
    rng = np.random.default_rng(H + C)
    imgs = _renderings(rng, B, V, H, W, C)
    np.random.seed(3)
    random.seed(4)
    params = [D.draw_train_params(V, cfg) for _ in range(B)]
    params[0].flips = [True] + list(params[0].flips[1:])             # both flip states and a non-trivial permutation are covered
    params[-1].flips = list(params[-1].flips[:-1]) + [False]
    params[0].perm = [2, 0, 1]
    got = D.augment_views(torch.from_numpy(imgs).to(dev), params, cfg).cpu().numpy()
    assert got.shape == (B, V, 3, 224, 224) and got.dtype == np.float32
    for b, p in enumerate(params):
        ref = OD.transform_views(imgs[b], dict(bg=np.asarray(p.bg), jitter_value=p.jitter_value, jitter_order=list(p.jitter_order),
                                               noise_alpha=np.asarray(p.noise_alpha), flips=list(p.flips), perm=list(p.perm)))
        # the reference computes in float64 after the float32 resize, the kernel in float32 throughout
        assert np.abs(got[b] - ref).max() < 5e-6, (b, np.abs(got[b] - ref).max())
    # validation pipeline (core/train.py:60-65): fixed background, no jitter / noise / flips
    vp = [D.val_params(V, cfg) for _ in range(B)]
    gv = D.augment_views(torch.from_numpy(imgs).to(dev), vp, cfg).cpu().numpy()
    for b, p in enumerate(vp):
        ref = OD.transform_views(imgs[b], dict(bg=np.asarray(p.bg), jitter_value=[1, 1, 1], jitter_order=[0, 1, 2], noise_alpha=np.zeros(3),
                                               flips=[False] * V, perm=[0, 1, 2]))
        assert np.abs(gv[b] - ref).max() < 2e-6
    assert gv.min() >= -1.0 - 1e-6 and gv.max() <= 1.0 + 1e-6         # what core/train.py:226 clamps to anyway


def test_augment_views_argument_errors(dev):
    cfg = S.default_cfg()
    x = torch.zeros(1, 2, 137, 137, 4, dtype=torch.uint8, device=dev)
    p = D.val_params(2, cfg)
    with pytest.raises(ValueError, match="one AugParams"):
        D.augment_views(x, [p, p], cfg)
    with pytest.raises(RuntimeError, match="uint8"):
        D.augment_views(x.float(), [p], cfg)
    with pytest.raises(RuntimeError, match="GPU"):
        D.augment_views(x.cpu(), [p], cfg)
    bad = D.val_params(2, cfg)
    bad.perm = [0, 0, 1]
    with pytest.raises(ValueError, match="malformed"):
        D.augment_views(x, [bad], cfg)
    with pytest.raises(RuntimeError, match="augment_views"):
        D.augment_views(torch.zeros(1, 2, 8, 8, 2, dtype=torch.uint8, device=dev), [p], cfg)
