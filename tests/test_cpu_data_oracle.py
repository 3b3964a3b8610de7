"""Input-preparation oracle (oracle/data.py) against the fixtures the reference's own utils/binvox_rw.py produced
(tests/golden/make_data_golden.py), and the host-side logic of swinvox_amd/data.py (header parsing, random draws)."""
import os
import random

import numpy as np
import pytest

import swinvox_amd as S
from oracle import data as OD
from swinvox_amd import data as D

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "binvox_cases.npz"))
CASES = sorted(k[:-5] for k in G.files if k.endswith("_file"))


def golden(name):
    dims = [int(d) for d in G[name + "_dims"]]
    n = int(np.prod(dims))
    xyz = np.unpackbits(G[name + "_xyz"])[:n].astype(bool).reshape(dims[0], dims[2], dims[1])
    xzy = np.unpackbits(G[name + "_xzy"])[:n].astype(bool).reshape(dims)
    return G[name + "_file"].tobytes(), dims, xyz, xzy


@pytest.mark.parametrize("name", CASES)
def test_oracle_binvox_reader_and_writer_match_reference_fixtures(name):
    raw, dims, xyz, xzy = golden(name)
    got, d, tr, sc = OD.read_binvox(raw)
    assert d == dims and tr == [0.0, 0.0, 0.0] and sc == 1.0 and np.array_equal(got, xyz)
    assert np.array_equal(OD.read_binvox(raw, fix_coords=False)[0], xzy)
    assert OD.write_binvox(xyz) == raw
    assert D.parse_binvox_header(raw)[:3] == (dims, [0.0, 0.0, 0.0], 1.0)
    pos = D.parse_binvox_header(raw)[3]
    assert raw[pos - 5:pos] == b"data\n" and (len(raw) - pos) % 2 == 0


def test_binvox_header_errors():
    with pytest.raises(IOError):
        OD.read_binvox(b"#notbinvox\n")
    with pytest.raises(IOError):
        D.parse_binvox_header(b"#notbinvox 1\ndim 1 1 1\ntranslate 0 0 0\nscale 1\ndata\n")
    with pytest.raises(RuntimeError, match="GPU"):
        D.decode_binvox_batch([golden(CASES[0])[0]], "cpu")


def test_host_draws_follow_the_oracle_call_order():
    cfg = S.default_cfg()
    np.random.seed(11)
    random.seed(12)
    a = D.draw_train_params(5, cfg)
    np.random.seed(11)
    random.seed(12)
    b = OD.draw_train_params(5, dict(cfg.TRAIN))
    assert a.bg == pytest.approx(b["bg"].tolist()) and a.jitter_value == pytest.approx(b["jitter_value"])
    assert list(a.jitter_order) == b["jitter_order"] and list(a.perm) == b["perm"] and list(a.flips) == b["flips"]
    assert a.noise_alpha == pytest.approx(b["noise_alpha"].tolist())
    assert a.noise_rgb() == pytest.approx(OD.noise_rgb_of(b["noise_alpha"]))
    assert all(225 / 255 <= v <= 1.0 for v in a.bg) and all(0.6 < v < 1.4 for v in a.jitter_value)
    # RGB renderings: RandomBackground draws nothing (data_transforms.py:428-430), so the jitter sees the first uniform draws
    np.random.seed(11)
    random.seed(12)
    c = D.draw_train_params(5, cfg, n_channels=3)
    np.random.seed(11)
    random.seed(12)
    d = OD.draw_train_params(5, dict(cfg.TRAIN), n_channels=3)
    np.random.seed(11)
    first = 1 + np.random.uniform(low=-cfg.TRAIN.BRIGHTNESS, high=cfg.TRAIN.BRIGHTNESS)
    assert c.jitter_value[0] == pytest.approx(first) and c.jitter_value == pytest.approx(d["jitter_value"])
    assert list(c.flips) == d["flips"] and c.jitter_value != pytest.approx(a.jitter_value)
    v = D.val_params(3, cfg)
    assert v.bg == pytest.approx([240 / 255] * 3) and list(v.flips) == [False] * 3 and list(v.jitter_value) == [1.0, 1.0, 1.0]


def test_resize_restatement_properties():
    """cv2.resize(INTER_LINEAR) restatement: identity at equal size, constants preserved, corners clamp, half-pixel centres."""
    rng = np.random.default_rng(0)
    img = rng.random((128, 128, 4)).astype(np.float32)
    assert np.array_equal(OD.resize_linear(img, 128, 128), img)
    assert np.allclose(OD.resize_linear(np.full((9, 7, 3), 0.37, np.float32), 20, 31), 0.37, atol=1e-7)
    up = OD.resize_linear(img, 224, 224)
    assert up.shape == (224, 224, 4) and up.dtype == np.float32
    assert up[0, 0, 0] == pytest.approx(img[0, 0, 0]) and up[-1, -1, 2] == pytest.approx(img[-1, -1, 2])
    ramp = np.tile(np.arange(4, dtype=np.float32)[None, :, None], (1, 1, 1))
    got = OD.resize_linear(ramp, 1, 8)[0, :, 0]
    assert got == pytest.approx([0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.0])
