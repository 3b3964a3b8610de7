#!/usr/bin/env python3
"""Golden vectors for the binvox run-length reader, produced by the REFERENCE's own utils/binvox_rw.py (numpy only, importable
in the build container).  Each case is a volume written with the reference writer and read back with the reference reader;
the fixture stores the file bytes and the decoded array (bit-packed).  Also checks oracle/data.py against the same module.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_data_golden.py      (never runs on the GPU box)"""
import importlib.util
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
from oracle import data as OD  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_binvox_rw", "/root/reference/utils/binvox_rw.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(7)
cases = {}


def blob(d, p):
    z = rng.random((d, d, d)) < p
    return z


vols = {
    "empty32": np.zeros((32, 32, 32), bool),
    "full32": np.ones((32, 32, 32), bool),                  # runs longer than 255 must be split
    "sparse32": blob(32, 0.03),
    "dense32": blob(32, 0.6),
    "slab32": np.zeros((32, 32, 32), bool),
    "asym32": np.zeros((32, 32, 32), bool),
    "cube16": blob(16, 0.2),
}
vols["slab32"][4:20, 7:9, :] = True
vols["asym32"][1, 2:5, 3:30] = True                        # distinguishes the (0, 2, 1) transposition
out = {}
for name, v in vols.items():
    model = ref.Voxels(v, list(v.shape), [0.0, 0.0, 0.0], 1.0, "xyz")
    fp = io.BytesIO()
    ref.write(model, fp)
    raw = fp.getvalue()
    back = ref.read_as_3d_array(io.BytesIO(raw)).data
    assert back.dtype == bool and np.array_equal(back, v), name
    raw_xzy = ref.read_as_3d_array(io.BytesIO(raw), fix_coords=False).data
    mine, dims, _, _ = OD.read_binvox(raw)
    assert np.array_equal(mine, back) and dims == list(v.shape), name
    assert np.array_equal(OD.read_binvox(raw, fix_coords=False)[0], raw_xzy), name
    assert OD.write_binvox(v) == raw, name                   # the restated writer emits the same bytes
    out[name + "_file"] = np.frombuffer(raw, dtype=np.uint8)
    out[name + "_xyz"] = np.packbits(back.reshape(-1))
    out[name + "_xzy"] = np.packbits(raw_xzy.reshape(-1))
    out[name + "_dims"] = np.array(v.shape)
np.savez_compressed(os.path.join(HERE, "binvox_cases.npz"), **out)
print("wrote", len(vols), "binvox cases; oracle/data.py reader and writer agree with the reference module")
