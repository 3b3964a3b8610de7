"""Device-side input preparation for one training batch (B x V renderings + B volumes) next to the CPU restatement of the
reference's per-sample numpy path.  python scripts/bench_data.py [--batch 32]"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S  # noqa: E402
from oracle import data as OD  # noqa: E402
from swinvox_amd import data as D  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--views", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = S.default_cfg()
rng = np.random.default_rng(0)
imgs = rng.integers(0, 256, size=(a.batch, a.views, 137, 137, 4), dtype=np.uint8)
imgs[..., 3] = np.where(rng.random((a.batch, a.views, 137, 137)) < 0.5, 0, 255)
vols = [rng.random((32, 32, 32)) < 0.05 for _ in range(a.batch)]
files = [OD.write_binvox(v) for v in vols]
params = [D.draw_train_params(a.views, cfg) for _ in range(a.batch)]
x = torch.from_numpy(imgs).to(dev)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


t_aug = timed(lambda: D.augment_views(x, params, cfg))
t_vox = timed(lambda: D.decode_binvox_batch(files, dev, check=False))
t0 = time.perf_counter()
for b in range(min(a.batch, 4)):
    p = params[b]
    OD.transform_views(imgs[b], dict(bg=np.asarray(p.bg), jitter_value=p.jitter_value, jitter_order=list(p.jitter_order),
                                     noise_alpha=np.asarray(p.noise_alpha), flips=list(p.flips), perm=list(p.perm)))
    OD.read_binvox(files[b])
t_cpu = (time.perf_counter() - t0) / min(a.batch, 4) * a.batch * 1e3
n = a.batch * a.views
print(f"augment_views   {t_aug:7.3f} ms / batch of {n} views ({n / t_aug * 1e3:9.0f} views/s, host parameter packing included)")
print(f"binvox decode   {t_vox:7.3f} ms / {a.batch} volumes (header parsing + H2D of the run-length bytes included)")
print(f"numpy restatement of the reference path, one core: {t_cpu:8.1f} ms / batch ({n / t_cpu * 1e3:7.0f} views/s)")
