"""Host logic of the learning-rate range test (swinvox_amd/lr_finder.py) against the restatement of the reference's rules
(oracle/lr_finder.py, utils/lr_finder.py:206-256 and core/train.py:335-337).  No GPU, no kernels."""
import random

import numpy as np

from oracle import lr_finder as O
from swinvox_amd import Cfg, default_cfg
from swinvox_amd import lr_finder as L


def _curve(n, seed):
    rng = np.random.default_rng(seed)
    lrs = np.geomspace(1e-7, 1e-2, n)
    loss = 1.0 - 0.5 / (1 + np.exp(-(np.log10(lrs) + 4.5) * 3)) + np.where(lrs > 2e-3, (lrs / 2e-3) ** 2 * 0.1, 0) + 0.01 * rng.standard_normal(n)
    return lrs, loss


def test_suggestion_and_smoothing_follow_the_reference_rule():
    for n, seed, beta in [(200, 0, 0.9), (1000, 1, 0.98), (60, 2, 0.5), (7, 3, 0.9)]:
        lrs, loss = _curve(n, seed)
        sm = L.smooth(list(loss), beta)
        ref = [loss[0]]
        for l in loss[1:]:
            ref.append(ref[-1] * beta + l * (1 - beta))
        assert np.allclose(sm, ref, rtol=0, atol=1e-15)
        assert L.suggest_lr(lrs, sm) == O.suggest(lrs, sm)
    assert L.suggest_lr([1, 2, 3], [3, 2, 1]) is None                      # five points or fewer: no suggestion
    lrs = np.geomspace(1e-6, 1e-2, 20)
    assert L.suggest_lr(lrs, np.linspace(1.0, 2.0, 20)) is None            # minimum at the first point: empty search window
    assert L.suggest_lr(lrs, np.linspace(2.0, 1.0, 20)) == O.suggest(lrs, np.linspace(2.0, 1.0, 20))


def test_random_view_count_per_epoch():
    cfg = default_cfg()
    cfg.CONST.N_VIEWS_RENDERING = 8
    assert L.next_n_views_rendering(cfg) == 8                               # TRAIN.UPDATE_N_VIEWS_RENDERING defaults to False (config.py:124)
    cfg.TRAIN.UPDATE_N_VIEWS_RENDERING = True
    a, b = random.Random(5), random.Random(5)
    draws = [L.next_n_views_rendering(cfg, a) for _ in range(50)]
    assert draws == [b.randint(1, 8) for _ in range(50)] and min(draws) >= 1 and max(draws) <= 8 and len(set(draws)) > 3
