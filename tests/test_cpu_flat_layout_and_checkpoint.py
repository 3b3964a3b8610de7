"""Host logic around the hot path that needs no GPU: the flat parameter / gradient layout shared by GradStore, the flat
solvers and the gradient all-reduce (swinvox_amd/optim.py), and the checkpoint dict of core/train.py:347-369."""
import os

import pytest
import torch

import swinvox_amd as S
from swinvox_amd import harness
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
from swinvox_amd.optim import ALIGN, FlatLayout, flat_region


def test_flat_layout_offsets_views_and_locate():
    ps = [torch.randn(3, 5), torch.randn(16), torch.randn(1), torch.randn(2, 2, 2, 2, 2), torch.randn(7)]
    lay = FlatLayout(ps)
    assert all(o % ALIGN == 0 for o in lay.offsets) and lay.total % ALIGN == 0
    assert lay.offsets == [0, 16, 32, 48, 80] and lay.total == 96
    flat = torch.zeros(lay.total)
    views = lay.views(flat)
    assert [tuple(v.shape) for v in views] == [tuple(p.shape) for p in ps]
    for v, p in zip(views, ps):
        v.copy_(p)
    assert float(flat.abs().sum()) == pytest.approx(float(sum(p.abs().sum() for p in ps)), rel=1e-6)   # padding stays zero
    assert lay.locate(views) == flat.data_ptr()
    assert lay.locate(ps) is None and lay.locate([None] + views[1:]) is None
    assert lay.matches(ps) and not lay.matches(ps[:-1])
    reg = flat_region(views[1:4])
    assert reg is not None and reg.data_ptr() == views[1].data_ptr() and reg.numel() == 48 + 32 - 16
    assert flat_region(ps) is None and flat_region([views[0], views[2]]) is None     # not consecutive -> no in-place region


def test_gradstore_uses_the_layout_of_the_module():
    from swinvox_amd.models._base import GradStore
    m = Merger(S.default_cfg())
    lay = m._grad_layout()
    assert lay is m._grad_layout() and lay.matches(list(m.parameters()))
    gs = GradStore(m._param_list(), lay)
    grads = gs.as_tuple()
    assert lay.locate(grads) == gs.flat.data_ptr() and gs.flat.numel() == lay.total
    assert all(g.shape == p.shape for g, p in zip(grads, m.parameters()))


def test_checkpoint_dict_round_trip(tmp_path):
    cfg = S.default_cfg()
    torch.manual_seed(0)
    nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    ck = harness.checkpoint_dict(nets, cfg, epoch_idx=4, best_iou=0.61, best_epoch=3)
    assert set(ck) == {"epoch_idx", "best_iou", "best_epoch", "encoder_state_dict", "decoder_state_dict", "scaler_state_dict",
                       "refiner_state_dict", "merger_state_dict"}                         # core/train.py:358-369
    assert all(k.startswith("module.") for k in ck["encoder_state_dict"])                  # DataParallel keys, as the reference saves
    assert "module.layer5.0.weight" in ck["decoder_state_dict"] and "module.layer8.0.weight" in ck["refiner_state_dict"]
    path = os.path.join(tmp_path, "checkpoint-best.pth")
    torch.save(ck, path)
    torch.manual_seed(1)
    fresh = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    assert harness.load_checkpoint(fresh, cfg, path) == (4, pytest.approx(0.61), 3)
    for a, b in zip(nets, fresh):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    plain = harness.checkpoint_dict(nets, cfg, 0, -1, -1, module_prefix=False)             # un-prefixed keys load as well
    harness.load_checkpoint(fresh, cfg, plain)
    cfg.NETWORK.USE_REFINER = False
    assert "refiner_state_dict" not in harness.checkpoint_dict(nets, cfg, 0, -1, -1)


def test_flat_solvers_refuse_cpu_parameters():
    from swinvox_amd.optim import FlatAdam
    with pytest.raises(RuntimeError, match="GPU"):
        FlatAdam(torch.nn.Linear(4, 4).parameters(), lr=1e-3)


def test_taxonomy_aggregation_and_fscore_definition():
    """core/test.py:187-203 (per-taxonomy mean, sample-weighted overall mean) and :155-163 (F-score with 1e-8 epsilons)."""
    import oracle as O
    table, overall = harness.aggregate_by_taxonomy(["a", "b", "a", "a"], [[1.0, 0.0], [0.5, 0.5], [0.0, 1.0], [0.5, 0.5]])
    assert list(table) == ["a", "b"] and table["a"][0] == 3 and table["b"] == (1, [0.5, 0.5])
    assert table["a"][1] == pytest.approx([0.5, 0.5]) and overall == pytest.approx([0.5, 0.5])
    x = torch.full((1, 4, 4, 4), -9.0)
    x[0, 0, 0, :2] = 9.0                      # two predicted voxels
    g = torch.zeros(1, 4, 4, 4)
    g[0, 0, 0, 1:4] = 1                       # three true voxels, one shared: TP 1, FP 1, FN 2
    assert O.iou_at_thresholds(x, g)[0][0] == pytest.approx(1 / 4)
    assert O.fscore_at_thresholds(x, g)[0][0] == pytest.approx(2 * 0.5 * (1 / 3) / (0.5 + 1 / 3), rel=1e-6)
    assert O.fscore_at_thresholds(torch.full((1, 4, 4, 4), -9.0), torch.zeros(1, 4, 4, 4)) == [[0.0] * 4]
