"""Boundary behaviour of the product beyond the four modules' arithmetic: the HIP BCE loss as an autograd node
(core/train.py:165,249,255), re-entrancy of the launch path (two Python threads driving separate module instances at the same
time - what a threaded caller such as DataParallel's parallel_apply does), evaluate() in eval mode (core/test.py:100-104)."""
import copy
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
import swinvox_amd as S  # noqa: E402
from swinvox_amd import harness  # noqa: E402
from swinvox_amd.losses import BCEWithLogitsLoss, bce_with_logits  # noqa: E402
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner  # noqa: E402


def test_bce_with_logits_forward_and_backward(dev):
    g = torch.Generator().manual_seed(0)
    x = (3 * torch.randn(3, 32, 32, 32, generator=g)).to(dev).requires_grad_(True)
    t = (torch.rand(3, 32, 32, 32, generator=g) < 0.1).float().to(dev)
    x2 = x.detach().clone().requires_grad_(True)
    la = bce_with_logits(x, t)
    lb = torch.nn.functional.binary_cross_entropy_with_logits(x2, t)
    (2.5 * la).backward()
    (2.5 * lb).backward()
    assert abs(float(la) - float(lb)) < 1e-6 * max(1.0, abs(float(lb)))
    assert float((x.grad - x2.grad).abs().max()) < 1e-7 * max(1.0, float(x2.grad.abs().max()) * 1e3)
    # extreme logits stay finite (log1p(exp(-|x|)) form), sum of two losses backpropagates through both
    y = torch.tensor([[-80.0, 80.0, 0.0, 30.0]], device=dev, requires_grad=True)
    tt = torch.tensor([[1.0, 0.0, 1.0, 1.0]], device=dev)
    l2 = BCEWithLogitsLoss()(y, tt) + bce_with_logits(y, 1 - tt)
    l2.backward()
    assert bool(torch.isfinite(l2)) and bool(torch.isfinite(y.grad).all())
    ref = torch.nn.functional.binary_cross_entropy_with_logits(y.detach(), tt) + torch.nn.functional.binary_cross_entropy_with_logits(y.detach(), 1 - tt)
    assert abs(float(l2) - float(ref)) < 1e-4 * float(ref)
    with pytest.raises(ValueError, match="Target size"):
        bce_with_logits(y, tt[:, :2])
    with pytest.raises(RuntimeError, match="GPU"):
        bce_with_logits(torch.zeros(2), torch.zeros(2))


def _tail(dev, seed):
    cfg = S.default_cfg()
    nets = [Decoder(cfg), Merger(cfg), Refiner(cfg)]
    for i, n in enumerate(nets):
        O.seeded_weights_(n, seed=seed + i)
        n.to(dev).train()
    return nets


def _tail_step(nets, feat, gt):
    for n in nets:
        n.zero_grad(set_to_none=True)
    raw, vol = nets[0](feat)
    merged = nets[1](raw, vol)
    loss = bce_with_logits(merged, gt) + bce_with_logits(nets[2](merged), gt)
    loss.backward()
    return loss.detach()


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_two_threads_run_modules_concurrently(dev, storage):
    """Two threads, each with its own stream and its own module instances, interleave their launch sequences: per-thread call
    context (weight-pack cache, zero arena, weight-gradient stream hand-off, BatchNorm tick list, raw-stream device index) must
    not leak between them.  Results must equal the same work done sequentially."""
    S.set_math("bf16" if storage == "bf16" else "f32")
    if storage == "bf16":
        S.set_storage("bf16")
    try:
        g = torch.Generator().manual_seed(1)
        data = [((torch.randn(2, 3, 256, 7, 7, generator=g)).to(dev), (torch.rand(2, 32, 32, 32, generator=g) < 0.1).float().to(dev)),
                ((torch.randn(3, 2, 256, 7, 7, generator=g)).to(dev), (torch.rand(3, 32, 32, 32, generator=g) < 0.1).float().to(dev))]
        seq_nets = [_tail(dev, 10), _tail(dev, 20)]
        thr_nets = [copy.deepcopy(n) for n in seq_nets]
        want = []
        for nets, (feat, gt) in zip(seq_nets, data):
            for _ in range(3):
                loss = _tail_step(nets, feat, gt)
            want.append(loss)
        torch.cuda.synchronize()
        got, errs = [None, None], []
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        start = threading.Barrier(2)

        def work(i):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[i]):
                    start.wait()
                    for _ in range(3):
                        loss = _tail_step(thr_nets[i], *data[i])
                    got[i] = loss
                streams[i].synchronize()
            except Exception as e:   # noqa: BLE001
                errs.append(e)

        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())
        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        torch.cuda.synchronize()
        tol = 1e-5 if storage == "f32" else 2e-3
        for i in range(2):
            assert abs(float(got[i]) - float(want[i])) <= tol * max(1.0, abs(float(want[i])))
            for a, b in zip(seq_nets[i], thr_nets[i]):
                for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
                    sc = float(pa.grad.abs().max()) + 1e-12
                    if sc < 1e-7:      # analytically zero (conv bias in front of a train-mode BatchNorm): rounding noise only
                        assert float(pb.grad.abs().max()) < 1e-7
                        continue
                    assert float((pa.grad - pb.grad).abs().max()) <= (1e-3 if storage == "f32" else 5e-2) * sc, (i, k)
                for (k, ba), bb in zip(a.named_buffers(), b.buffers()):     # running statistics and num_batches_tracked ticked 3 times each
                    assert torch.allclose(ba.float(), bb.float(), rtol=1e-3 if storage == "f32" else 2e-2, atol=1e-5), (i, k)
    finally:
        S.set_math("f32")


def test_evaluate_runs_in_eval_mode_and_restores_training(dev):
    cfg = S.default_cfg()
    nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    for i, n in enumerate(nets):
        O.seeded_weights_(n, seed=3 + i)
        n.to(dev).train()
    g = torch.Generator().manual_seed(2)
    x = (0.5 * torch.randn(1, 2, 3, 224, 224, generator=g)).to(dev)
    gt = (torch.rand(1, 32, 32, 32, generator=g) < 0.1).float().to(dev)
    before = {k: v.clone() for n in nets for k, v in n.state_dict().items() if "running" in k or "num_batches" in k}
    a = harness.evaluate(nets, cfg, x, gt)
    b = harness.evaluate(nets, cfg, x, gt)
    assert all(n.training for n in nets)                                    # mode restored
    after = {k: v for n in nets for k, v in n.state_dict().items() if "running" in k or "num_batches" in k}
    assert all(torch.equal(before[k], after[k]) for k in before)            # no statistics update, no tick
    assert torch.equal(a[2], b[2]) and float((a[0] - b[0]).abs()) <= 1e-5 * float(a[0].abs())   # no dropout / drop-path (loss atomics reorder)


def test_golden_recipe_inside_the_product(dev):
    """swinvox_amd.goldens rebuilds the golden network on the HIP modules (name-seeded fill identical to the oracle's, calibration
    in exact-fp32 mode): the stored fp32 outputs of case_B1_V2 are reproduced - this is what bench.py's `iou_delta_vs_oracle` uses."""
    import json
    import os
    import numpy as np
    from swinvox_amd import goldens
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    o, p = O.Decoder(O.default_cfg()), Decoder(S.default_cfg())
    O.seeded_weights_(o, seed=101)
    goldens.seeded_fill_(p, 101)
    assert all(torch.equal(a, b) for a, b in zip(o.state_dict().values(), p.state_dict().values()))
    seed = json.load(open(os.path.join(gdir, "manifest.json")))["cases"]["B1_V2"]["seed"]
    S.set_math("f32")
    nets, x, gt = goldens.golden_case(dev, 1, 2, seed)
    gold = np.load(os.path.join(gdir, "case_B1_V2.npz"))
    with torch.no_grad():
        f = nets[0](x)
        raw, vol = nets[1](f)
        refined = nets[3](nets[2](raw, vol)).cpu()
    assert float((f.cpu() - torch.from_numpy(gold["features"])).abs().max()) < 1e-3 * float(np.abs(gold["features"]).max())
    assert float((refined - torch.from_numpy(gold["refined"])).abs().max()) < 3e-3
    iou, _ = harness.voxel_metrics(refined.to(dev), gt, S.default_cfg().TEST.VOXEL_THRESH)
    assert np.abs(iou.cpu().numpy() - gold["iou"]).max() < 1e-3


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_graph_replay_equals_the_eager_step(dev, storage):
    """graph.GraphedStep: one forward + BCE + backward of all four modules captured into a hipGraph (three streams, ~1 400 launches)
    and replayed.  With the stochastic layers off the replayed loss and gradients equal the eager ones; with them on every replay
    draws new dropout / drop-path masks (device-side seed epoch) while forward and backward of one replay agree."""
    from swinvox_amd.graph import GraphedStep
    S.set_math("bf16" if storage == "bf16" else "f32")
    if storage == "bf16":
        S.set_storage("bf16")
    try:
        cfg = S.default_cfg()
        nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
        for i, n in enumerate(nets):
            O.seeded_weights_(n, seed=30 + i)
            n.to(dev).train()
            n.stochastic = False
        g = torch.Generator().manual_seed(5)
        x = (0.5 * torch.randn(1, 2, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
        gt = (torch.rand(1, 32, 32, 32, generator=g) < 0.1).float().to(dev)
        xs, gs = x.clone(), gt.clone()

        def compute():
            for n in nets:
                for p in n.parameters():
                    p.grad = None
            raw, vol = nets[1](nets[0](xs))
            merged = nets[2](raw, vol)
            total = bce_with_logits(merged, gs) + bce_with_logits(nets[3](merged), gs)
            total.backward()
            return total.detach()

        bn_before = {k: v.clone() for n in nets for k, v in n.state_dict().items() if "num_batches" in k}
        want = compute().clone()
        want_g = [p.grad.clone() for n in nets for p in n.parameters()]
        step = GraphedStep(compute, static_inputs=[xs, gs], warmup=1)
        got = step(x, gt).clone()
        tol = 1e-5 if storage == "f32" else 2e-3
        assert abs(float(got) - float(want)) <= tol * max(1.0, abs(float(want)))
        worst = 0.0
        for a, p in zip(want_g, [p for n in nets for p in n.parameters()]):
            sc = float(a.abs().max())
            if sc > 1e-7:
                worst = max(worst, float((a - p.grad).abs().max()) / sc)
        assert worst < (2e-3 if storage == "f32" else 0.15), worst       # atomics reorder sums; running BN stats moved between the runs
        # a different batch through the same graph: the static inputs are overwritten, the loss follows
        x2 = (0.5 * torch.randn(1, 2, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
        other = float(step(x2, gt))
        assert abs(other - float(want)) > 1e-6
        ticks = {k: v for n in nets for k, v in n.state_dict().items() if "num_batches" in k}
        assert all(int(ticks[k]) == int(bn_before[k]) + 2 + step.replays for k in ticks)   # eager + warm-up + capture run nothing twice
        # stochastic layers on: replays differ from each other (fresh masks), stay finite
        for n in nets:
            n.stochastic = True
        step2 = GraphedStep(compute, static_inputs=[xs, gs], warmup=1)
        l1 = float(step2(x, gt)); g1 = nets[0].fusion_layer[0].weight.grad.clone()
        l2 = float(step2(x, gt)); g2 = nets[0].fusion_layer[0].weight.grad.clone()
        assert l1 != l2 and not torch.equal(g1, g2) and bool(torch.isfinite(g2).all())
    finally:
        S.set_math("f32")


def test_lr_range_test_follows_the_oracle_loop_and_restores_the_state(dev):
    """swinvox_amd.lr_finder.lr_range_test (utils/lr_finder.py:84-276 on the HIP train step) against the CPU restatement of that loop on the
    oracle modules: same learning rates, losses within the fp32 tolerance for the first steps and within 1e-2 once the updates have moved the weights visibly, the same
    suggestion rule, initial parameters back in place afterwards."""
    import oracle as O
    from oracle import lr_finder as OL
    from swinvox_amd import lr_finder as L
    from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
    S.set_math("f32")
    ocfg, pcfg = O.default_cfg(), S.default_cfg()
    onets = [O.Encoder(ocfg), O.Decoder(ocfg), O.Merger(ocfg), O.Refiner(ocfg)]
    for i, n in enumerate(onets):
        O.seeded_weights_(n, seed=70 + i)
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, O.model.SwinBlock):
                m.dp = 0.0
    pnets = [Encoder(pcfg), Decoder(pcfg), Merger(pcfg), Refiner(pcfg)]
    for p, o in zip(pnets, onets):
        p.load_state_dict(o.state_dict())
        p.to(dev).train()
        p.stochastic = False
    before = [{k: v.clone() for k, v in p.state_dict().items()} for p in pnets]
    g = torch.Generator().manual_seed(3)
    batches = [((0.5 * torch.randn(1, 2, 3, 224, 224, generator=g)).clamp(-1, 1), (torch.rand(1, 32, 32, 32, generator=g) < 0.1).float()) for _ in range(6)]
    betas = tuple(pcfg.TRAIN.BETAS)
    lrs_o, losses_o, sm_o = OL.range_test(onets, ocfg, batches, 1e-5, 3e-3, 6, 0.9, betas=betas)
    res = L.lr_range_test(pnets, pcfg, [(x.to(dev), y.to(dev)) for x, y in batches], start_lr=1e-5, end_lr=3e-3, num_batches=6, avg_beta=0.9)
    assert len(res["lrs"]) == 6 and max(abs(a - b) / b for a, b in zip(res["lrs"], lrs_o)) < 1e-12
    assert abs(losses_o[-1] - losses_o[0]) > 1e-3                                   # the sweep did train (otherwise the comparison says nothing)
    for i, (a, b) in enumerate(zip(res["losses"], losses_o)):     # every update amplifies the fp32 summation-order differences of the step before
        assert abs(a - b) < (2e-3 if i < 3 else 1e-2) * max(1.0, abs(b))
    assert max(abs(a - b) for a, b in zip(res["smoothed"], L.smooth(res["losses"], 0.9))) < 1e-12
    assert res["suggested_lr"] == L.suggest_lr(res["lrs"], res["smoothed"]) and res["diverged_at"] is None
    for p, sd in zip(pnets, before):                                                # lr_finder.py:270-276: initial state restored
        now = p.state_dict()
        assert all(torch.equal(now[k], v) for k, v in sd.items())
