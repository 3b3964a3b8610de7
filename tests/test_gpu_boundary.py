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
    assert torch.equal(a[2], b[2]) and float((a[0] - b[0]).abs()) == 0.0    # deterministic: no dropout / drop-path
