"""world_size-2 data-parallel test on CPU (gloo): sample-sharded batch + bucketed gradient all-reduce must reproduce
the single-process gradients of the concatenated batch (BatchNorm in eval mode: per-rank batch statistics differ by
design, exactly as under the reference's DataParallel - SURVEY 8e).  Uses the CPU oracle modules as the model, since
swinvox_amd.dp is model-agnostic plumbing over torch.distributed."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_nets():
    import oracle as O
    cfg = O.default_cfg()
    nets = [O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
    for i, n in enumerate(nets):
        O.seeded_weights_(n, seed=50 + i)
        n.eval()            # BN uses running statistics -> the loss is a plain mean over samples
    return nets


def _loss(nets, feat, gt):
    import oracle as O
    raw, vol = nets[0](feat)
    merged = nets[1](raw, vol)
    return O.bce_logits(merged, gt) + O.bce_logits(nets[2](merged), gt)


def _data():
    g = torch.Generator().manual_seed(3)
    return torch.randn(4, 2, 256, 7, 7, generator=g), (torch.rand(4, 32, 32, 32, generator=g) < 0.1).float()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swinvox_amd.dp import GradAllReducer, shard_batch
    nets = _make_nets()
    # every rank starts from DIFFERENT weights and BatchNorm statistics: the reducer's constructor must make them rank 0's
    if rank != 0:
        with torch.no_grad():
            for n in nets:
                for t in list(n.parameters()) + list(n.buffers()):
                    t.add_(1) if t.dtype.is_floating_point else t.add_(3)
    reducer = GradAllReducer([nets[2], nets[1], nets[0]], bucket_bytes=8 << 20)   # several buckets for the refiner
    assert len(reducer.buckets) > 3
    for n, r in zip(nets, _make_nets()):
        for (k, a), (_, b) in zip(n.state_dict().items(), r.state_dict().items()):
            assert torch.equal(a, b), f"rank {rank}: {k} differs from rank 0 after construction"
    feat, gt = _data()
    for step in range(2):   # two steps: hooks / countdowns must re-arm
        for n in nets:
            n.zero_grad(set_to_none=True)
        _loss(nets, shard_batch(feat, rank, world), shard_batch(gt, rank, world)).backward()
        reducer.finish()
    out = {f"{i}.{k}": p.grad.numpy().copy() for i, n in enumerate(nets) for k, p in n.named_parameters()}
    if rank == 0:
        q.put(out)
    # gradients that are views of one flat buffer (what a HipModule's backward returns) are reduced in place
    from swinvox_amd.optim import FlatLayout
    lin = torch.nn.Linear(5, 3)
    red2 = GradAllReducer([lin])
    lay = FlatLayout(list(lin.parameters()))
    flat = torch.full((lay.total,), float(rank + 1))
    for p, v in zip(lin.parameters(), lay.views(flat)):
        p.grad = v
    red2.finish()
    assert lin.weight.grad.data_ptr() == flat.data_ptr() and all(bool((p.grad == 1.5).all()) for p in lin.parameters())
    _early_group_check(rank, world)
    dist.barrier()
    dist.destroy_process_group()


class _ToyFn(torch.autograd.Function):
    """Stand-in for models/_base.py::_ModuleFn on the CPU: ONE autograd node for the module, gradients returned as views of one
    flat buffer in the FlatLayout of the parameter list, parameter groups announced INSIDE the backward (HipModule._announce)."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod, ctx.x = mod, x
        with torch.no_grad():
            return mod.b(torch.relu(mod.a(x)))

    @staticmethod
    def backward(ctx, dout):
        from swinvox_amd.optim import FlatLayout
        mod = ctx.mod
        params = list(mod.parameters())
        with torch.enable_grad():
            gs = torch.autograd.grad(mod.b(torch.relu(mod.a(ctx.x))), params, dout)
        lay = FlatLayout(params)
        flat = torch.zeros(lay.total)
        views = lay.views(flat)
        for v, g_ in zip(views, gs):
            v.copy_(g_)
        by_id = {id(p): v for p, v in zip(params, views)}
        mod.announced = []
        for k, grp in enumerate(mod.grad_groups()):          # group 0 (the later layer) completes first, as in a real backward
            if mod.grad_ready_hook is not None:
                mod.grad_ready_hook(k, [by_id[id(p)] for p in grp])
                mod.announced.append(k)
        return (None, None) + tuple(views)


class _Toy(torch.nn.Module):
    grad_ready_hook = None

    def __init__(self):
        super().__init__()
        self.a, self.b = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)

    def grad_groups(self):
        return [list(self.b.parameters()), list(self.a.parameters())]

    def forward(self, x):
        return _ToyFn.apply(self, x, *self.parameters())


def _early_group_check(rank, world):
    """Split buckets: a module that announces parameter groups inside its backward gets each group reduced from the announcement
    (in place on its slice of the flat buffer); the post-accumulate hooks that fire afterwards must not reduce them again."""
    from swinvox_amd.dp import GradAllReducer
    torch.manual_seed(7)
    toy, tail = _Toy(), torch.nn.Linear(3, 2)
    plain = GradAllReducer([tail, toy])              # default: per-parameter hooks only, nobody listens to the announcements
    assert plain.stats()["early_groups"] == 0 and toy.grad_ready_hook is None and len(plain.buckets) == 2
    plain.remove()
    red = GradAllReducer([tail, toy], early_groups=True)
    assert len(red.buckets) == 3 and red.stats()["early_groups"] == 2 and toy.grad_ready_hook is not None
    g = torch.Generator().manual_seed(11)
    xs = torch.randn(4, 6, generator=g)
    for step in range(2):
        for m in (toy, tail):
            m.zero_grad(set_to_none=True)
        tail(toy(xs[rank * 2:(rank + 1) * 2])).square().mean().backward()
        assert toy.announced == [0, 1]
        launched = len(red._pending)
        red.finish()
        assert launched == 3, launched            # two early groups + the tail module's bucket, nothing twice
    ref_toy, ref_tail = _Toy(), torch.nn.Linear(3, 2)
    ref_toy.load_state_dict(toy.state_dict()); ref_tail.load_state_dict(tail.state_dict())
    ref_toy.grad_ready_hook = None
    ref_tail(ref_toy(xs)).square().mean().backward()          # single process, whole batch = mean of the two shards' gradients
    for (k, p), q in zip(list(toy.named_parameters()) + list(tail.named_parameters()), list(ref_toy.parameters()) + list(ref_tail.parameters())):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), (rank, k)
    # a gradient that is already in place (accumulation, zero_grad(set_to_none=False)): autograd ADDS the reduced view into another
    # buffer, so the early reduction did not reduce p.grad - the reducer must refuse instead of leaving local + world-summed sums behind
    try:
        tail(toy(xs[rank * 2:(rank + 1) * 2])).square().mean().backward()     # second backward, p.grad exists
        red.finish()
        raised = False
    except RuntimeError as e:
        raised = "early gradient groups" in str(e)
    assert raised
    for work, _, _ in red._pending:          # both ranks launched the same collectives before the refusal: let them complete
        work.wait()
    red._reset()
    red.remove()
    assert toy.grad_ready_hook is None


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_matches_single_process():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.set_num_threads(4)
    nets = _make_nets()
    feat, gt = _data()
    _loss(nets, feat, gt).backward()
    worst = (0.0, "", 0.0)
    for i, n in enumerate(nets):
        for k, p in n.named_parameters():
            ref = p.grad
            if float(ref.abs().max()) < 1e-8:      # analytically zero (e.g. the bias in front of the view softmax): rounding noise only
                assert float(torch.from_numpy(got[f"{i}.{k}"]).abs().max()) < 1e-7
                continue
            # L1-relative: different batch sizes pick different CPU conv blockings, and a max-pool arg-max flipping at a
            # near-tie re-routes single gradient elements (same effect as documented in test_gpu_modules.grad_report)
            err = float((torch.from_numpy(got[f"{i}.{k}"]) - ref).abs().sum() / (ref.abs().sum() + 1e-20))
            worst = max(worst, (err, f"{i}.{k}", float(ref.abs().max())))
    assert worst[0] < 2e-3, worst


def test_shard_batch_requires_equal_shards():
    from swinvox_amd.dp import shard_batch
    t = torch.arange(12).view(6, 2)
    assert torch.equal(shard_batch(t, 1, 3), t[2:4])
    with pytest.raises(AssertionError):
        shard_batch(t, 0, 4)
