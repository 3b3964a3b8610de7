#!/usr/bin/env python3
"""views/sec, forward+backward, SwinVox-T 224^2 -> 32^3 voxels, n_views=8 (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path over one synthetic batch: Encoder -> Decoder -> Merger -> Refiner forward,
two BCE-with-logits losses, full backward (reference core/train.py:226-272 without the optimizer), plus - for
N > 1 - the bucketed RCCL gradient all-reduce.  Workload per GPU (weak scaling): B samples x V=8 views of
224x224 (BASELINE config "Full pipeline ... n_views=8, 1xMI355X"), train mode (dropout / drop-path / batch-stat
BatchNorm active), inputs resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks (MI355X_MICROARCH.md): fp32-input MFMA 157.3 TFLOP/s, bf16 ~2500 TFLOP/s; HBM3E 8 TB/s
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_VIEW = {1: 21.0, 8: 19.4, 24: 19.3}   # forward; fwd+bwd = 3x


def cpu_baseline(views: int, threads: int, batch: int = 2, steps: int = 6):
    """The CPU oracle (our restatement of the reference path, pinned against it in the build container) timed on this
    host's cores on a bounded sample: one untimed warm-up + `steps` timed fwd+bwd steps of `batch` x `views` views, fp32
    (about 10 s of CPU work on 16 cores)."""
    import oracle as O
    torch.set_num_threads(threads)
    cfg = O.default_cfg()
    nets = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
    for n in nets:
        n.apply(O.init_weights)
        n.train()
    g = torch.Generator().manual_seed(0)
    x = (0.5 * torch.randn(batch, views, 3, 224, 224, generator=g)).clamp(-1, 1)
    gt = (torch.rand(batch, 32, 32, 32, generator=g) < 0.1).float()

    def one():
        for n in nets:
            n.zero_grad(set_to_none=True)
        total, *_ = O.train_step_loss(nets, cfg, x, gt)
        total.backward()

    one()                                  # warm-up (allocator, oneDNN primitive caches)
    t0 = time.time()
    for _ in range(steps):
        one()
    dt = time.time() - t0
    return {"value": steps * batch * views / dt, "unit": "views/s", "cores": threads, "kind": "port",
            "sample": f"{steps} fwd+bwd steps (after 1 warm-up) of B={batch} x V={views} views 224x224, fp32 torch CPU oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32,
                    help="samples per GPU (x --views images each); 32 = the reference's cfg.CONST.BATCH_SIZE (config.py:64)")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--math", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--storage", default="bf16", choices=["bf16", "f32"], help="HBM element type of the activations inside the modules")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run every module on one stream (no branch / weight-gradient streams)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra untimed pass that times the engine without stream overlap")
    ap.add_argument("--detail", action="store_true", help="print the per-geometry timing table of the contraction engine to stderr")
    ap.add_argument("--cpu-views", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal hooks for a one-GPU box: SV_FORCE_DEVICE pins every rank to one card, SV_DIST_BACKEND=gloo replaces RCCL
    local = int(os.environ.get("SV_FORCE_DEVICE", local))
    backend = os.environ.get("SV_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    import swinvox_amd as S
    from swinvox_amd import hip
    from swinvox_amd.dp import GradAllReducer
    from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
    from swinvox_amd.helpers import init_weights
    from swinvox_amd.losses import bce_with_logits as bce

    hip.load()
    S.set_math(args.math)
    S.set_storage(args.storage if args.math == "bf16" else "f32")
    S.set_overlap(not args.no_overlap)
    torch.manual_seed(1234)              # the SAME weights on every rank (the reducer also broadcasts rank 0's at construction)
    cfg = S.default_cfg()
    nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    for n in nets:
        n.apply(init_weights)            # reference weight recipe (utils/helpers.py:20-44); values do not affect speed
        n.to(dev).train()
    torch.manual_seed(4321 + rank)       # per-rank stream for dropout / drop-path seeds
    reducer = GradAllReducer([nets[3], nets[2], nets[1], nets[0]]) if world > 1 else None

    B, V = args.batch, args.views
    g = torch.Generator().manual_seed(rank)
    images = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
    gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float().to(dev)

    def step():
        for n in nets:
            for p in n.parameters():
                p.grad = None
        raw, vol = nets[1](nets[0](images))
        merged = nets[2](raw, vol)
        refined = nets[3](merged)
        total = bce(merged, gt) + bce(refined, gt)
        total.backward()
        if reducer is not None:
            reducer.finish()
        return total

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    tracer = hip.Tracer({"sv_conv_gather", "sv_tconv_gather", "sv_conv_wgrad", "sv_window_attention_fwd", "sv_window_attention_bwd"})
    # HIP events bracket every engine / attention launch in the LAST two timed steps only (~800 launches): recording them in
    # every step costs ~2.4 % of the throughput (an event pair ends the back-to-back overlap of consecutive kernels)
    traced_steps = min(2, args.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        hip.TRACE = tracer if (rank == 0 and i >= args.steps - traced_steps) else None
        loss = step()
    host_dt = time.perf_counter() - t0          # host time to ENQUEUE the K steps (the GPU may still be running)
    barrier()
    dt = time.perf_counter() - t0
    hip.TRACE = None
    # the same engine launches once more WITHOUT the two-stream overlap of the encoder branches (untimed, rank 0): per-kernel
    # durations in the timed region are stretched by whatever runs beside them, this pass gives the undisturbed figures
    iso = None
    if not args.no_isolated:          # every rank runs the two extra steps (their gradient all-reduces are collectives)
        S.set_overlap(False)
        iso = hip.Tracer(tracer.names) if rank == 0 else None
        hip.TRACE = iso
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        hip.TRACE = None
        S.set_overlap(not args.no_overlap)
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    assert bool(torch.isfinite(loss)), "non-finite loss"

    if rank == 0:
        views_total = world * B * V * args.steps
        value = views_total / dt
        summ = tracer.summary()
        # dominant kernel family = the contraction engine (sv_conv_gather + sv_tconv_gather + sv_conv_wgrad)
        eng = [summ[k] for k in ("sv_conv_gather", "sv_tconv_gather", "sv_conv_wgrad") if k in summ]
        eng_ms = sum(d["ms"] for d in eng)
        eng_fl = sum(d["flops"] for d in eng)
        eng_n = sum(d["launches"] for d in eng)
        achieved = eng_fl / (eng_ms * 1e-3) / 1e12 if eng_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.math]
        eng_bytes = sum(d["bytes"] for d in eng)
        # which roofline bounds the family: algorithmic intensity against the ridge point peak_flops / peak_bytes
        ai = eng_fl / max(eng_bytes, 1.0)
        ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
        hbm_bound = ai < ridge
        gbs = eng_bytes / (eng_ms * 1e-3) / 1e9 if eng_ms > 0 else 0.0
        isolated = None
        if iso is not None:
            isum = iso.summary()
            ie = [isum[k] for k in ("sv_conv_gather", "sv_tconv_gather", "sv_conv_wgrad") if k in isum]
            ims, ifl, iby, inl = (sum(d[k] for d in ie) for k in ("ms", "flops", "bytes", "launches"))
            isolated = {"note": "same launches, encoder branches on ONE stream (untimed extra pass)", "avg_launch_us": ims * 1e3 / max(inl, 1),
                        "GB/s": iby / (ims * 1e-3) / 1e9, "TFLOP/s": ifl / (ims * 1e-3) / 1e12, "ms_per_step": ims / 2}
        # HBM traffic per launch from the committed PMC passes of this same command (profiles/, rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 wide-read correction); null when not collected
        traffic = None
        try:
            import glob
            summ_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
            pm = json.load(open(summ_files[-1])).get("pmc_traffic", {})
            traffic = [v["hbm_bytes_per_launch_corrected"] for k, v in pm.items() if k.startswith("contraction engine")][0]
        except Exception:
            traffic = None
        out = {
            "metric": "views/sec fwd+bwd SwinVox-T 224^2, 32^3 voxel, n_views=8", "value": value, "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.math == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"SwinVox-T full pipeline (ResNet50[:layer3] || Swin-T multi-stage, cross-view attention, "
                                   f"decoder, merger, refiner), fwd+2xBCE+bwd, train mode, B={B} samples x V={V} views of 224x224 per GPU",
                       "global_batch": world * B, "n_views": V, "images_per_gpu": B * V,
                       "parallelism": f"dp{world} (sample-sharded, RCCL gradient all-reduce)" if world > 1 else "single GPU",
                       "math": (f"bf16 MFMA inputs, fp32 accumulate, {S.get_storage()} activations / fp32 weights, statistics and gradients of weights in HBM"
                                if args.math == "bf16" else "exact fp32 MFMA, fp32 storage")},
            "roofline": {"bound": "hbm" if hbm_bound else "mfma",
                         "kernel": "implicit-GEMM contraction engine (igemm_kernel / wgrad_kernel: all Linear/Conv/ConvTranspose fwd+dgrad+wgrad)",
                         "achieved": gbs if hbm_bound else achieved, "peak": PEAK_HBM_GBS if hbm_bound else peak,
                         "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": (gbs / PEAK_HBM_GBS) if hbm_bound else (achieved / peak), "traffic": traffic,
                         "algorithmic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                         "mfma_tflops": achieved, "mfma_frac": achieved / peak, "isolated": isolated,
                         "algorithmic_bytes_per_launch": eng_bytes / max(eng_n, 1),
                         "algorithmic_flops_per_launch": eng_fl / max(eng_n, 1),
                         "launches_per_step": eng_n / traced_steps, "avg_launch_us": eng_ms * 1e3 / max(eng_n, 1),
                         "traced_steps": traced_steps, "share_of_step_time": (eng_ms / traced_steps) / (dt / args.steps * 1e3)},
            "kernels": {k: {"launches_per_step": v["launches"] / traced_steps, "ms_per_step": v["ms"] / traced_steps,
                            "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
                        for k, v in summ.items()},
            "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
            "model_flops_tflops_per_gpu": 3 * FWD_GFLOP_PER_VIEW.get(V, 19.4) * 1e9 * value / world / 1e12,
        }
        if args.detail:
            rows = sorted(tracer.detail().items(), key=lambda kv: -kv[1][1])
            for (name, tag), (cnt, ms, fl, by) in rows[:60]:
                print(f"{ms / traced_steps:8.3f} ms/step  x{cnt / traced_steps:5.1f}  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {by / max(ms, 1e-9) / 1e6:7.0f} GB/s  {name:18s} {tag}",
                      file=sys.stderr)
        if not args.no_cpu_baseline:
            # host share of a one-GPU box is 16 cores (os.cpu_count() reports the whole node): cap the thread pool there
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(args.cpu_views, max(1, min(ncpu, 16)))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
