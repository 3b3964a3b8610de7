#!/usr/bin/env python3
"""views/sec, forward+backward, SwinVox-T 224^2 -> 32^3 voxels, n_views=8 (BASELINE.json metric) on N MI355X.

One "step" = one pass of the hot path over one synthetic batch: Encoder -> Decoder -> Merger -> Refiner forward,
two BCE-with-logits losses (HIP kernel), full backward (reference core/train.py:226-272 without the optimizer), plus - for
N > 1 - the bucketed RCCL gradient all-reduce.  Workload per GPU (weak scaling): B samples x V=8 views of
224x224 (BASELINE config "Full pipeline ... n_views=8, 1xMI355X"), train mode (dropout / drop-path / batch-stat
BatchNorm active), inputs resident in HBM before the timed region.

The timed region enqueues every launch from the host (~1 300 launches on three HIP streams; the host needs ~21 ms per step with
an idle queue, the GPU ~70 ms, so the host runs ahead).  `--graph` replays the step from ONE hipGraph captured on a single stream
(swinvox_amd/graph.py: host cost 0.4 ms per step; measured round 2: 77.6 ms per step against 72.7 eager, because the single-stream
graph gives up the overlap of the encoder branches, and a three-stream capture replays slower than eager on ROCm 7.2: 96.8 ms).

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
FWD_GFLOP_PER_VIEW = {1: 21.0, 8: 19.4, 24: 19.3}   # forward, Swin-T; fwd+bwd = 3x
ENGINE = ("sv_conv_gather", "sv_tconv_gather", "sv_conv_wgrad")


def cpu_baseline(threads: int):
    """The CPU oracle (our restatement of the reference path, pinned against it in the build container) timed on this host's
    cores on a BOUNDED sample of the workload: fwd+bwd steps of B=2 x V=8 views (a reduced configuration 3) in fp32 and under
    torch.autocast('cpu', bfloat16) - the mode the reference's CPU path really runs in (core/train.py:235; GradScaler disables
    itself without CUDA) - and of B=2 x V=1 (BASELINE configuration 1, the reference's own CPU-runnable case).  The headline
    `value` is the fp32 V=8 leg (parity is defined against fp32)."""
    import oracle as O
    torch.set_num_threads(threads)
    cfg = O.default_cfg()
    nets = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
    for n in nets:
        n.apply(O.init_weights)
        n.train()
    g = torch.Generator().manual_seed(0)

    def leg(batch, views, autocast, steps):
        x = (0.5 * torch.randn(batch, views, 3, 224, 224, generator=g)).clamp(-1, 1)
        gt = (torch.rand(batch, 32, 32, 32, generator=g) < 0.1).float()

        def one():
            for n in nets:
                n.zero_grad(set_to_none=True)
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                total, *_ = O.train_step_loss(nets, cfg, x, gt)
            total.backward()

        one()                                  # warm-up (allocator, oneDNN primitive caches)
        t0 = time.time()
        for _ in range(steps):
            one()
        dt = time.time() - t0
        return steps * batch * views / dt, dt

    v8, t8 = leg(2, 8, False, 4)
    v8b, t8b = leg(2, 8, True, 3)
    v1, t1 = leg(2, 1, False, 6)
    return {"value": v8, "unit": "views/s", "cores": threads, "kind": "port",
            "sample": f"fp32 torch CPU oracle, fwd+bwd, after 1 warm-up step each: 4 steps of B=2 x V=8 views 224x224 ({t8:.1f} s); "
                      f"3 steps under torch.autocast('cpu', bfloat16) ({t8b:.1f} s); 6 steps of B=2 x V=1 = BASELINE config 1 ({t1:.1f} s)",
            "autocast_bf16_views_per_s": v8b, "config1_B2_V1_views_per_s": v1}


def bf16_parity_report(dev):
    """IoU / logit deviation of the benchmarked mode (bf16 MFMA + bf16 storage), whole pipeline, n_views = 8, eval forward on the
    golden inputs against the committed fp32 golden vectors (tests/golden/case_B2_V8.npz).  The weights come from the golden
    recipe (name-seeded fill + calibration, swinvox_amd/goldens.py: HIP modules only, no oracle).  Reported, not asserted."""
    try:
        import numpy as np
        import swinvox_amd as S
        from swinvox_amd.goldens import golden_case
        from swinvox_amd.harness import voxel_metrics
        gold = np.load(os.path.join(ROOT, "tests", "golden", "case_B2_V8.npz"))
        seed = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["cases"]["B2_V8"]["seed"]
        nets, x, gt = golden_case(dev, 2, 8, seed)
        with torch.no_grad():
            raw, vol = nets[1](nets[0](x))
            refined = nets[3](nets[2](raw, vol))
        ref = torch.from_numpy(gold["refined"]).to(dev)
        iou, _ = voxel_metrics(refined, gt, S.default_cfg().TEST.VOXEL_THRESH)
        d = (refined - ref).abs()
        return {"case": "B2_V8 golden inputs, eval forward, vs fp32 golden vectors", "max_abs_dlogit": float(d.max()),
                "mean_abs_dlogit": float(d.mean()), "logit_absmax": float(ref.abs().max()),
                "max_abs_dIoU": float(np.abs(iou.cpu().numpy() - gold["iou"]).max())}
    except Exception as e:   # noqa: BLE001  (fixtures absent: the number is simply not reported)
        return {"unavailable": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64,
                    help="samples per GPU (x --views images each).  64 x 8 views = 512 images per step use ~76 GB of the 288 GB; the "
                         "reference's cfg.CONST.BATCH_SIZE (config.py:64) is 32 for its 16-GB-class Colab GPU (round 2: 3 560 views/s at 32, 3 774 at 64)")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--variant", default="tiny", choices=["tiny", "base"], help="Swin-T (the metric) or Swin-B (BASELINE config 5)")
    ap.add_argument("--math", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--storage", default="bf16", choices=["bf16", "f32"], help="HBM element type of the activations inside the modules")
    ap.add_argument("--fp8-attention", action="store_true", help="QK^T / PV of the window attention forward on fp8 (e4m3) MFMA operands (BASELINE config 5)")
    ap.add_argument("--graph", action="store_true", help="replay the step from one hipGraph (captured on a single stream) instead of enqueueing every launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run every module on one stream (no branch / weight-gradient streams)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra untimed pass that times the engine without stream overlap")
    ap.add_argument("--no-parity", action="store_true", help="skip the bf16-vs-golden IoU report")
    ap.add_argument("--profile", action="store_true", help="for rocprofv3 runs: warm-up + timed steps only, no instrumentation passes, short JSON line")
    ap.add_argument("--detail", action="store_true", help="print the per-geometry timing table of the contraction engine to stderr")
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
    from swinvox_amd.graph import GraphedStep
    from swinvox_amd.helpers import init_weights
    from swinvox_amd.losses import bce_with_logits as bce
    from swinvox_amd.models import Decoder, Encoder, Merger, Refiner

    hip.load()
    S.set_math(args.math)
    S.set_storage(args.storage if args.math == "bf16" else "f32")
    S.set_overlap(not args.no_overlap)
    S.set_attention_fp8(args.fp8_attention)
    use_graph = args.graph
    early = False
    torch.manual_seed(1234)              # the SAME weights on every rank (the reducer also broadcasts rank 0's at construction)
    cfg = S.default_cfg()
    nets = [Encoder(cfg, variant=args.variant), Decoder(cfg), Merger(cfg), Refiner(cfg)]
    for n in nets:
        n.apply(init_weights)            # reference weight recipe (utils/helpers.py:20-44); values do not affect speed
        n.to(dev).train()
    torch.manual_seed(4321 + rank)       # per-rank stream for dropout / drop-path seeds
    order = [nets[3], nets[2], nets[1], nets[0]]      # the order the modules' gradients complete in
    # eager: buckets launched from hooks inside the backward; graph replay: reduce_all() after the replay
    # Default = the plain schedule (a bucket starts from the post-accumulate hook of its last gradient): the one that has run on a GPU.
    # SV_DP_EARLY=1 opts in to the encoder's early parameter groups (their all-reduce starts inside the encoder backward on a staging
    # stream; <= 4 ms of a 113 ms step at stake) - kept off until a hardware RCCL run of the plain schedule exists.
    early = os.environ.get("SV_DP_EARLY", "0") != "0"
    reducer = GradAllReducer(order, hooks=not use_graph, early_groups=early) if world > 1 else None
    param_checksum_spread = 0.0
    if world > 1:
        # after construction (rank 0's parameters broadcast) every rank must hold the same model: all-reduce a checksum, max - min = 0.
        # On a mismatch the process exits non-zero; nothing is retried in a process that has touched the GPU.
        cs = torch.stack([p.detach().double().sum() for n in nets for p in n.parameters()]).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        param_checksum_spread = float(hi - lo)
        if param_checksum_spread != 0.0:
            raise SystemExit(f"bench.py: ranks hold different parameters after the broadcast (checksum spread {param_checksum_spread})")

    B, V = args.batch, args.views
    g = torch.Generator().manual_seed(rank)
    images = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
    gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float().to(dev)

    def compute():
        for n in nets:
            for p in n.parameters():
                p.grad = None
        raw, vol = nets[1](nets[0](images))
        merged = nets[2](raw, vol)
        refined = nets[3](merged)
        total = bce(merged, gt) + bce(refined, gt)
        total.backward()
        return total.detach()

    def eager_step():
        total = compute()
        if reducer is not None:
            if use_graph:
                reducer.reduce_all()
            reducer.finish()
        return total

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eager_step()
    if use_graph:
        graphed = GraphedStep(compute, warmup=1, device=dev)      # captured with set_overlap(False): one stream

        def step():
            total = graphed()
            if reducer is not None:
                reducer.reduce_all()
                reducer.finish()
            return total
        step()                               # one untimed replay
    else:
        step = eager_step
    if reducer is not None:
        reducer.stats()                      # drop the warm-up samples

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step()
    host_dt = time.perf_counter() - t0          # host time to ENQUEUE the K steps (the GPU may still be running)
    barrier()
    dt = time.perf_counter() - t0
    dp_stats = reducer.stats() if reducer is not None else None
    if args.profile:
        if rank == 0:
            print(json.dumps({"metric": "views/sec (profiling run: no roofline / cpu legs)", "value": world * args.batch * args.views * args.steps / dt,
                              "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                              "steps_executed": args.steps + args.warmup + (2 if use_graph else 0)}))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- untimed instrumentation passes (every rank runs them: their gradient all-reduces are collectives) -------------
    # (1) the same step enqueued eagerly with HIP events around every engine / attention launch, streams overlapping as in the
    #     timed region: per-kernel durations (events cannot be recorded inside a graph replay);
    # (2) once more on ONE stream: durations undisturbed by whatever runs beside the kernel -> the roofline figures.
    names = set(ENGINE) | {"sv_window_attention_fwd", "sv_window_attention_bwd", "sv_swin_mlp_fwd", "sv_swin_mlp_bwd", "sv_swin_mlp_wgrad",
                               "sv_tconv4s2_fwd", "sv_swin_attn_block_fwd", "sv_swin_attn_block_bwd"}
    traced_steps = 2
    tracer = hip.Tracer(names) if rank == 0 else None
    eager_step()
    hip.TRACE = tracer
    for _ in range(traced_steps):
        eager_step()
    torch.cuda.synchronize()
    hip.TRACE = None
    iso = None
    if not args.no_isolated:
        S.set_overlap(False)
        iso = hip.Tracer(names) if rank == 0 else None
        eager_step()
        hip.TRACE = iso
        for _ in range(traced_steps):
            eager_step()
        torch.cuda.synchronize()
        hip.TRACE = None
        S.set_overlap(not args.no_overlap)
    ar_iso_ms = reducer.isolated_allreduce_ms() if reducer is not None else None

    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    assert bool(torch.isfinite(loss)), "non-finite loss"

    if rank == 0:
        views_total = world * B * V * args.steps
        value = views_total / dt
        summ = tracer.summary()
        isumm = iso.summary() if iso is not None else None

        def family(s):
            eng = [s[k] for k in ENGINE if k in s]
            return {k: sum(d[k] for d in eng) for k in ("ms", "flops", "bytes", "launches")}

        def rates(f):
            return {"avg_launch_us": f["ms"] * 1e3 / max(f["launches"], 1), "GB/s": f["bytes"] / (f["ms"] * 1e-3) / 1e9,
                    "TFLOP/s": f["flops"] / (f["ms"] * 1e-3) / 1e12, "ms_per_step": f["ms"] / traced_steps}

        ov, peak = family(summ), PEAK_TFLOPS[args.math]
        src = family(isumm) if isumm is not None else ov
        # dominant kernel family = the contraction engine.  Which roofline bounds it: algorithmic intensity against the ridge
        ai = src["flops"] / max(src["bytes"], 1.0)
        ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
        hbm_bound = ai < ridge
        r = rates(src)
        # HBM traffic per launch from the committed PMC passes (profiles/, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # runs of this command, corrected as MI355X_MICROARCH.md prescribes); null when none exists
        traffic, traffic_src = None, None
        try:
            import glob
            summ_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
            prof = json.load(open(summ_files[-1]))
            traffic_src = os.path.basename(summ_files[-1])
            if prof.get("kernel_source_hash") == hip.kernel_source_hash():
                pm = prof.get("pmc_traffic", {})
                traffic = [v["hbm_bytes_per_launch_corrected"] for k, v in pm.items() if k.startswith("contraction engine")][0]
            else:   # the committed counters describe another build of the kernels: do not quote them for this one
                traffic_src += f" (kernel sources changed since: profile {prof.get('kernel_source_hash')}, tree {hip.kernel_source_hash()})"
        except Exception:
            traffic = None
        out = {
            "metric": "views/sec fwd+bwd SwinVox-T 224^2, 32^3 voxel, n_views=8" if args.variant == "tiny" else
                      "views/sec fwd+bwd SwinVox-B (Swin-B encoder variant) 224^2, 32^3 voxel, n_views=8",
            "value": value, "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.math == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"SwinVox-{'T' if args.variant == 'tiny' else 'B'} full pipeline (ResNet50[:layer3] || Swin multi-stage, cross-view "
                                   f"attention, decoder, merger, refiner), fwd+2xBCE+bwd, train mode, B={B} samples x V={V} views of 224x224 per GPU",
                       "global_batch": world * B, "n_views": V, "images_per_gpu": B * V,
                       "parallelism": f"dp{world} (sample-sharded, RCCL gradient all-reduce)" if world > 1 else "single GPU",
                       "launch": "hipGraph replay of the step captured on one stream" if use_graph else "eager (one host launch per kernel, three streams)",
                       "math": (f"bf16 MFMA inputs, fp32 accumulate, {S.get_storage()} activations / fp32 weights, statistics and gradients of weights in HBM"
                                + ("; window-attention forward QK^T / PV on fp8 e4m3 MFMA operands (per-tile scales)" if args.fp8_attention else "")
                                if args.math == "bf16" else "exact fp32 MFMA, fp32 storage")},
            "roofline": {"bound": "hbm" if hbm_bound else "mfma",
                         "kernel": "implicit-GEMM contraction engine (igemm_kernel / gemm_dense_kernel / gemm_wide_kernel / conv_halo / conv3x3_halo_blocked / conv3x3_wgrad_halo kernels / wgrad_kernel / wgrad_wide_kernel: Linear/Conv/ConvTranspose fwd+dgrad+wgrad)",
                         "achieved": r["GB/s"] if hbm_bound else r["TFLOP/s"], "peak": PEAK_HBM_GBS if hbm_bound else peak,
                         "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": (r["GB/s"] / PEAK_HBM_GBS) if hbm_bound else (r["TFLOP/s"] / peak),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "measured_on": ("HIP events around every launch of the family in 2 eager steps on ONE stream right after the timed region "
                                         "(a graph replay admits no events; on three streams co-running kernels stretch each other: see `overlapped`)")
                                        if isumm is not None else "HIP events, 2 eager steps after the timed region (three streams)",
                         "algorithmic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
                         "mfma_tflops": r["TFLOP/s"], "mfma_frac": r["TFLOP/s"] / peak,
                         "algorithmic_bytes_per_launch": src["bytes"] / max(src["launches"], 1),
                         "algorithmic_flops_per_launch": src["flops"] / max(src["launches"], 1),
                         "launches_per_step": src["launches"] / traced_steps, "avg_launch_us": r["avg_launch_us"],
                         "overlapped": rates(ov)},
            "kernels": {k: {"launches_per_step": v["launches"] / traced_steps, "ms_per_step": v["ms"] / traced_steps,
                            "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None,
                            "GB/s": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 and v["bytes"] > 0 else None}
                        for k, v in (isumm if isumm is not None else summ).items()},
            # the other hand-written MFMA kernels of the Swin blocks against THEIR bound min(MFMA peak, algorithmic intensity x 8 TB/s)
            "roofline_kernels": {k: (lambda v: {"TFLOP/s": v["flops"] / (v["ms"] * 1e-3) / 1e12, "GB/s": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                                                "intensity_flop_per_byte": v["flops"] / v["bytes"],
                                                "bound_TFLOP/s": min(peak, v["flops"] / v["bytes"] * PEAK_HBM_GBS * 1e9 / 1e12),
                                                "frac_of_bound": (v["flops"] / (v["ms"] * 1e-3) / 1e12) / min(peak, v["flops"] / v["bytes"] * PEAK_HBM_GBS * 1e9 / 1e12)})(v)
                                 for k, v in (isumm if isumm is not None else summ).items()
                                 if k not in ENGINE and v["ms"] > 0 and v["flops"] > 0 and v["bytes"] > 0},
            "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
            "model_flops_tflops_per_gpu": (3 * FWD_GFLOP_PER_VIEW.get(V, 19.4) * 1e9 * value / world / 1e12) if args.variant == "tiny" else None,
            "data_parallel": {"world_size": (dist.get_world_size() if world > 1 else 1), "backend": (backend if world > 1 else None),
                              "payload_bytes": dp_stats["payload_bytes"] if dp_stats else 0, "buckets": dp_stats["buckets"] if dp_stats else 0,
                              "bucket_bytes": dp_stats["bucket_bytes"] if dp_stats else [], "param_checksum_spread": param_checksum_spread,
                              "allreduce_exposed_ms_per_step": (dp_stats["exposed_ms_per_step"] or 0.0) if dp_stats else 0.0,
                              "allreduce_isolated_ms": ar_iso_ms or 0.0,
                              "overlap": None if world == 1 else ("none: all buckets are reduced after the graph replay" if use_graph else
                                                                  ("buckets start inside the backward (module hooks + encoder groups)" if early else
                                                                   "buckets start inside the backward (module hooks)"))},
        }
        if args.math == "bf16" and S.get_storage() == "bf16" and args.variant == "tiny" and not args.no_parity:
            out["iou_delta_vs_oracle"] = bf16_parity_report(dev)
        if args.detail:
            rows = sorted((iso if iso is not None else tracer).detail().items(), key=lambda kv: -kv[1][1])
            for (name, tag), (cnt, ms, fl, by) in rows[:70]:
                print(f"{ms / traced_steps:8.3f} ms/step  x{cnt / traced_steps:5.1f}  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {by / max(ms, 1e-9) / 1e6:7.0f} GB/s  {name:18s} {tag}",
                      file=sys.stderr)
        if not args.no_cpu_baseline:
            # host share of a one-GPU box is 16 cores (os.cpu_count() reports the whole node): cap the thread pool there
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            out["cpu_baseline"] = cpu_baseline(max(1, min(ncpu, 16)))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
