#!/usr/bin/env python3
"""Turn rocprofv3 CSV output under gpurun_out/ into the small, tracked summaries under profiles/.

  python scripts/summarize_profile.py --stats gpurun_out/prof_X --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write \
         --tag r1 --steps-in-trace 4 --steps-in-pmc 2
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name: str) -> str:
    n = name.split("(")[0]
    if "swin_mlp" in n:
        for k in ("swin_mlp_fwd_kernel", "swin_mlp_bwd_kernel", "swin_mlp_wgrad_kernel", "swin_mlp_pack_kernel"):
            if k in n:
                return "fused Swin MLP: " + k
    if ("igemm_kernel" in n or ("wgrad_kernel" in n and "stencil" not in n) or "gemm_dense_kernel" in n or "gemm_wide_kernel" in n or "wgrad_wide_kernel" in n
            or "conv_halo_kernel" in n or "conv3x3_halo_blocked_kernel" in n or "conv3x3_wgrad_halo_kernel" in n):
        # (the halo-tile kernels run behind the same entry points: sv_conv_gather / sv_tconv_gather / sv_conv_wgrad)
        return "contraction engine (igemm_kernel + gemm_dense_kernel + gemm_wide_kernel + halo-tile kernels + wgrad_kernel + wgrad_wide_kernel)"
    if "swin_attn_block_fwd" in n:
        return "fused Swin attention branch forward (swin_attn_block_fwd_kernel)"
    if "swin_attn_block_bwd" in n:
        return "fused Swin attention branch backward (swin_attn_block_bwd_kernel)"
    if "win_attn_fwd" in n:
        return "window attention forward (win_attn_fwd_*)"
    if "win_attn_bwd" in n:
        return "window attention backward (win_attn_bwd_*)"
    if "bn_bwd" in n or "scale_shift_act" in n or "bn_finalize" in n or "bn_stats" in n or "bn_act_maxpool" in n or "bn_pool" in n:
        return "BatchNorm passes (bn_bwd_* + scale_shift_act_* + bn_finalize + bn_stats + the passes fused with a max-pool: bn_act_maxpool*, bn_pool*_bwd_*)"
    if "ln_fwd" in n or "ln_bwd" in n or "lnl_" in n:
        return "LayerNorm passes (ln_* + lnl_*)"
    if "stencil3" in n:
        return "merger stencils (stencil3_*)"
    if "tconv4s2" in n:
        return "decoder layer4 halo transposed convolution (tconv4s2_fwd_kernel)"
    return n.replace("void ", "").strip()[-60:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--mfma", help="PMC pass with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE")
    ap.add_argument("--tag", default="r1")
    ap.add_argument("--steps-in-trace", type=int, default=4)
    ap.add_argument("--steps-in-pmc", type=int, default=2)
    a = ap.parse_args()
    import sys
    sys.path.insert(0, ROOT)
    from swinvox_amd.hip import kernel_source_hash
    out = {"tag": a.tag, "kernel_source_hash": kernel_source_hash()}    # the build the counters describe (bench.py checks it)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    if a.stats:
        f = max(glob.glob(os.path.join(a.stats, "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # gpurun_out/ is merged across calls: the newest run
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        fam = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            k = family(r["Name"])
            fam[k][0] += int(r["Calls"])
            fam[k][1] += float(r["TotalDurationNs"])
        out["kernel_time"] = {"gpu_ms_per_step": tot / 1e6 / a.steps_in_trace, "families": {
            k: {"launches_per_step": v[0] / a.steps_in_trace, "ms_per_step": v[1] / 1e6 / a.steps_in_trace,
                "avg_launch_us": v[1] / 1e3 / v[0], "share": v[1] / tot}
            for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:20]}}
    if a.fetch and a.write:
        def load(d):
            f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                k = family(r["Kernel_Name"])
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
            return agg
        fe, wr = load(a.fetch), load(a.write)
        pm = {}
        for k in sorted(fe, key=lambda k: -fe[k][1])[:14]:
            n = fe[k][0]
            pm[k] = {"launches_per_step": n / a.steps_in_pmc,
                     "FETCH_SIZE_GB_per_step_raw": fe[k][1] / 1024 ** 2 / a.steps_in_pmc,
                     "FETCH_GB_per_step_x2_gfx950_wide_read_correction": 2 * fe[k][1] / 1024 ** 2 / a.steps_in_pmc,
                     "WRITE_SIZE_GB_per_step": wr.get(k, [0, 0.0])[1] / 1024 ** 2 / a.steps_in_pmc,
                     "hbm_bytes_per_launch_corrected": (2 * fe[k][1] + wr.get(k, [0, 0.0])[1]) * 1024 / n}
        out["pmc_traffic"] = pm
    if a.mfma:
        # MFMA utilisation per kernel family: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs: 32 per v_mfma_f32_32x32x16_bf16,
        # 16 per 16x16x32 - MI355X_MICROARCH.md cycle constants) against the SIMD-cycles the dispatch had: GRBM_GUI_ACTIVE (summed over the
        # 8 XCDs) / 8 x 1024 SIMDs.  Calibration: the fused MLP forward issues a known number of MFMAs (see profiles/README.md).
        f = max(glob.glob(os.path.join(a.mfma, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = family(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
        mf = {}
        for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))[:14]:
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            simd_cycles = gui / 8.0 * 1024.0
            mf[k] = {"launches_per_step": cnt[k] / a.steps_in_pmc,
                     "SQ_VALU_MFMA_BUSY_CYCLES_per_step": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / a.steps_in_pmc,
                     "SQ_BUSY_CYCLES_per_step": c.get("SQ_BUSY_CYCLES", 0.0) / a.steps_in_pmc,
                     "SQ_WAVE_CYCLES_per_step": c.get("SQ_WAVE_CYCLES", 0.0) / a.steps_in_pmc,
                     "GRBM_GUI_ACTIVE_per_step": gui / a.steps_in_pmc,
                     "mfma_busy_fraction_of_simd_cycles": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles) if simd_cycles else None}
        out["mfma_utilisation"] = mf
    p = os.path.join(ROOT, "profiles", f"{a.tag}_summary.json")
    json.dump(out, open(p, "w"), indent=1)
    print("wrote", p)


if __name__ == "__main__":
    main()
