#!/usr/bin/env python3
"""Cycles per phase of the halo-tile convolution (workgroup 0, thread 0), from a -DSV_HC_PROFILE build of conv_halo.hip:
   make -C swinvox_amd/csrc clean && make -C swinvox_amd/csrc FLAGS_conv_halo='-mllvm -amdgpu-atomic-optimizer-strategy=None -DSV_HC_PROFILE'"""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); lib = hip.load(); S.set_math("bf16"); S.set_storage("bf16"); ops.set_conv_halo(2)
STEM = bool(os.environ.get("SV_STEM"))
n, H = 512, (112 if STEM else 56)
sp = ConvSpec.conv2d(16, 64, 4, 1, 2, og_fixed=(1, 112, 112)) if STEM else ConvSpec.conv2d(64, 64, 3, 1, 1)
M = n * H * H
x = torch.randn(M, 16 if STEM else 64, device=dev).bfloat16()
w = torch.randn(64, 16, 4, 4, device=dev) / 16.0 if STEM else torch.randn(64, 64, 3, 3, device=dev) / 24.0
wf = ops.pack_one(sp, w, "f")
out = ops.empty(M, 64, device=dev)
stats = torch.zeros(ops.BN_SLOTS, 128, dtype=torch.float64, device=dev)
buf = (C.c_longlong * 8)()
lib.sv_conv_halo_prof.argtypes = [C.POINTER(C.c_longlong), C.c_int]
for it in range(3):
    sp.forward(x, n, (1, H, H), wf, out, stats=stats)
    torch.cuda.synchronize()
    lib.sv_conv_halo_prof(buf, 1)
    v = list(buf)
    t = 98 if STEM else 28
    print(f"run {it}: loop {v[5]} cycles ({v[5] / t:.0f} per tile)  preamble {v[0]}  after the loop {v[1]}  per tile: contraction {v[2] / t:.0f}  patch store {v[3] / t:.0f}  epilogue {v[4] / t:.0f}  (clock64 ticks)   "
          f"workgroup life {v[6] / 100:.1f} us = {v[7]} cycles -> {v[7] / max(v[6], 1) / 10:.2f} GHz")
wg = (C.c_longlong * 2048)()
lib.sv_conv_halo_prof_wg.argtypes = [C.POINTER(C.c_longlong)]
lib.sv_conv_halo_prof_wg(wg)
import numpy as np
a = np.array(list(wg), dtype=np.int64).reshape(512, 4)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
st, lp, en, tl = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0, a[:, 3]
print(f"last launch, all workgroups (us from the first start): start {st.min():.1f}..{st.max():.1f}  loop start {lp.min():.1f}..{lp.max():.1f}  end {en.min():.1f}..{en.max():.1f}"
      f"  tiles {tl.min()}..{tl.max()} (sum {tl.sum()})  life {np.percentile(en - st, [0, 50, 100])}")
for x in range(8):
    m = np.arange(len(a)) % 8 == x
    print(f"  xcd {x}: start {st[m].min():.1f}..{st[m].max():.1f}  end {en[m].min():.1f}..{en[m].max():.1f}  tiles {tl[m].min()}..{tl[m].max()} sum {tl[m].sum()}")
