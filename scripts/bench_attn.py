#!/usr/bin/env python3
"""Micro-benchmark of the window-attention kernels per Swin stage (B*V = 64 images, bf16 storage)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip
from swinvox_amd.hip import call, ptr
if os.environ.get("SV_LIB"):
    hip.LIB_PATH = os.environ["SV_LIB"]      # A/B builds of the library
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
I = int(os.environ.get("SV_ATTN_I", "64"))
for H, heads in ((56, 3), (28, 6), (14, 12), (7, 24)):
    C = heads * 32
    rows = I * H * H
    qkv = torch.randn(rows, 3 * C, device=dev).bfloat16(); dout = torch.randn(rows, C, device=dev).bfloat16()
    table = torch.randn(169, heads, device=dev) * 0.1
    out = torch.empty(rows, C, device=dev, dtype=torch.bfloat16); dqkv = torch.empty_like(qkv); dt = torch.zeros(169, heads, device=dev); ws = torch.zeros(16 * 169 * heads, device=dev)
    for shift in ((0, 3) if H > 7 else (0,)):
        tf = timeit(lambda: call("sv_window_attention_fwd", ptr(qkv), ptr(table), ptr(out), I, H, H, C, heads, shift, hip.MATH_BF16))
        tb = timeit(lambda: call("sv_window_attention_bwd", ptr(qkv), ptr(table), ptr(dout), ptr(dqkv), ptr(dt), ptr(ws), I, H, H, C, heads, shift, hip.MATH_BF16))
        mb = rows * C * 2 / 1e6
        fl = 4.0 * rows * 49 * C   # QK^T + AV, 49-token algorithmic count
        print(f"H={H:3d} heads={heads:2d} shift={shift}  fwd {tf:7.1f} us ({4*mb/tf*1e3:6.0f} GB/s, {fl/tf/1e6:6.1f} TF/s)   bwd {tb:7.1f} us ({7*mb/tb*1e3:6.0f} GB/s, {2.5*fl/tb/1e6:6.1f} TF/s)")
