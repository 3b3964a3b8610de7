#!/usr/bin/env python3
"""Micro-benchmark of the BatchNorm / LayerNorm kernels on bench shapes (bf16 storage): time and effective GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.hip import call, ptr
from swinvox_amd.ops import ACT_RELU, BatchNormState
if os.environ.get("SV_LIB"):
    hip.LIB_PATH = os.environ["SV_LIB"]
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
SC = int(os.environ.get("SV_NORM_SCALE", "1"))   # 4 = the shapes of the default bench (256 images)
for M, C in ((802816 * SC, 64), (200704 * SC, 256), (200704 * SC, 64), (50176 * SC, 512), (50176 * SC, 128), (12544 * SC, 1024), (12544 * SC, 256)):
    bn = torch.nn.BatchNorm1d(C).to(dev)
    x = torch.randn(M, C, device=dev).bfloat16(); dz = torch.randn(M, C, device=dev).bfloat16()
    z = torch.empty_like(x); dx = torch.empty_like(x); res = torch.randn(M, C, device=dev).bfloat16()
    st = BatchNormState(bn, M, True)
    call("sv_bn_stats", ptr(x), M, C, C, ptr(st.sums)); st.finalize()
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    mb = M * C * 2 / 1e6
    t1 = timeit(lambda: st.apply(x, C, z, C, ACT_RELU, 0.0))
    t2 = timeit(lambda: st.apply(x, C, z, C, ACT_RELU, 0.0, res, C))
    t3 = timeit(lambda: st.backward(dz, C, z, C, x, C, dx, C, dg, db, ACT_RELU, 0.0))
    print(f"BN M={M:7d} C={C:5d} ({mb:6.1f} MB/tensor)  apply {t1:7.1f} us {2*mb/t1*1e3:6.0f} GB/s | apply+res {t2:7.1f} us {3*mb/t2*1e3:6.0f} GB/s | bwd(reduce+apply) {t3:7.1f} us {7*mb/t3*1e3:6.0f} GB/s")
for rows, C in ((200704 * SC, 96), (50176 * SC, 192), (12544 * SC, 384), (3136 * SC, 768)):
    x = torch.randn(rows, C, device=dev).bfloat16(); dy = torch.randn(rows, C, device=dev).bfloat16()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dx = torch.empty_like(x); dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, rows, C)
    mb = rows * C * 2 / 1e6
    t1 = timeit(lambda: ops.layernorm_fwd(x, g, b, rows, C))
    t2 = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, rows, C))
    print(f"LN rows={rows:7d} C={C:5d} ({mb:6.1f} MB/tensor)  fwd {t1:7.1f} us {2*mb/t1*1e3:6.0f} GB/s | bwd {t2:7.1f} us {3*mb/t2*1e3:6.0f} GB/s")
