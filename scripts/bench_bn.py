#!/usr/bin/env python3
"""BatchNorm passes at the ResNet shapes of the bench (512 images, bf16): achieved HBM rate of the forward apply (+ residual) and the
backward (reduce + fold + apply), per layer shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.hip import call, ptr
from swinvox_amd.ops import ACT_RELU, BatchNormState
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
I = int(os.environ.get("SV_I", "512"))
for H, C, res in ((56, 64, False), (56, 256, True), (28, 128, False), (28, 512, True), (14, 256, False), (14, 1024, True)):
    M = I * H * H
    y = torch.randn(M, C, device=dev).bfloat16()
    r = torch.randn(M, C, device=dev).bfloat16() if res else None
    dz = torch.randn(M, C, device=dev).bfloat16()
    bn = torch.nn.BatchNorm2d(C).to(dev)
    st = BatchNormState(bn, M, True)
    call("sv_bn_stats", ptr(y), M, C, C, ptr(st.sums)); st.finalize()
    z, dy = ops.empty(M, C, like=y), ops.empty(M, C, like=y)
    dres = ops.empty(M, C, like=y) if res else None
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    unit = M * C * 2 / 1e6
    tf = timeit(lambda: st.apply(y, C, z, C, ACT_RELU, 0.0, r, C if res else 0))
    tb = timeit(lambda: st.backward(dz, C, z if res else None, C, y, C, dy, C, dg, db, ACT_RELU, 0.0, dres, C if res else 0))
    nf = 3 if res else 2
    nb = 6 if res else 5
    print(f"{H:3d}^2 x {C:4d}{' +res' if res else '     '}: tensor {unit:6.0f} MB   apply {tf:6.0f} us ({nf * unit / tf * 1e3 / 1e3:5.2f} TB/s, {nf} passes)   "
          f"backward {tb:6.0f} us ({nb * unit / tb * 1e3 / 1e3:5.2f} TB/s, {nb} passes)")
