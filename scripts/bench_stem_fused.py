#!/usr/bin/env python3
"""ResNet stem passes around the max-pool, fused (sv_bn_act_maxpool_fwd / sv_bn_maxpool_bwd) against separate, 512 images, bf16."""
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
N, H, C = int(os.environ.get("SV_I", "512")), 112, 64
M, Mo = N * H * H, N * 56 * 56
y = torch.randn(M, C, device=dev).bfloat16()
dmp = torch.randn(Mo, C, device=dev).bfloat16()
bn = torch.nn.BatchNorm2d(C).to(dev)
st = BatchNormState(bn, M, True)
call("sv_bn_stats", ptr(y), M, C, C, ptr(st.sums)); st.finalize()
z, mp, dz, dy = ops.empty(M, C, like=y), ops.empty(Mo, C, like=y), ops.empty(M, C, like=y), ops.empty(M, C, like=y)
idx = torch.empty(Mo * C, dtype=torch.uint8, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
unit = M * C * 2 / 1e6
t_apply = timeit(lambda: st.apply(y, C, z, C, ACT_RELU, 0.0))
t_pool = timeit(lambda: call("sv_maxpool2d_fwd", ptr(z), ptr(mp), ptr(idx), N, H, H, C))
t_ffwd = timeit(lambda: call("sv_bn_act_maxpool_fwd", ptr(y), ptr(st.scale), ptr(st.shift), ptr(mp), ptr(idx), N, H, H, C, ACT_RELU, 0.0))
print(f"forward : apply {t_apply:6.0f} + pool {t_pool:6.0f} = {t_apply + t_pool:6.0f} us   fused {t_ffwd:6.0f} us ({(unit * 1.375) / t_ffwd * 1e3:5.0f} GB/s)")
t_pb = timeit(lambda: call("sv_maxpool2d_bwd", ptr(dmp), ptr(idx), ptr(dz), N, H, H, C))
t_bb = timeit(lambda: st.backward(dz, C, None, C, y, C, dy, C, dg, db, ACT_RELU, 0.0))
def fb():
    ws = ops.zeros_f64((ops.BN_BWD_SLOTS + 1) * 2 * C + 2, dev)
    call("sv_bn_maxpool_bwd", ptr(dmp), ptr(idx), ptr(y), ptr(bn.weight), ptr(st.mean), ptr(st.rstd), ptr(st.scale), ptr(st.shift), N, H, H, C,
         ACT_RELU, 0.0, 1, ptr(dy), ptr(dg), ptr(db), ptr(ws))
t_fb = timeit(fb)
print(f"backward: pool {t_pb:6.0f} + BatchNorm {t_bb:6.0f} = {t_pb + t_bb:6.0f} us   fused {t_fb:6.0f} us ({(unit * 3.75) / t_fb * 1e3:5.0f} GB/s)")
