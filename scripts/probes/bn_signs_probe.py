"""Minimal reproduction harness for the BatchNorm sign-word path: one launch at a time, synchronised, with progress lines."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swinvox_amd import hip, ops
from swinvox_amd.hip import call, ptr
dev = torch.device("cuda", 0)
M, C = 1031, 256
x = torch.randn(M, C, device=dev); res = torch.randn(M, C, device=dev); y = torch.empty(M, C, device=dev)
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
signs = torch.zeros(M * (C // 64), dtype=torch.int64, device=dev)
print("plain", flush=True)
call("sv_scale_shift_act", ptr(x), C, ptr(sc), ptr(sh), ptr(res), C, ptr(y), C, M, C, 1, 0.0)
torch.cuda.synchronize(); print("plain ok", flush=True)
call("sv_scale_shift_act_signs", ptr(x), C, ptr(sc), ptr(sh), ptr(res), C, ptr(y), C, M, C, 1, 0.0, ptr(signs))
torch.cuda.synchronize(); print("signs ok", flush=True)
ref = (x + res) > 0
w = signs.view(M, C // 256, 4)
bits = torch.stack([(w[:, :, j:j + 1] >> torch.arange(64, device=dev)) & 1 for j in range(4)], -1)   # [M, C/256, 64 lanes, 4 j]
got = bits.reshape(M, C).bool()
print("mask equal:", bool((got == ref).all()), flush=True)

# the op test's sequence, one synchronised step at a time
from swinvox_amd.ops import ACT_RELU
bn = torch.nn.BatchNorm1d(C).to(dev)
st = ops.BatchNormState(bn, M, True)
call("sv_bn_stats", ptr(x), M, C, C, ptr(st.sums)); torch.cuda.synchronize(); print("bn_stats ok", flush=True)
st.finalize(); torch.cuda.synchronize(); print("finalize ok", flush=True)
z = ops.empty(M, C, device=dev)
st.apply(x, C, z, C, ACT_RELU, 0.2, res, C); torch.cuda.synchronize(); print("apply ok, signs:", st.signs is not None, flush=True)
dz = torch.randn(M, C, device=dev)
dx, dres = ops.empty(M, C, device=dev), ops.empty(M, C, device=dev)
dg, db = ops.zeros(C, device=dev), ops.zeros(C, device=dev)
st.backward(dz, C, z, C, x, C, dx, C, dg, db, ACT_RELU, 0.2, dres, C); torch.cuda.synchronize(); print("backward ok", flush=True)
print("dres equal:", bool(torch.equal(dres, dz * (z > 0))), flush=True)
