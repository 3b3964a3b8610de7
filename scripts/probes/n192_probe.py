import math, os, sys, torch
sys.path.insert(0, '/root/repo')
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("SV_GEMM_WIDE =", os.environ.get("SV_GEMM_WIDE", "1"))
for M, K, N in ((401408, 768, 192), (401408, 192, 192), (401408, 192, 768), (401408, 192, 576), (100352, 384, 384), (100352, 1536, 384)):
    sp = ConvSpec.linear(K, N)
    x = torch.randn(M, K, device=dev).bfloat16()
    r = torch.randn(M, N, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev) / math.sqrt(K)
    b = torch.zeros(N, device=dev)
    wf = ops.pack_one(sp, w, "f")
    out = ops.empty(M, N, device=dev)
    t0 = timeit(lambda: sp.forward(x, M, (1, 1, 1), wf, out, bias=b))
    t1 = timeit(lambda: sp.forward(x, M, (1, 1, 1), wf, out, bias=b, residual=r, ldr=N))
    fl = 2.0 * M * K * N
    print(f"{M} {K:5d} -> {N:4d}: bias {t0:6.0f} us ({fl / t0 / 1e6:5.0f} TF/s)   bias + residual {t1:6.0f} us ({fl / t1 / 1e6:5.0f} TF/s)")
