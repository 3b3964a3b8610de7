"""Fused Swin MLP kernels at the benchmark shapes (I = 256 images): ms per launch, algorithmic TFLOP/s and GB/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swinvox_amd import hip
from swinvox_amd.hip import call, ptr

dev = torch.device("cuda:0")
I = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for C, res in ((96, 56), (192, 28), (128, 56)):
    M = I * res * res
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    w1 = torch.randn(4 * C, C, device=dev) / C ** 0.5
    w2 = torch.randn(C, 4 * C, device=dev) / (4 * C) ** 0.5
    b1, b2, lg, lb = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    packs = torch.empty(16 * C * C, dtype=torch.bfloat16, device=dev)
    call("sv_swin_mlp_pack", ptr(w1), ptr(w2), ptr(packs), C)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dw1, db1, dw2, db2 = torch.zeros(4 * C, C, device=dev), torch.zeros(4 * C, device=dev), torch.zeros(C, 4 * C, device=dev), torch.zeros(C, device=dev)
    w1r, w2tr = w1.to(torch.bfloat16).contiguous(), w2.t().to(torch.bfloat16).contiguous()
    unit = 2.0 * M * C * 4 * C
    f = timeit(lambda: call("sv_swin_mlp_fwd", ptr(x), ptr(out), ptr(lg), ptr(lb), ptr(packs), ptr(b1), ptr(b2), None, res * res, M, C, 1e-5))
    b = timeit(lambda: call("sv_swin_mlp_bwd", ptr(x), ptr(dy), ptr(out), ptr(lg), ptr(lb), ptr(packs), ptr(b1), None, res * res, ptr(dg), ptr(db), M, C, 1e-5))
    w = timeit(lambda: call("sv_swin_mlp_wgrad", ptr(x), ptr(dy), ptr(lg), ptr(lb), ptr(w1r), ptr(w2tr), ptr(b1), None, res * res, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), M, C, 1e-5))
    print(f"C={C:3d} M={M:7d}  fwd {f:6.3f} ms {2 * unit / f / 1e9:6.0f} TF/s {4.0 * M * C / f / 1e6:6.0f} GB/s | "
          f"bwd {b:6.3f} ms {3 * unit / b / 1e9:6.0f} TF/s | wgrad {w:6.3f} ms {4 * unit / w / 1e9:6.0f} TF/s | sum {f + b + w:6.3f} ms", flush=True)
