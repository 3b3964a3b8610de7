import os, sys, torch
sys.path.insert(0, "/root/repo")
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M, K, N in ((200704, 64, 256), (200704, 256, 64), (200704, 64, 64), (50176, 128, 512), (12544, 256, 1024), (401408, 64, 256)):
    sp = ConvSpec.linear(K, N)
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev) * 0.05
    wf = sp.pack_fwd(w); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    stats = torch.zeros(ops.BN_SLOTS * 2 * N, dtype=torch.float64, device=dev)
    t0 = timeit(lambda: sp.forward(x, M, (1, 1, 1), wf, y))
    t1 = timeit(lambda: sp.forward(x, M, (1, 1, 1), wf, y, stats=stats))
    print(f"M={M} {K}->{N}: plain {t0:.1f} us, with BN stats {t1:.1f} us")
