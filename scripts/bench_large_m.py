import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import swinvox_amd as S
from swinvox_amd import hip
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
for M, K, N in ((802816, 96, 384), (802816, 384, 96), (802816, 96, 288), (802816, 64, 256), (802816, 256, 64), (200704, 192, 768), (200704, 768, 192), (50176, 384, 1536), (50176, 1536, 384)):
    sp = ConvSpec.linear(K, N)
    x = torch.randn(M, K, device=dev).bfloat16(); dy = torch.randn(M, N, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev) * 0.05; wf, wd = sp.pack_fwd(w), sp.pack_dgrad(w)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16); dw = torch.zeros_like(w)
    by = (M * K + M * N) * 2
    tf = timeit(lambda: sp.forward(x, M, (1, 1, 1), wf, y)); td = timeit(lambda: sp.dgrad(dy, M, (1, 1, 1), wd, dx)); tw = timeit(lambda: sp.wgrad(dy, x, M, (1, 1, 1), dw))
    print(f"M={M} {K}->{N}: fwd {tf:7.1f} us {by/tf/1e3:6.0f} GB/s | dgrad {td:7.1f} us {by/td/1e3:6.0f} GB/s | wgrad {tw:7.1f} us {by/tw/1e3:6.0f} GB/s  ({2.0*M*K*N/tw/1e6:5.0f} TF/s)")
