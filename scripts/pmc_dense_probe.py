import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import swinvox_amd as S
from swinvox_amd import hip
from swinvox_amd.ops import ConvSpec
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
M, N = 200704, 256
for K in (64, 512):
    sp = ConvSpec.linear(K, N)
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev) * 0.05
    wf = sp.pack_fwd(w); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): sp.forward(x, M, (1, 1, 1), wf, y)
torch.cuda.synchronize()
