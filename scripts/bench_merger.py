"""Merger forward + backward (the LDS-halo stencil kernels) at the default bench shape.  SV_LIB=<path> for A/B builds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip
if os.environ.get("SV_LIB"):
    hip.LIB_PATH = os.environ["SV_LIB"]
from swinvox_amd.models import Merger
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
B, V = int(os.environ.get("SV_B", "32")), 8
m = Merger(S.default_cfg()).to(dev).train()
raw = torch.randn(B, V, 9, 32, 32, 32, device=dev, requires_grad=True)
vol = torch.randn(B, V, 32, 32, 32, device=dev, requires_grad=True)
def step():
    m.zero_grad(set_to_none=True)
    out = m(raw, vol)
    out.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(f"merger fwd+bwd B={B} V={V}: {e0.elapsed_time(e1) / 10:.3f} ms")
