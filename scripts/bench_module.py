"""One module's forward + backward at the default bench shape (B=32, V=8), for kernel traces of a single module.
python scripts/bench_module.py {encoder,decoder,merger,refiner}   (SV_LIB=<path> for A/B builds, SV_B=<samples>)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip
if os.environ.get("SV_LIB"):
    hip.LIB_PATH = os.environ["SV_LIB"]
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
which = sys.argv[1] if len(sys.argv) > 1 else "merger"
B, V = int(os.environ.get("SV_B", "32")), 8
cfg = S.default_cfg()
rg = dict(device=dev, requires_grad=True)
if which == "encoder":
    m, ins = Encoder(cfg), [torch.randn(B, V, 3, 224, 224, **rg)]
elif which == "decoder":
    m, ins = Decoder(cfg), [torch.randn(B, V, 256, 7, 7, **rg)]
elif which == "merger":
    m, ins = Merger(cfg), [torch.randn(B, V, 9, 32, 32, 32, **rg), torch.randn(B, V, 32, 32, 32, **rg)]
else:
    m, ins = Refiner(cfg), [torch.randn(B, 32, 32, 32, **rg)]
m = m.to(dev).train()
def step():
    m.zero_grad(set_to_none=True)
    out = m(*ins)
    out = out if isinstance(out, torch.Tensor) else sum(o.sum() for o in out)
    out.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize()
print(f"{which} fwd+bwd B={B} V={V}: {e0.elapsed_time(e1) / 10:.3f} ms")
