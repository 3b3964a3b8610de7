"""Per-launch timing (HIP events on the launch stream, one stream) of the contraction-engine calls of ONE module's forward + backward.
python scripts/detail_module.py {encoder,decoder,merger,refiner}   (SV_B=<samples>)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16"); S.set_overlap(False)
which = sys.argv[1] if len(sys.argv) > 1 else "refiner"
B, V = int(os.environ.get("SV_B", "64")), 8
cfg = S.default_cfg()
rg = dict(device=dev, requires_grad=True)
m, ins = {"encoder": (Encoder, [(B, V, 3, 224, 224)]), "decoder": (Decoder, [(B, V, 256, 7, 7)]),
          "merger": (Merger, [(B, V, 9, 32, 32, 32), (B, V, 32, 32, 32)]), "refiner": (Refiner, [(B, 32, 32, 32)])}[which]
m = m(cfg).to(dev).train()
ins = [torch.randn(*s, **rg) for s in ins]
def step():
    m.zero_grad(set_to_none=True)
    out = m(*ins)
    out = out if isinstance(out, torch.Tensor) else sum(o.float().sum() for o in out)
    out.sum().backward()
for _ in range(3): step()
names = {"sv_conv_gather", "sv_tconv_gather", "sv_conv_wgrad", "sv_stencil3_fwd", "sv_stencil3_wgrad", "sv_tconv4s2_fwd"}
tr = hip.Tracer(names); hip.TRACE = tr
for _ in range(3): step()
torch.cuda.synchronize(); hip.TRACE = None
tr.summary()
for (name, tag), (cnt, ms, fl, by) in sorted(tr.detail().items(), key=lambda kv: -kv[1][1]):
    print(f"{ms / 3:8.3f} ms/step  x{cnt / 3:4.1f}  {fl / max(ms, 1e-9) / 1e9:7.1f} TF/s  {name:18s} {tag}")
