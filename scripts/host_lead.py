"""How far the host runs ahead of the GPU at the end of every step (and after the forward): lead = time the GPU finishes the
work minus time the host finished enqueuing it.  A lead near zero means the GPU waits for launches (host-bound)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner
dev = torch.device("cuda", 0)
S.set_math("bf16"); S.set_storage("bf16")
cfg = S.default_cfg()
B, V = int(os.environ.get("SV_B", "32")), 8
nets = [m(cfg).to(dev).train() for m in (Encoder, Decoder, Merger, Refiner)]
x = (0.5 * torch.randn(B, V, 3, 224, 224, device=dev)).clamp(-1, 1)
gt = (torch.rand(B, 32, 32, 32, device=dev) < 0.1).float()
bce = torch.nn.functional.binary_cross_entropy_with_logits
def fwd():
    raw, vol = nets[1](nets[0](x)); m = nets[2](raw, vol); r = nets[3](m)
    return bce(m, gt) + bce(r, gt)
for _ in range(3):
    for n in nets: n.zero_grad(set_to_none=True)
    fwd().backward()
torch.cuda.synchronize()
ev0 = torch.cuda.Event(enable_timing=True); ev0.record(); torch.cuda.synchronize(); h0 = time.perf_counter()
rec = []
for i in range(8):
    for n in nets: n.zero_grad(set_to_none=True)
    loss = fwd()
    ef = torch.cuda.Event(enable_timing=True); ef.record(); hf = time.perf_counter()
    loss.backward()
    eb = torch.cuda.Event(enable_timing=True); eb.record(); hb = time.perf_counter()
    rec.append((ef, hf, eb, hb))
torch.cuda.synchronize()
for i, (ef, hf, eb, hb) in enumerate(rec):
    gf, gb = ev0.elapsed_time(ef), ev0.elapsed_time(eb)
    print(f"step {i}: host fwd done {1e3*(hf-h0):7.1f} ms, GPU fwd done {gf:7.1f} (lead {gf-1e3*(hf-h0):6.1f}) | host bwd done {1e3*(hb-h0):7.1f}, GPU bwd done {gb:7.1f} (lead {gb-1e3*(hb-h0):6.1f})")
