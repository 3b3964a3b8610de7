"""Per-part GPU time of one training step (cuda events around sub-chains)."""
import sys, time, torch
sys.path.insert(0, '.')
import oracle as O, swinvox_amd as S
from swinvox_amd import ops, hip
from swinvox_amd.models import Encoder, Decoder, Merger, Refiner
from swinvox_amd.models import swin_transformer as ST
dev = torch.device('cuda:0')
S.set_math("bf16")
cfg = S.default_cfg()
nets = [Encoder(cfg), Decoder(cfg), Merger(cfg), Refiner(cfg)]
for n in nets:
    n.apply(O.init_weights); n.to(dev).train()
B, V = 8, 8
g = torch.Generator().manual_seed(0)
images = (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1).to(dev)
gt = (torch.rand(B, 32, 32, 32, generator=g) < 0.1).float().to(dev)
bce = torch.nn.functional.binary_cross_entropy_with_logits
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
# wrap sub-chains of the encoder
orig_swin_fwd, orig_swin_bwd = ST.swin_forward, ST.swin_backward
import swinvox_amd.models.encoder as ENC
def sf(*a, **k):
    mark("enc.fwd.resnet(+stem,reduce,pool) done"); r = orig_swin_fwd(*a, **k); mark("enc.fwd.swin done"); return r
def sb(*a, **k):
    mark("enc.bwd.post+cva+neck done"); r = orig_swin_bwd(*a, **k); mark("enc.bwd.swin done"); return r
ENC.swin_forward, ENC.swin_backward = sf, sb
def step():
    for n in nets:
        for p in n.parameters(): p.grad = None
    mark("start")
    f = nets[0](images); mark("enc.fwd.neck+cva+post done")
    raw, vol = nets[1](f); mark("dec.fwd")
    m = nets[2](raw, vol); mark("mer.fwd")
    r = nets[3](m); mark("ref.fwd")
    total = bce(m, gt) + bce(r, gt); mark("loss")
    total.backward(); mark("enc.bwd.resnet done (end)")
for _ in range(2): marks.clear(); step()
marks.clear(); step(); torch.cuda.synchronize()
# backward marks for tail: add hooks via timing of autograd is implicit; report consecutive deltas
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    print(f"{n1:42s} {e0.elapsed_time(e1):8.2f} ms")
print("total", marks[0][1].elapsed_time(marks[-1][1]))
