import copy, sys, torch
sys.path.insert(0, '.')
import oracle as O, swinvox_amd as S
from swinvox_amd import ops
from swinvox_amd.models import Encoder, Decoder, Merger, Refiner
dev = torch.device('cuda:0')
torch.manual_seed(0)
def synth_images(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)
cfg = O.default_cfg()
onets = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
for i, n in enumerate(onets): O.seeded_weights_(n, seed=100 + i)
O.calibrate_(onets, synth_images(2, 2, 1234))
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12)), float((a - b).abs().mean() / (b.abs().mean() + 1e-12))
for mode in ("eval", "train"):
    for n in onets:
        n.train(mode == "train")
        for m in n.modules():
            if isinstance(m, torch.nn.Dropout): m.p = 0.0
            if isinstance(m, O.model.SwinBlock): m.dp = 0.0
    pn = [Encoder(S.default_cfg()), Decoder(S.default_cfg()), Merger(S.default_cfg()), Refiner(S.default_cfg())]
    for p, o in zip(pn, onets):
        p.load_state_dict(o.state_dict()); p.to(dev).train(mode == "train"); p.stochastic = False
    x = synth_images(2, 2, 44)
    with torch.no_grad():
        oc = [copy.deepcopy(n) for n in onets]
        f_o = oc[0](x); raw_o, vol_o = oc[1](f_o); mer_o = oc[2](raw_o, vol_o); ref_o = oc[3](mer_o)
        for math in ("f32", "bf16"):
            ops.set_math(math)
            pc = pn
            f = pc[0](x.to(dev))
            # each stage fed with the ORACLE's input, to see per-module error
            raw, vol = pc[1](f_o.to(dev)); mer = pc[2](raw_o.to(dev), vol_o.to(dev)); ref = pc[3](mer_o.to(dev))
            print(mode, math, "encoder", rel(f, f_o), "decoder(vol)", rel(vol, vol_o), "merger", rel(mer, mer_o), "refiner", rel(ref, ref_o))
        ops.set_math("f32")
