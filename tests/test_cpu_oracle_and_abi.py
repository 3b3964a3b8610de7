"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors and the reference's KATs, the host
logic of the drop-in modules, and the C-ABI library (loads, exports every symbol include/swinvox_hip.h declares)."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import oracle as O
import swinvox_amd as S
from swinvox_amd import hip
from swinvox_amd.models import Decoder, Encoder, Merger, Refiner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def synth_images(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    return (0.5 * torch.randn(B, V, 3, 224, 224, generator=g)).clamp(-1, 1)


def synth_gt(B, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return (torch.rand(B, 32, 32, 32, generator=g) < 0.10).float()


def count(m):
    return sum(p.numel() for p in m.parameters())


def test_manifest_records_oracle_pins():
    """The oracle was compared with the reference's own modules when the fixtures were made (make_golden.py)."""
    pins = json.load(open(os.path.join(GOLD, "manifest.json")))["pins"]
    for k in ("decoder_eval_maxdiff", "merger_eval_maxdiff", "refiner_eval_maxdiff", "decoder_train_maxdiff",
              "cva_V1_maxdiff", "cva_V5_maxdiff", "tail_backward_rel_maxdiff", "encoder_plumbing_multi1_maxdiff",
              "encoder_plumbing_multi0_maxdiff",
              # ATT_SPATIAL_DOWNSAMPLE_RATIO = 1 and 2, forward + every gradient vs the reference module (make_cva_ratio_pin.py)
              "cva_ds1_V1_fwd_bwd_maxdiff", "cva_ds1_V3_fwd_bwd_maxdiff", "cva_ds2_V1_fwd_bwd_maxdiff", "cva_ds2_V3_fwd_bwd_maxdiff"):
        assert pins[k] <= 1e-6, k
    for i in range(4):
        assert pins[f"swin_vs_hf_stage{i}_maxdiff"] < 1e-4 * max(1.0, pins[f"swin_vs_hf_stage{i}_absmax"])
    assert pins["notebook_backbone_keys_absent_in_oracle"] == []


def test_parameter_count_kats():
    """Notebook cell 47: Encoder 40,339,770 (single stage) / Decoder 3,817,944 / Refiner 34,880,352 / Merger 17,877."""
    c1 = O.default_cfg(); c1.NETWORK.USE_SWIN_T_MULTI_STAGE = False; c1.NETWORK.SWIN_T_STAGES = [3]
    assert count(O.Encoder(c1)) == 40339770
    c = O.default_cfg()
    assert (count(O.Encoder(c)), count(O.Decoder(c)), count(O.Merger(c)), count(O.Refiner(c))) == (45109818, 3817944, 17877, 34880352)
    p1 = S.default_cfg(); p1.NETWORK.USE_SWIN_T_MULTI_STAGE = False; p1.NETWORK.SWIN_T_STAGES = [3]
    assert count(Encoder(p1)) == 40339770
    pc = S.default_cfg()
    assert (count(Encoder(pc)), count(Decoder(pc)), count(Merger(pc)), count(Refiner(pc))) == (45109818, 3817944, 17877, 34880352)


def test_state_dict_contract():
    """Keys / shapes equal the reference's (oracle pinned by strict load into the reference modules); backbone key
    names also match the checkpoint key list printed in the reference notebook (cell 68)."""
    oc, pc = O.default_cfg(), S.default_cfg()
    for o, p in ((O.Encoder(oc), Encoder(pc)), (O.Decoder(oc), Decoder(pc)), (O.Merger(oc), Merger(pc)), (O.Refiner(oc), Refiner(pc))):
        a, b = o.state_dict(), p.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    keys = json.load(open(os.path.join(GOLD, "notebook_backbone_keys.json")))
    have = set(Encoder(pc).state_dict().keys())
    assert len(keys) > 400 and all(k in have for k in keys)


def test_reference_init_weights_applies_to_holders():
    """utils/helpers.py:20-44 is isinstance-based: it must reach every Conv/Linear/BN holder of the HIP modules."""
    torch.manual_seed(0)
    m = Decoder(S.default_cfg())
    m.apply(O.init_weights)
    assert float(m.layer1[1].weight.min()) == 1.0 and float(m.layer1[1].bias.abs().max()) == 0.0
    assert 0 < float(m.layer2[0].weight.abs().max()) < 0.1


def test_init_weights_matches_reference_recipe():
    """swinvox_amd.helpers.init_weights (utils/helpers.py:20-44 inside the product) draws the same numbers as the oracle's
    restatement from the same seed - on the HIP-backed holders and on the oracle modules - and touches every Conv/Linear/BN."""
    from swinvox_amd.helpers import count_parameters, init_weights
    for mk_p, mk_o in ((lambda: Decoder(S.default_cfg()), lambda: O.Decoder(O.default_cfg())),
                       (lambda: Refiner(S.default_cfg()), lambda: O.Refiner(O.default_cfg())),
                       (lambda: Encoder(S.default_cfg()), lambda: O.Encoder(O.default_cfg()))):
        p, o = mk_p(), mk_o()
        torch.manual_seed(5)
        p.apply(init_weights)
        torch.manual_seed(5)
        o.apply(O.init_weights)
        assert count_parameters(p) == count_parameters(o)
        kinds = (torch.nn.modules.conv._ConvNd, torch.nn.Linear, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)
        om, touched = dict(o.named_modules()), 0
        for name, m in p.named_modules():       # everything the recipe touches (LayerNorms and bias tables keep their constructor values)
            if isinstance(m, kinds):
                for (k, a), (k2, b) in zip(m.named_parameters(recurse=False), om[name].named_parameters(recurse=False)):
                    assert k == k2 and torch.equal(a, b), f"{name}.{k}"
                    touched += a.numel()
        assert touched > 0.95 * count_parameters(p) or isinstance(p, Encoder)
    lin = torch.nn.Linear(64, 64)
    init_weights(lin)
    assert float(lin.bias.abs().max()) == 0.0 and 5e-4 < float(lin.weight.std()) < 2e-3     # N(0, 0.01) * 0.1


def test_oracle_matches_golden_vectors():
    cfg = O.default_cfg()
    nets = [O.Encoder(cfg), O.Decoder(cfg), O.Merger(cfg), O.Refiner(cfg)]
    for i, n in enumerate(nets):
        O.seeded_weights_(n, seed=100 + i)
    O.calibrate_(nets, synth_images(2, 2, 1234))
    for n in nets:
        n.eval()
    man = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]
    for key in ("B2_V1", "B1_V2", "B2_V8"):
        B, V = int(key[1]), int(key[4])
        gold = np.load(os.path.join(GOLD, f"case_{key}.npz"))
        x, gt = synth_images(B, V, man[key]["seed"]), synth_gt(B, man[key]["seed"])
        with torch.no_grad():
            f = nets[0](x)
            raw, vol = nets[1](f)
            merged = nets[2](raw, vol)
            refined = nets[3](merged)
        assert np.abs(f.numpy() - gold["features"]).max() < 1e-4 * np.abs(gold["features"]).max()
        assert np.abs(refined.numpy() - gold["refined"]).max() < 1e-3
        assert np.abs(np.array(O.iou_at_thresholds(refined, gt)) - gold["iou"]).max() < 1e-3
        assert 1.0 < float(refined.std()) < 3.0       # calibrated logits, not the degenerate init_weights regime


def test_swin_b_pin_and_golden():
    """BASELINE config 5's encoder variant: the manifest records the oracle-vs-transformers pin of the Swin-B backbone, the oracle
    reproduces its committed golden features, and the product's holder tree has the same parameters."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    for i in range(4):
        assert man["pins"][f"swin_b_vs_hf_stage{i}_maxdiff"] < 1e-4 * max(1.0, man["pins"][f"swin_b_vs_hf_stage{i}_absmax"])
    case = man["cases"]["swin_b_B1_V2"]
    enc = O.Encoder(O.default_cfg(), variant="base")
    O.seeded_weights_(enc, seed=case["weights_seed"])
    enc.eval()
    with torch.no_grad():
        f = enc(synth_images(1, 2, case["seed"]))
    gold = np.load(os.path.join(GOLD, "case_swin_b_B1_V2.npz"))["features"]
    assert np.abs(f.numpy() - gold).max() < 1e-4 * np.abs(gold).max()
    p = Encoder(S.default_cfg(), variant="base")
    assert list(p.state_dict().keys()) == list(enc.state_dict().keys())
    assert sum(q.numel() for q in p.parameters()) == man["pins"]["swin_b_encoder_params"] == 104832376


def test_iou_edge_cases():
    """core/test.py:141-153: both empty -> 1.0; prediction empty, gt not -> 0."""
    z = torch.full((1, 32, 32, 32), -20.0)
    assert O.iou_at_thresholds(z, torch.zeros(1, 32, 32, 32)) == [[1.0] * 4]
    g = torch.zeros(1, 32, 32, 32); g[0, 0, 0, 0] = 1
    assert O.iou_at_thresholds(z, g) == [[0.0] * 4]
    assert O.iou_at_thresholds(-z, g)[0][0] == pytest.approx(1 / 32768)


def test_library_builds_and_exports_every_declared_symbol():
    assert os.path.exists(hip.LIB_PATH), "run `python __graft_entry__.py build` first"
    hdr = open(os.path.join(ROOT, "include", "swinvox_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", hdr)))
    lib = ctypes.CDLL(hip.LIB_PATH)           # dlopen only: no GPU call is made
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(hip.EXPORTED_SYMBOLS) == declared, set(hip.EXPORTED_SYMBOLS) ^ set(declared)
    nm = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (sv_[a-z0-9_]+)", nm)))
    assert exported == declared


def test_tile_draw_register_is_untouched_by_the_compiler():
    """gemm_wide_kernel parks the return of its tile-draw atomic in v167 across inline-asm statements (csrc/igemm.hip); nothing but this
    check keeps the register allocator out of it.  Disassembles the SHIPPED library: every use of v167 inside every gemm_wide_kernel must
    be one of the three hand-written instructions, and the checker itself must flag a compiler-style use."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_tile_draw_register", os.path.join(ROOT, "scripts", "check_tile_draw_register.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    if not os.path.exists(chk.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain is not installed here")
    assert chk.check(hip.LIB_PATH) == []
    # the checker's own teeth: ranges and plain uses are seen, the three hand-written forms pass
    assert chk.names_reg("v_pk_mul_f32 v[166:167], v[2:3], v[4:5]") and chk.names_reg("v_add_f32 v1, v167, v2")
    assert not chk.names_reg("v_mov_b32 v16, v1670") and not chk.names_reg("ds_read_b128 v[160:163], v5")
    assert chk.allowed("v_mov_b32_e32 v167, -1") == "sentinel" and chk.allowed("v_mov_b32_e32 v32, v167") == "read"
    assert chk.allowed("global_atomic_add v167, v[8:9], v87, off sc0") == "draw" and chk.allowed("v_add_u32 v167, v1, v2") == ""


def test_ctypes_prototypes_match_the_header():
    """Every prototype of include/swinvox_hip.h against the ctypes argument list the host side binds (count and class of
    every argument: pointer / int / long long / float / uint32), and the workspace-size helpers against the host constants
    (pure host functions, no GPU)."""
    from swinvox_amd import ops
    hdr = open(os.path.join(ROOT, "include", "swinvox_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = re.findall(r"\b(?:int|size_t)\s+(sv_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 50

    def code(a):
        a = a.strip()
        if a in ("void", ""):
            return None
        if "*" in a:
            return "P"
        for key, c in (("long long", "L"), ("uint32_t", "U"), ("double", "D"), ("float", "F"), ("int", "I")):
            if key in a:
                return c
        raise AssertionError(a)

    cmap = {ctypes.c_void_p: "P", ctypes.c_int: "I", ctypes.c_longlong: "L", ctypes.c_float: "F", ctypes.c_uint32: "U", ctypes.c_double: "D"}
    for name, args in protos:
        want = [c for c in (code(a) for a in args.split(",")) if c]
        got = [cmap.get(t, "P") for t in hip._argtypes(name)]
        assert want == got, (name, "".join(want), "".join(got))
    lib = hip.load()
    assert lib.sv_bn_bwd_workspace_doubles(64) == (ops.BN_BWD_SLOTS + 1) * 2 * 64 + 2
    assert lib.sv_layernorm_bwd_workspace_floats(96) == 2 * (ops.LN_BWD_SLOTS * 96 + 1)
    assert lib.sv_stencil3_wgrad_workspace_floats(9, 36) == 2 * (8 * 9 * 36 * 27)
    assert lib.sv_window_attention_bwd_workspace_floats(3) == 2 * (8 * 169 * 3)
    assert lib.sv_pack_weights_block_elems() > 0 and ctypes.sizeof(hip.PackDesc) == 48


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    m = Refiner(S.default_cfg())
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 32, 32, 32))
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libswinvox_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        hip.load()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "swinvox_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
