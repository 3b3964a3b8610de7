#!/usr/bin/env python3
"""Micro-benchmark of the fused stage-0 attention branch (sv_swin_attn_block_fwd / _bwd) against the unfused chains they replace
(forward: sv_layernorm_fwd -> qkv linear -> sv_window_attention_fwd -> proj linear + residual; backward data path: proj data gradient ->
sv_window_attention_bwd -> qkv data gradient -> sv_layernorm_bwd), I = 512 images of 56 x 56 tokens, bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip, ops
from swinvox_amd.hip import call, ptr
from swinvox_amd.ops import ConvSpec
if os.environ.get("SV_LIB"):
    hip.LIB_PATH = os.environ["SV_LIB"]      # A/B builds of the library
dev = torch.device("cuda", 0); hip.load(); S.set_math("bf16"); S.set_storage("bf16")
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
I, H, C, heads = int(os.environ.get("SV_ATTN_I", "512")), 56, 96, 3
M = I * H * H
b16 = dict(dtype=torch.bfloat16, device=dev)
x = torch.randn(M, C, device=dev).bfloat16()
lg, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
wq, bq = torch.randn(3 * C, C, device=dev) / C ** 0.5, torch.zeros(3 * C, device=dev)
wp, bp = torch.randn(C, C, device=dev) / C ** 0.5, torch.zeros(C, device=dev)
table = torch.randn(169, heads, device=dev) * 0.1
x1, ln1, qkv, att = torch.empty(M, C, **b16), torch.empty(M, C, **b16), torch.empty(M, 3 * C, **b16), torch.empty(M, C, **b16)
m1, r1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
flops = 2.0 * M * C * 4 * C + 4.0 * 49 * 32 * M * heads
unit = M * C * 2 / 1e6    # MB of one [M, C] bf16 tensor
s_qkv, s_proj = ConvSpec.linear(C, 3 * C), ConvSpec.linear(C, C)
for shift in (0, 3):
    def fused(side):
        call("sv_swin_attn_block_fwd", ptr(x), ptr(lg), ptr(lb), ptr(wq), ptr(bq), ptr(table), ptr(wp), ptr(bp), None, ptr(x1),
             ptr(ln1) if side else None, ptr(m1) if side else None, ptr(r1) if side else None, ptr(qkv) if side else None, ptr(att) if side else None,
             I, H, H, C, heads, shift, 1e-5)
    def chain():
        call("sv_layernorm_fwd", ptr(x), ptr(lg), ptr(lb), ptr(ln1), ptr(m1), ptr(r1), M, C, 1e-5, 0, 0)
        ops.linear_fwd(ln1, M, s_qkv, wq, qkv, bias=bq)
        call("sv_window_attention_fwd", ptr(qkv), ptr(table), ptr(att), I, H, H, C, heads, shift, hip.MATH_BF16)
        ops.linear_fwd(att, M, s_proj, wp, x1, bias=bp, residual=x, ldr=C)
    tt, ti, tc = timeit(lambda: fused(True)), timeit(lambda: fused(False)), timeit(chain)
    print(f"shift={shift}  fused+side {tt:7.1f} us ({7*unit/tt*1e3:6.0f} GB/s, {flops/tt/1e6:6.1f} TF/s)   fused lean {ti:7.1f} us ({2*unit/ti*1e3:6.0f} GB/s, "
          f"{flops/ti/1e6:6.1f} TF/s)   unfused chain {tc:7.1f} us ({13*unit/tc*1e3:6.0f} GB/s)")

    # ---- backward data path
    dx1 = torch.randn(M, C, device=dev).bfloat16()
    dqkv, dx, datt, dln1 = torch.empty(M, 3 * C, **b16), torch.empty(M, C, **b16), torch.empty(M, C, **b16), torch.empty(M, C, **b16)
    dg, db, dt = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(169, heads, device=dev)
    ws = torch.zeros(int(hip.load().sv_window_attention_bwd_workspace_floats(heads)), device=dev)
    wsl = torch.zeros(2 * (32 * C + 1), device=dev)
    bflops = 2.0 * M * C * 4 * C + 10.0 * 49 * 32 * M * heads
    def fused_bwd():
        call("sv_swin_attn_block_bwd", ptr(dx1), ptr(qkv), ptr(x), ptr(m1), ptr(r1), ptr(lg), ptr(wq), ptr(wp), ptr(table), None, ptr(dqkv), ptr(dx), None,
             ptr(dg), ptr(db), ptr(dt), ptr(ws), I, H, H, C, heads, shift)
    wpd, wqd = s_proj.pack_dgrad(wp), s_qkv.pack_dgrad(wq)
    def chain_bwd():
        ops.linear_dgrad(dx1, M, s_proj, wpd, datt)
        call("sv_window_attention_bwd", ptr(qkv), ptr(table), ptr(datt), ptr(dqkv), ptr(dt), ptr(ws), I, H, H, C, heads, shift, hip.MATH_BF16)
        ops.linear_dgrad(dqkv, M, s_qkv, wqd, dln1)
        call("sv_layernorm_bwd", ptr(dln1), ptr(x), ptr(lg), ptr(m1), ptr(r1), ptr(dx), ptr(dg), ptr(db), ptr(wsl), M, C, 0, 0, 0)
    tf, tcb = timeit(fused_bwd), timeit(chain_bwd)
    print(f"shift={shift}  backward: fused {tf:7.1f} us ({9*unit/tf*1e3:6.0f} GB/s, {bflops/tf/1e6:6.1f} TF/s)   unfused chain {tcb:7.1f} us ({17*unit/tcb*1e3:6.0f} GB/s)")
