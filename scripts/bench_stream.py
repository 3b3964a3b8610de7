#!/usr/bin/env python3
"""HBM streaming calibration: sv_axpby (2 reads + 1 write) and torch copy on large buffers, bf16 and fp32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swinvox_amd as S
from swinvox_amd import hip
from swinvox_amd.hip import call, ptr
dev = torch.device("cuda", 0); hip.load()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for mb in (32, 128, 512, 2048):
    for dt, code in ((torch.float32, hip.F32), (torch.bfloat16, hip.BF16)):
        n = mb * 1024 * 1024 // (2 if dt == torch.bfloat16 else 4)
        a = torch.randn(n, device=dev).to(dt); b = torch.randn(n, device=dev).to(dt); o = torch.empty_like(a)
        us = timeit(lambda: call("sv_axpby", ptr(a), ptr(b), ptr(o), 1.0, 1.0, n, act=code))
        us2 = timeit(lambda: o.copy_(a))
        print(f"{mb:5d} MB/buffer {str(dt):15s} axpby {us:8.1f} us = {3 * mb * 1.048576 / us * 1e3:7.0f} GB/s    torch copy {us2:8.1f} us = {2 * mb * 1.048576 / us2 * 1e3:7.0f} GB/s")
