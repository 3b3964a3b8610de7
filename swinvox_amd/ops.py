"""Host-side operator layer over the C ABI: geometry bookkeeping, weight packing, forward / data-grad /
weight-grad calls of the contraction engine, and the normalisation / pooling wrappers.

Everything here only enqueues HIP kernels on the current torch stream; torch is used for buffer
allocation (caching allocator) and nothing else.  Activations are channels-last and live in HBM in the current STORAGE
dtype (fp32, or bf16 under set_storage("bf16")); parameters, their gradients and all statistics are always fp32.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import hip
from .hip import ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, Epilogue, Geom, call, ptr  # noqa: F401

# Process-wide CONFIGURATION (set_math / set_storage / set_overlap): read-only while modules run.
_STATE = {"math": hip.MATH_F32, "store": torch.float32}


class _CallContext(threading.local):
    """Per-thread context of the module pass that is running on this thread (weight-pack cache, zero arena, weight-gradient
    stream, BatchNorm tick list).  torch.nn.DataParallel (reference core/train.py:156-161) calls the modules' forward from
    one Python thread per device, so nothing mutable on the launch path may be shared between threads."""
    packs = None
    arena = None
    awg = None
    bn_tick = None


_CTX = _CallContext()
BN_SLOTS = 16   # SV_BN_SLOTS of include/swinvox_hip.h
BN_BWD_SLOTS, LN_BWD_SLOTS = 16, 32   # slot counts behind sv_bn_bwd_workspace_doubles / sv_layernorm_bwd_workspace_floats


def set_math(mode: str) -> None:
    """'f32' = exact fp32 MFMA (parity mode); 'bf16' = bf16 MFMA inputs, fp32 accumulate.  Switching to 'f32' also
    switches the activation storage back to fp32 (bf16 storage needs bf16 math)."""
    _STATE["math"] = {"f32": hip.MATH_F32, "fp32": hip.MATH_F32, "bf16": hip.MATH_BF16}[mode]
    if _STATE["math"] == hip.MATH_F32:
        set_storage("f32")


def get_math() -> str:
    return "bf16" if _STATE["math"] == hip.MATH_BF16 else "f32"


def set_storage(mode: str) -> None:
    """HBM element type of the activations (and activation gradients) INSIDE the four modules: 'f32' or 'bf16'.
    Module inputs / outputs stay fp32 torch tensors (converted at the boundary); arithmetic, statistics, parameters and
    parameter gradients stay fp32.  'bf16' halves the activation traffic of the path and requires set_math('bf16')."""
    dt = {"f32": torch.float32, "fp32": torch.float32, "bf16": torch.bfloat16}[mode]
    if dt == torch.bfloat16 and _STATE["math"] != hip.MATH_BF16:
        raise RuntimeError("swinvox_amd: bf16 activation storage requires set_math('bf16')")
    _STATE["store"] = dt
    hip.ACT = hip.BF16 if dt == torch.bfloat16 else hip.F32


def get_storage() -> str:
    return "bf16" if _STATE["store"] == torch.bfloat16 else "f32"


def empty(*shape, like: torch.Tensor = None, device=None) -> torch.Tensor:
    """ACTIVATION buffer (current storage dtype)."""
    return torch.empty(*shape, dtype=_STATE["store"], device=like.device if like is not None else device)


def zeros(*shape, like: torch.Tensor = None, device=None) -> torch.Tensor:
    """Zero-filled ACTIVATION buffer (current storage dtype)."""
    return torch.zeros(*shape, dtype=_STATE["store"], device=like.device if like is not None else device)


def fempty(*shape, like: torch.Tensor = None, device=None) -> torch.Tensor:
    """fp32 buffer: weight packs, statistics, workspaces, parameter-shaped scratch."""
    return torch.empty(*shape, dtype=torch.float32, device=like.device if like is not None else device)


def fzeros(*shape, like: torch.Tensor = None, device=None) -> torch.Tensor:
    return torch.zeros(*shape, dtype=torch.float32, device=like.device if like is not None else device)


def to_store(t: torch.Tensor) -> torch.Tensor:
    """fp32 module input -> contiguous activation in the current storage dtype (no-op copy avoided for fp32 storage)."""
    t = t.contiguous()
    if t.dtype == _STATE["store"]:
        return t
    out = torch.empty(t.shape, dtype=_STATE["store"], device=t.device)
    call("sv_cast", ptr(t), hip.F32 if t.dtype == torch.float32 else hip.BF16, ptr(out), hip.ACT, t.numel())
    return out


def to_f32(t: torch.Tensor) -> torch.Tensor:
    """contiguous activation -> fp32 module output."""
    if t.dtype == torch.float32:
        return t
    assert t.is_contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    call("sv_cast", ptr(t), hip.BF16, ptr(out), hip.F32, t.numel())
    return out


def _t3(v) -> Tuple[int, int, int]:
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(x) for x in v)
    return (1,) * (3 - len(v)) + v if len(v) < 3 else v


def _pad3(v, fill) -> Tuple[int, int, int]:
    """2-D parameters are lifted to 3-D with a unit depth axis."""
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(x) for x in v)
    return (fill,) * (3 - len(v)) + v


def _epilogue(ldc, col_off=0, bias=None, residual=None, ldr=0, row_scale=None, rows_per_scale=1, pre_act=None, stats=None,
              act=ACT_NONE, slope=0.0, act_grad_src=None, act_grad_kind=ACT_NONE) -> Epilogue:
    return Epilogue(ptr(bias), ptr(residual), ldr, ptr(row_scale), rows_per_scale, ptr(pre_act), ptr(stats), act, slope,
                    ptr(act_grad_src), act_grad_kind, ldc, col_off)


@dataclass
class ConvSpec:
    """Geometry of one Linear / Conv / ConvTranspose layer (3-D form; 2-D layers use a unit depth axis)."""
    cin: int
    cout: int
    k: Tuple[int, int, int] = (1, 1, 1)
    s: Tuple[int, int, int] = (1, 1, 1)
    p: Tuple[int, int, int] = (0, 0, 0)
    transposed: bool = False
    cin_mem: Optional[int] = None    # channels the input activation holds in memory (zero-padded), default cin
    cout_mem: Optional[int] = None   # channels the output-gradient holds in memory, default cout
    og_fixed: Optional[Tuple[int, int, int]] = None   # output grid when it is not the symmetric-padding formula (asymmetric pads)

    def __post_init__(self):
        self.cin_mem = self.cin_mem or self.cin
        self.cout_mem = self.cout_mem or self.cout
        self.taps = self.k[0] * self.k[1] * self.k[2]

    @staticmethod
    def conv2d(cin, cout, k, s=1, p=0, **kw):
        return ConvSpec(cin, cout, _pad3(k if not isinstance(k, int) else (k, k), 1), _pad3(s if not isinstance(s, int) else (s, s), 1),
                        _pad3(p if not isinstance(p, int) else (p, p), 0), False, **kw)

    @staticmethod
    def conv3d(cin, cout, k, s=1, p=0, transposed=False, **kw):
        return ConvSpec(cin, cout, _t3(k), _t3(s), _t3(p), transposed, **kw)

    @staticmethod
    def linear(cin, cout):
        return ConvSpec(cin, cout)

    def out_grid(self, g):
        if self.og_fixed is not None:
            return tuple(self.og_fixed)
        if self.transposed:
            return tuple((g[i] - 1) * self.s[i] - 2 * self.p[i] + self.k[i] for i in range(3))
        return tuple((g[i] + 2 * self.p[i] - self.k[i]) // self.s[i] + 1 for i in range(3))

    # ---- weight packs (device copies made once per step; element type = the activation storage dtype) --------
    def pack_args(self, kind: str):
        """(A, B, T, swap, pad_to, rows_out, inner_out) of sv_pack_weight for the forward ('f': [cout][tap][cin_mem]) or the
        data-gradient ('d': [cin][tap][cout_mem]) pack of the native parameter ([cout,cin,k..] conv / [cin,cout,k..] tconv)."""
        A, B = (self.cin, self.cout) if self.transposed else (self.cout, self.cin)
        if kind == "f":
            swap, pad = (1 if self.transposed else 0), self.cin_mem
        else:
            swap, pad = (0 if self.transposed else 1), self.cout_mem
        rows, inner = (B, max(A, pad)) if swap else (A, max(B, pad))
        return A, B, self.taps, swap, pad, rows, inner

    def _pack(self, w: torch.Tensor, kind: str) -> torch.Tensor:
        dt = _STATE["store"]
        if kind == "f" and dt == torch.float32 and not self.transposed and self.taps == 1 and self.cin_mem == self.cin:
            return w  # Linear / 1x1 conv in fp32: the native layout already is the packed layout
        cache = _CTX.packs
        if cache is not None and isinstance(w, torch.nn.Parameter):   # temporaries (re-indexed copies) would add an entry per step
            return cache.get(self, w, kind)
        return pack_one(self, w, kind)

    def pack_fwd(self, w: torch.Tensor) -> torch.Tensor:
        return self._pack(w, "f")

    def pack_dgrad(self, w: torch.Tensor) -> torch.Tensor:
        return self._pack(w, "d")

    def _geom(self, n, gathered_grid, produced_grid, ci, co, ldi) -> Geom:
        return Geom(n, *gathered_grid, *produced_grid, ci, co, *self.k, *self.s, *self.p, ldi)

    def _traced(self, name, n, in_grid, *args):
        """Launch a contraction; when bench.py's tracer is active, bracket it with HIP events and book its ALGORITHMIC
        work: 2 * positions * taps_that_contribute * cin * cout flops (no channel padding, no masked taps)."""
        tr = hip.TRACE
        if tr is None or name not in tr.names:
            return call(name, *args)
        og = self.out_grid(in_grid)
        if self.transposed:   # every input position meets every tap exactly once
            pairs = n * in_grid[0] * in_grid[1] * in_grid[2] * self.taps
        else:                 # interior count; border taps falling into the padding are a small over-estimate (<3%)
            pairs = n * og[0] * og[1] * og[2] * self.taps
        flops = 2.0 * pairs * self.cin * self.cout
        nin, nout = n * in_grid[0] * in_grid[1] * in_grid[2], n * og[0] * og[1] * og[2]
        esz = 2.0 if _STATE["store"] == torch.bfloat16 else 4.0
        nbytes = esz * (nin * self.cin + nout * self.cout) + 4.0 * self.taps * self.cin * self.cout
        tr.begin(name, flops, nbytes, tag=f"n={n} grid={in_grid} cin={self.cin} cout={self.cout} k={self.k} s={self.s} t={int(self.transposed)}")
        call(name, *args)
        tr.end()

    def _halo_tconv(self, in_grid, ldi, ldc, epi) -> bool:
        return (self.transposed and self.k == (4, 4, 4) and self.s == (2, 2, 2) and self.p == (1, 1, 1) and self.cin == 32
                and self.cin_mem == 32 and self.cout == 8 and _STATE["store"] == torch.bfloat16 and _STATE["math"] != hip.MATH_F32
                and ldi in (None, 32) and ldc in (None, 8) and not (set(epi) - {"bias", "stats"})
                and in_grid[0] % 4 == 0 and in_grid[1] % 4 == 0 and in_grid[2] % 8 == 0)

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x, n, in_grid, w_packed, out, *, ldi=None, ldc=None, **epi):
        og = self.out_grid(in_grid)
        if isinstance(w_packed, torch.nn.Parameter) or w_packed.dtype != _STATE["store"]:
            w_packed = self.pack_fwd(w_packed)     # raw parameter: pack / convert (identity for a Linear in fp32 storage)
        if self._halo_tconv(in_grid, ldi, ldc, epi):
            # k4 s2 p1, 32 -> 8 channels (decoder layer4): parity-class stencils on an LDS halo brick instead of L2 gathers
            self._traced("sv_tconv4s2_fwd", n, in_grid, ptr(x), ptr(w_packed), ptr(epi.get("bias")), ptr(out), ptr(epi.get("stats")),
                         n, *in_grid, self.cin, self.cout)
            return og
        g = self._geom(n, in_grid, og, self.cin_mem, self.cout, ldi or self.cin_mem)
        e = _epilogue(ldc or self.cout, **epi)
        self._traced("sv_tconv_gather" if self.transposed else "sv_conv_gather", n, in_grid, ptr(x), ptr(w_packed), ptr(out),
                     C.byref(g), C.byref(e), _STATE["math"])
        return og

    # ---- data gradient: dx[., cin] from dy[., cout_mem] -----------------------------------------------
    def dgrad(self, dy, n, in_grid, w_dgrad, dx, *, lddy=None, lddx=None, **epi):
        og = self.out_grid(in_grid)
        g = self._geom(n, og, in_grid, self.cout_mem, self.cin, lddy or self.cout_mem)
        e = _epilogue(lddx or self.cin_mem, **epi)
        # 1x1 / stride 1 / pad 0 (every Linear and 1x1 conv): the data-gradient is the same dense gather with the
        # transposed weight pack -> use the plain gather kernel, no parity classes
        dense = self.taps == 1 and self.s == (1, 1, 1) and self.p == (0, 0, 0)
        self._traced("sv_conv_gather" if (self.transposed or dense) else "sv_tconv_gather", n, in_grid, ptr(dy), ptr(w_dgrad), ptr(dx),
                     C.byref(g), C.byref(e), _STATE["math"])

    # ---- weight gradient, accumulated into dw (native layout) ------------------------------------------
    def wgrad(self, dy, x, n, in_grid, dw, *, lddy=None, ldx=None, db=None, async_ok=True):
        """dw += ...; for (non-transposed) conv / Linear layers db (bias gradient = column sums of dy) is folded into the
        same kernel; transposed convs anchor on x, so their bias gradient needs the separate column-sum kernel.
        Inside a module backward the launch goes to the weight-gradient stream (AsyncWgrad): nothing on the data-gradient
        chain waits for a weight gradient.  async_ok=False keeps it on the caller's stream - required when dy or x is
        modified in place afterwards."""
        aw = _CTX.awg
        if aw is not None and async_ok:
            cur = torch.cuda.current_stream()
            aw.stream.wait_stream(cur)                 # dy and x are complete on the producing stream
            aw.held.append((dy, x))                    # keep the operands alive (and unrecycled) until the join
            with torch.cuda.stream(aw.stream):
                self._wgrad(dy, x, n, in_grid, dw, lddy, ldx, db)
            return
        self._wgrad(dy, x, n, in_grid, dw, lddy, ldx, db)

    def _wgrad(self, dy, x, n, in_grid, dw, lddy, ldx, db):
        og = self.out_grid(in_grid)
        if self.transposed:   # anchor = x (cin), gathered = dy (cout)
            g = self._geom(n, og, in_grid, self.cout_mem, self.cin, lddy or self.cout_mem)
            ws = fempty(self.cin * self.taps * self.cout_mem, like=dw) if self.taps > 1 else None
            self._traced("sv_conv_wgrad", n, in_grid, ptr(x), ldx or self.cin_mem, ptr(dy), ptr(dw), C.byref(g), self.cout, ptr(ws),
                         None, _STATE["math"])
            if db is not None:
                colsum(dy, n * og[0] * og[1] * og[2], self.cout, lddy or self.cout_mem, db)
        else:                 # anchor = dy (cout), gathered = x (cin)
            g = self._geom(n, in_grid, og, self.cin_mem, self.cout, ldx or self.cin_mem)
            ws = fempty(self.cout * self.taps * self.cin_mem, like=dw) if self.taps > 1 else None
            self._traced("sv_conv_wgrad", n, in_grid, ptr(dy), lddy or self.cout_mem, ptr(x), ptr(dw), C.byref(g), self.cin, ptr(ws),
                         ptr(db), _STATE["math"])


def pack_one(spec: "ConvSpec", w: torch.Tensor, kind: str) -> torch.Tensor:
    """One weight pack with its own launch (first step of a module, tests)."""
    A, B, T, swap, pad, rows, inner = spec.pack_args(kind)
    out = torch.empty(rows * T * inner, dtype=_STATE["store"], device=w.device)
    call("sv_pack_weight", ptr(w), ptr(out), A, B, T, swap, pad, hip.ACT)
    return out


class PackCache:
    """All weight packs of one module from ONE batched launch per forward (sv_pack_weights).

    The first forward/backward of a module packs each weight individually and registers the request; from the next
    refresh() on, every registered pack is produced by one launch into a flat buffer and get() only hands out views.
    Packs made at forward time serve the backward of the same step (weights change only in optimizer.step())."""

    def __init__(self):
        self.tables = {}   # storage dtype -> dict(entries={key: [spec, w, kind, view]}, dirty, flat, descs, sig, nblocks)

    def __deepcopy__(self, memo):
        return PackCache()    # keyed by parameter identity: a copied module starts with an empty cache

    def _table(self):
        return self.tables.setdefault(_STATE["store"], dict(entries={}, dirty=False, flat=None, descs=None, sig=None, nblocks=0))

    def get(self, spec, w, kind):
        t = self._table()
        key = (id(w), kind, spec.cin_mem, spec.cout_mem)
        e = t["entries"].get(key)
        if e is not None and e[3] is not None:
            return e[3]
        if e is None:
            t["entries"][key] = [spec, w, kind, None]
            t["dirty"] = True
        return pack_one(spec, w, kind)

    def refresh(self):
        """(Re)build the descriptor table when the set of packs or a parameter's address changed, then pack everything."""
        t = self._table()
        ents = list(t["entries"].values())
        if not ents:
            return
        sig = tuple(e[1].data_ptr() for e in ents)
        if t["dirty"] or sig != t["sig"]:
            per_block = int(hip.load().sv_pack_weights_block_elems())
            dev = ents[0][1].device
            offs, total, blocks = [], 0, 0
            descs = (hip.PackDesc * len(ents))()
            for i, (spec, w, kind, _) in enumerate(ents):
                A, B, T, swap, pad, rows, inner = spec.pack_args(kind)
                n = rows * T * inner
                offs.append((total, n))
                descs[i].src, descs[i].A, descs[i].B, descs[i].T, descs[i].swap = w.data_ptr(), A, B, T, swap
                descs[i].rows_out, descs[i].inner_out, descs[i].block0 = rows, inner, blocks
                blocks += (n + per_block - 1) // per_block
                total += (n + 7) // 8 * 8          # every pack starts on a 16-byte boundary
            flat = torch.empty(max(total, 8), dtype=_STATE["store"], device=dev)
            esz = flat.element_size()
            for i, (o, n) in enumerate(offs):
                descs[i].dst = flat.data_ptr() + o * esz
                ents[i][3] = flat[o:o + n]
            raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
            t.update(dirty=False, flat=flat, descs=raw, sig=sig, nblocks=blocks)
        call("sv_pack_weights", ptr(t["descs"]), len(ents), t["nblocks"], hip.ACT)


class ZeroArena:
    """Zero-filled double scratch (BatchNorm statistic accumulators) for one module pass from ONE fill launch: the first
    pass measures the demand with individual torch.zeros calls, later passes slice one pre-zeroed buffer."""

    def __init__(self):
        self.need, self.used, self.buf, self.off = 0, 0, None, 0

    def begin(self, dev):
        self.used, self.off = 0, 0
        self.buf = torch.zeros(self.need, dtype=torch.float64, device=dev) if self.need else None

    def take(self, n, dev):
        self.used += n
        if self.buf is not None and self.off + n <= self.buf.numel():
            s = self.buf[self.off:self.off + n]
            self.off += n
            return s
        return torch.zeros(n, dtype=torch.float64, device=dev)

    def end(self):
        self.need, self.buf = max(self.need, self.used), None

    def __deepcopy__(self, memo):
        return ZeroArena()


def zeros_f64(n, dev):
    a = _CTX.arena
    return a.take(n, dev) if a is not None else torch.zeros(n, dtype=torch.float64, device=dev)


def set_arena(arena) -> None:
    _CTX.arena = arena


def bn_tick_flush() -> None:
    """num_batches_tracked += 1 for every BatchNorm that ran since the last flush, in one foreach launch."""
    lst = _CTX.bn_tick
    if lst:
        torch._foreach_add_(lst, 1)
    _CTX.bn_tick = []


class AsyncWgrad:
    """Weight-gradient stream of one module backward (see ConvSpec.wgrad).  join() makes the caller's stream wait for every
    weight gradient and releases the operand references."""
    _streams = {}

    def __init__(self, dev):
        key = (dev.type, dev.index)
        if key not in AsyncWgrad._streams:
            AsyncWgrad._streams[key] = torch.cuda.Stream(device=dev)
        self.stream, self.held = AsyncWgrad._streams[key], []

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.held.clear()


def set_async_wgrad(aw) -> None:
    _CTX.awg = aw


_SIDE = {}


def side_stream(dev) -> "torch.cuda.Stream":
    """Second HIP stream of a device for the independent ResNet / Swin branches of the encoder (forward and backward).
    Usage: side.wait_stream(main) [fork] ... with torch.cuda.stream(side): branch ... main.wait_stream(side) [join].
    Allocator safety: tensors allocated while the side stream is current return to ITS pool and are only re-used by later
    side-stream allocations, which follow a later fork (= wait on everything the main stream had queued); main-stream
    tensors read by the side branch must stay referenced until the join (callers keep them in locals / the tape)."""
    key = (dev.type, dev.index)
    if key not in _SIDE:
        # high priority: the Swin backbone that runs here is the longer of the two encoder branches (measured +0.35 %)
        _SIDE[key] = torch.cuda.Stream(device=dev, priority=-1)
    return _SIDE[key]


def set_seed_epoch(t: Optional[torch.Tensor]) -> None:
    """Device word (int32/uint32 tensor with one element, or None) that every stochastic launch mixes into its scalar seed ON
    THE DEVICE.  A captured hipGraph freezes scalar kernel arguments; a caller that replays a graph advances this word once per
    replay (graph.GraphedStep does) so that dropout / drop-path masks differ from step to step.  Process-wide configuration."""
    if t is not None and (not t.is_cuda or t.numel() != 1 or t.element_size() != 4):
        raise ValueError("seed epoch must be a one-element 32-bit tensor on the GPU")
    _STATE["seed_epoch"] = t


def seed_epoch_ptr():
    t = _STATE.get("seed_epoch")
    return t.data_ptr() if t is not None else None


_COMM = {}


def comm_stream(dev) -> "torch.cuda.Stream":
    """Staging stream of the data-parallel gradient hand-off (HipModule._announce)."""
    key = (dev.type, dev.index)
    if key not in _COMM:
        _COMM[key] = torch.cuda.Stream(device=dev)
    return _COMM[key]


def overlap_enabled() -> bool:
    return _STATE.get("overlap", True)


def set_overlap(on: bool) -> None:
    """Run the encoder's two backbone branches on two HIP streams (default on)."""
    _STATE["overlap"] = bool(on)


def set_pack_cache(cache) -> None:
    """Route ConvSpec.pack_fwd / pack_dgrad through `cache` (a PackCache, or None for one launch per pack)."""
    _CTX.packs = cache


def traced_call(name, flops, nbytes, *args, tag=""):
    """call() that books ALGORITHMIC flops / bytes with bench.py's tracer when it listens to `name`."""
    tr = hip.TRACE
    if tr is None or name not in tr.names:
        return call(name, *args)
    tr.begin(name, flops, nbytes, tag=tag)
    call(name, *args)
    tr.end()


def fused_mlp_enabled(C: int) -> bool:
    """The fused Swin MLP kernels (csrc/swin_mlp.hip) serve bf16 MFMA + bf16 storage and the channel counts they are built for."""
    import os
    # C = 192 (Swin-T stage 1) is built too, but since the wide dense kernels (csrc/igemm.hip) serve K = 192 the unfused chain is the faster
    # one there: 123.0 -> 121.7 ms per step at B = 64.  SV_FUSED_MLP_MAX_C overrides the limit for measurements.
    lim = int(os.environ.get("SV_FUSED_MLP_MAX_C", "128"))
    return (_STATE.get("fused_mlp", True) and _STATE["math"] == hip.MATH_BF16 and _STATE["store"] == torch.bfloat16
            and C <= lim and hip.load().sv_swin_mlp_supported(C) == 1)


def fused_attn_block_enabled(C: int, heads: int) -> bool:
    """The fused attention branch (LayerNorm -> qkv -> window attention -> proj -> residual, csrc/attn.hip) serves stage 0 of Swin-T
    (C = 96, 3 heads) in bf16 MFMA + bf16 storage; fp8 attention keeps the unfused chain (its core kernel is a separate build)."""
    import os
    if os.environ.get("SV_FUSED_ATTN_BLOCK", "1") == "0" or not _STATE.get("fused_attn_block", True) or _STATE.get("attn_fp8"):
        return False
    return (_STATE["math"] == hip.MATH_BF16 and _STATE["store"] == torch.bfloat16
            and hip.load().sv_swin_attn_block_supported(C, heads, hip.BF16, hip.MATH_BF16) == 1)


def fused_attn_block_bwd_enabled(C: int, heads: int) -> bool:
    """The fused BACKWARD of the attention branch (csrc/attn.hip swin_attn_block_bwd_kernel) serves the same blocks as the fused forward; it
    reads what either forward stores (qkv, LayerNorm statistics), so fp8 attention in the forward does not switch it off.
    A/B: SV_FUSED_ATTN_BWD=0 or set_fused_attn_block_bwd(False) route the branch through the unfused four-kernel chain."""
    import os
    if os.environ.get("SV_FUSED_ATTN_BWD", "1") == "0" or not _STATE.get("fused_attn_block_bwd", True):
        return False
    return (_STATE["math"] == hip.MATH_BF16 and _STATE["store"] == torch.bfloat16
            and hip.load().sv_swin_attn_block_supported(C, heads, hip.BF16, hip.MATH_BF16) == 1)


def set_fused_attn_block_bwd(on: bool) -> None:
    _STATE["fused_attn_block_bwd"] = bool(on)


def set_fused_attn_block(on: bool) -> None:
    """A/B switch: False routes the attention branch of every Swin block through the unfused LayerNorm / qkv / core / proj chain."""
    _STATE["fused_attn_block"] = bool(on)


def set_attention_fp8(on: bool) -> None:
    """BASELINE configuration 5: QK^T and PV of the Swin window attention FORWARD on fp8 (OCP e4m3) MFMA operands with per-(window, head)
    scales; everything else (and the whole backward) keeps bf16 operands.  Needs set_math('bf16')."""
    _STATE["attn_fp8"] = bool(on)


def attention_math() -> int:
    """math code of the window-attention forward call"""
    return hip.MATH_FP8 if (_STATE.get("attn_fp8") and _STATE["math"] == hip.MATH_BF16) else _STATE["math"]


def set_fused_mlp(on: bool) -> None:
    """A/B switch: False routes every Swin MLP through the unfused LayerNorm / fc1 / fc2 chain of the contraction engine."""
    _STATE["fused_mlp"] = bool(on)


def colsum(x, rows, cols, ld, out, accumulate=True):
    call("sv_colsum", ptr(x), rows, cols, ld, ptr(out), 1 if accumulate else 0)


# ---------------------------------------------------------------------------------------------------
# Linear helpers (rows x K) used by the Swin blocks
# ---------------------------------------------------------------------------------------------------
def linear_fwd(x, rows, spec: ConvSpec, w, out, **epi):
    spec.forward(x, rows, (1, 1, 1), w, out, **epi)


def linear_dgrad(dy, rows, spec: ConvSpec, w_t, dx, **epi):
    spec.dgrad(dy, rows, (1, 1, 1), w_t, dx, **epi)


def linear_wgrad(dy, x, rows, spec: ConvSpec, dw, db=None, async_ok=True):
    spec.wgrad(dy, x, rows, (1, 1, 1), dw, db=db, async_ok=async_ok)


# ---------------------------------------------------------------------------------------------------
# normalisation wrappers
# ---------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, rows, Cdim, merge_hw=(0, 0), eps=1e-5):
    y = empty(rows, Cdim, like=x)
    mean = fempty(rows, like=x)
    rstd = fempty(rows, like=x)
    call("sv_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, Cdim, eps, merge_hw[0], merge_hw[1])
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, Cdim, merge_hw=(0, 0), accumulate_dx=False):
    ws = zeros_f64(LN_BWD_SLOTS * Cdim + 1, dy.device)     # sv_layernorm_bwd_workspace_floats(C) floats, zero on entry
    call("sv_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), rows, Cdim,
         merge_hw[0], merge_hw[1], 1 if accumulate_dx else 0)


class BatchNormState:
    """Per-call state of one BatchNorm layer on channels-last [M, C] data."""

    def __init__(self, bn: torch.nn.Module, M: int, training: bool):
        self.bn, self.M, self.training, self.C = bn, M, training, bn.num_features
        dev = bn.weight.device
        self.sums = zeros_f64(BN_SLOTS * 2 * self.C, dev) if training else None   # [slot][2C] doubles
        buf = fempty(4 * self.C, device=dev)
        self.scale, self.shift, self.mean, self.rstd = buf[:self.C], buf[self.C:2 * self.C], buf[2 * self.C:3 * self.C], buf[3 * self.C:]

    def finalize(self):
        bn = self.bn
        if self.training and bn.num_batches_tracked is not None:
            if _CTX.bn_tick is None:
                _CTX.bn_tick = []
            _CTX.bn_tick.append(bn.num_batches_tracked)   # += 1 in one foreach launch (bn_tick_flush)
        mom = bn.momentum if bn.momentum is not None else 0.1
        call("sv_bn_finalize", ptr(self.sums), self.M, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
             float(mom), float(bn.eps), 1 if self.training else 0, ptr(self.scale), ptr(self.shift), ptr(self.mean), ptr(self.rstd), self.C)

    def probe(self, x, ldx):
        probe = _STATE.get("bn_probe")
        if probe is not None:        # debug hook (tests): the producer's statistics next to the output it stored
            probe(self, x, ldx)

    def apply(self, x, ldx, y, ldy, act=ACT_NONE, slope=0.0, residual=None, ldr=0):
        self.probe(x, ldx)
        self.signs = None
        if (residual is not None and act != ACT_NONE and self.training and _bn_signs_on() and self.C % 256 == 0
                and self.C % 4 == 0 and ldx % 4 == 0 and ldy % 4 == 0 and ldr % 4 == 0):
            # BatchNorm in front of a residual sum: the backward's activation mask as sign words (1/16 of the output's bytes) - see backward()
            self.signs = torch.empty(self.M * (self.C // 64), dtype=torch.int64, device=x.device)
            call("sv_scale_shift_act_signs", ptr(x), ldx, ptr(self.scale), ptr(self.shift), ptr(residual), ldr, ptr(y), ldy, self.M, self.C, act, slope,
                 ptr(self.signs))
            return
        call("sv_scale_shift_act", ptr(x), ldx, ptr(self.scale), ptr(self.shift), ptr(residual), ldr, ptr(y), ldy, self.M, self.C, act, slope)

    def backward(self, dz, lddz, z, ldz, x, ldx, dx, lddx, dgamma, dbeta, act=ACT_NONE, slope=0.0, dres=None, lddres=0):
        ws = zeros_f64((BN_BWD_SLOTS + 1) * 2 * self.C + 2, dz.device)     # sv_bn_bwd_workspace_doubles(C), zero on entry
        if getattr(self, "signs", None) is not None and dres is not None:
            # mask from the forward's sign words: the reduce pass reads (dz, x, words) and stores the masked gradient in dres, the apply pass
            # reads (dres, x): 6 passes over the tensor instead of the 8 of (dz, z, x) + (dz, z, x, dx, dres)
            call("sv_bn_bwd_signs", ptr(dz), lddz, ptr(self.signs), ptr(x), ldx, ptr(self.bn.weight), ptr(self.mean), ptr(self.rstd), self.M, self.C,
                 act, slope, 1 if self.training else 0, ptr(dx), lddx, ptr(dres), lddres, ptr(dgamma), ptr(dbeta), ptr(ws))
            return
        # z = None: the forward added no residual, so the activation mask is recomputed from x (one tensor read less per pass)
        call("sv_bn_bwd", ptr(dz), lddz, ptr(z), ldz, ptr(x), ldx, ptr(self.bn.weight), ptr(self.mean), ptr(self.rstd), self.M, self.C,
             act, slope, 1 if self.training else 0, ptr(dx), lddx, ptr(dres), lddres, ptr(dgamma), ptr(dbeta), ptr(ws),
             ptr(self.scale), ptr(self.shift))


def _bn_signs_on() -> bool:
    import os
    return _STATE.get("bn_signs", os.environ.get("SV_BN_SIGNS", "1") != "0")


def set_bn_signs(on: bool) -> None:
    """A/B switch: False keeps the stored output as the activation mask of the BatchNorms in front of a residual sum."""
    _STATE["bn_signs"] = bool(on)


def input_prep_enabled() -> bool:
    import os
    return _STATE.get("input_prep", os.environ.get("SV_INPUT_PREP", "1") != "0")


def set_input_prep(on: bool) -> None:
    """A/B switch of the encoder's input: True (default) = sv_encoder_prep (renderings -> stem space-to-depth image + Swin patch rows in one
    pass, patch embedding as a Linear(48, C)), False = cast + NCHW -> NHWC transpose + sv_stem_space_to_depth + the 4 x 4 / stride-4 convolution."""
    _STATE["input_prep"] = bool(on)


def bn_pool_fused_enabled() -> bool:
    import os
    return _STATE.get("bn_pool_fused", os.environ.get("SV_BN_POOL_FUSED", "1") != "0")


def set_bn_pool_fused(on: bool) -> None:
    """A/B switch of the BatchNorm -> activation -> max-pool chains (ResNet stem, the Refiner's three down-sampling layers): True (default) =
    one forward pass and the pool's backward inside the BatchNorm backward (sv_bn_act_maxpool*_fwd / sv_bn_maxpool*_bwd: neither the
    activation nor its gradient is stored), False = the separate passes."""
    _STATE["bn_pool_fused"] = bool(on)


def set_conv_halo(mode: int) -> None:
    """Halo-tile kernels behind the contraction engine (csrc/conv_halo.hip: 3 x 3 / 64 -> 64 and the ResNet stem): 0 off, 1 calls of
    >= 256 tiles (default), 2 every call of those shapes.  Process-wide; the library reads SV_CONV_HALO for its initial mode."""
    rc = hip.load().sv_set_conv_halo(int(mode))
    if rc != 0:
        raise RuntimeError(f"sv_set_conv_halo failed (rc={rc}): {hip.load().sv_last_error().decode()}")


def set_conv_halo_wgrad(mode: int) -> None:
    """Halo-tile weight gradient of the 3 x 3 / stride-1 convolutions with 64 k channels (csrc/conv_halo.hip) behind sv_conv_wgrad: 0 off,
    1 from 4 tiles per split on (default), 2 always.  Process-wide; SV_CONV_HALO_WGRAD sets the initial mode."""
    rc = hip.load().sv_set_conv_halo_wgrad(int(mode))
    if rc != 0:
        raise RuntimeError(f"sv_set_conv_halo_wgrad failed (rc={rc}): {hip.load().sv_last_error().decode()}")


def set_bn_probe(fn) -> None:
    """Debug hook of the tests: fn(state, stored_output, ld) is called for every BatchNorm layer right before its normalisation pass,
    when `state.sums` (the [BN_SLOTS][2C] double partial sums the producing kernel's epilogue accumulated) and the producer's stored
    output are both complete on the current stream.  None removes it.  Process-wide configuration."""
    _STATE["bn_probe"] = fn


def transpose(src, dst, batch, R, Cc, lds=None, ldd=None, sb=None, db=None):
    """dst[b][c][r] = src[b][r][c]; element type taken from the tensors (activations in the storage dtype, parameters fp32)"""
    lds = lds or Cc
    ldd = ldd or R
    assert src.dtype == dst.dtype
    call("sv_transpose", ptr(src), ptr(dst), batch, R, Cc, lds, ldd, sb if sb is not None else R * lds, db if db is not None else Cc * ldd,
         act=hip.BF16 if src.dtype == torch.bfloat16 else hip.F32)
