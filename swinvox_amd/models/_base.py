"""Shared machinery of the HIP-backed modules.

Each public module (Encoder, Decoder, Merger, Refiner) is ONE autograd node: its forward runs a hand-written
chain of HIP kernels and records a tape of device buffers, its backward walks the tape in reverse with the
matching gradient kernels and returns the parameter gradients.  Parameters live in stock nn.Conv*/nn.Linear/
nn.BatchNorm*/nn.LayerNorm holders so state_dict keys, init_weights (utils/helpers.py:20-44), optimizers,
clip_grad_norm_ and checkpoints behave exactly as with the reference modules.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from .. import hip, ops
from ..ops import ACT_LRELU, ACT_NONE, ACT_RELU, BatchNormState, ConvSpec, call, empty, ptr, zeros


class GradStore:
    """Zero-initialised gradient accumulators for every parameter of a module: views of one flat buffer in the FlatLayout
    of the parameter list (optim.py), so that a flat solver / the gradient all-reduce can take the buffer in place."""

    def __init__(self, params: Sequence[torch.Tensor], layout=None):
        from ..optim import FlatLayout
        self.params = list(params)
        self._map: Dict[int, torch.Tensor] = {}
        self.flat = None
        if self.params:
            layout = layout if layout is not None else FlatLayout(self.params)
            self.flat = torch.zeros(layout.total, dtype=torch.float32, device=self.params[0].device)
            for p, v in zip(self.params, layout.views(self.flat)):   # one C++ call cuts every parameter-shaped view
                self._map[id(p)] = v

    def __getitem__(self, p: torch.Tensor) -> torch.Tensor:
        return self._map[id(p)]

    def as_tuple(self):
        return tuple(self._map[id(p)] if p.requires_grad else None for p in self.params)


class _ModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, n_in, save, *args):
        inputs = args[:n_in]
        packs = mod.__dict__.setdefault("_packs", ops.PackCache())
        arena = mod.__dict__.setdefault("_arena_f", ops.ZeroArena())
        ops.set_pack_cache(packs)
        ops.set_arena(arena)
        try:
            packs.refresh()                      # every registered weight pack of the module in one launch
            arena.begin(inputs[0].device)        # every BatchNorm statistic accumulator of the pass from one fill
            outs, tape = mod._fwd(*inputs, save=save)
        finally:
            ops.bn_tick_flush()
            arena.end()
            ops.set_pack_cache(None)
            ops.set_arena(None)
        ctx.mod, ctx.tape, ctx.n_in = mod, tape, n_in
        ctx.store = ops.get_storage()
        ctx.in_needs = [isinstance(a, torch.Tensor) and a.requires_grad for a in inputs]
        return outs

    @staticmethod
    def backward(ctx, *douts):
        mod = ctx.mod
        if ctx.tape is None:
            raise RuntimeError("swinvox_amd: backward requested but the forward ran without a tape (no_grad?)")
        grads = GradStore(mod._param_list(), mod._grad_layout())
        if ops.get_storage() != ctx.store:
            raise RuntimeError("swinvox_amd: set_storage() changed between forward and backward")
        ops.set_pack_cache(mod.__dict__.setdefault("_packs", ops.PackCache()))
        arena = mod.__dict__.setdefault("_arena_b", ops.ZeroArena())
        ops.set_arena(arena)
        aw = ops.AsyncWgrad(grads.flat.device) if ops.overlap_enabled() else None
        ops.set_async_wgrad(aw)
        try:
            arena.begin(douts[0].device if douts[0] is not None else grads.flat.device)
            d_inputs = mod._bwd(ctx.tape, grads, ctx.in_needs, *douts)
        finally:
            ops.set_async_wgrad(None)
            if aw is not None:
                aw.join()                        # every parameter gradient is complete on the caller's stream
            arena.end()
            ops.set_pack_cache(None)
            ops.set_arena(None)
        ctx.tape = None
        return (None, None, None) + tuple(d_inputs) + grads.as_tuple()


class HipModule(nn.Module):
    """Base of the four drop-in modules.  Sub-classes implement _fwd(*inputs, save) -> (outputs, tape) and
    _bwd(tape, grads, in_needs, *douts) -> d_inputs."""

    stochastic = True  # dropout / drop-path active in train() (set False for deterministic gradient parity tests)
    # Data-parallel hand-off (dp.GradAllReducer): a module whose backward completes groups of parameter gradients early lists the
    # groups in grad_groups() (contiguous runs of the registration order) and calls _announce(grads, k) once group k is enqueued;
    # the hook then starts that group's all-reduce beside the rest of the backward.  None: nobody listens.
    grad_ready_hook = None

    def _announce(self, grads: "GradStore", group_index: int) -> None:
        hook = self.grad_ready_hook
        if hook is None:
            return
        views = [grads[p] for p in self._grad_groups_cached()[group_index] if p.requires_grad]
        if not views:
            return
        if not views[0].is_cuda:
            hook(group_index, views)
            return
        # the group's gradients were written by the current stream and by the weight-gradient stream: a staging stream waits
        # for both (neither of them is stalled) and the collective is enqueued behind it
        cur = torch.cuda.current_stream()
        cs = ops.comm_stream(views[0].device)
        cs.wait_stream(cur)
        aw = ops._CTX.awg
        if aw is not None:
            cs.wait_stream(aw.stream)
        with torch.cuda.stream(cs):
            hook(group_index, views)

    def _grad_groups_cached(self):
        """grad_groups() walks the parameter tree (~1 ms on the encoder, on the thread that enqueues the backward): cached next to
        the parameter list and dropped with it when the parameter count changes."""
        pl = self._param_list()
        gg = self.__dict__.get("_ggroups")
        if gg is None or gg[0] != len(pl):
            gg = self.__dict__["_ggroups"] = (len(pl), self.grad_groups())
        return gg[1]

    def _param_list(self) -> List[torch.Tensor]:
        """Parameters in registration order (cached: walking the module tree costs ~1 ms per call on the encoder).  The
        cache is checked against the parameter count so that added / removed parameters are picked up."""
        pl = self.__dict__.get("_plist")
        if pl is None or len(pl) != self.__dict__.get("_plist_n", -1):
            pl = list(self.parameters())
            self.__dict__["_plist"], self.__dict__["_plist_n"] = pl, len(pl)
        return pl

    def _grad_layout(self):
        from ..optim import FlatLayout
        pl = self._param_list()
        lay = self.__dict__.get("_glayout")
        if lay is None or not lay.matches(pl):
            lay = self.__dict__["_glayout"] = FlatLayout(pl)
        return lay

    def _run(self, *inputs):
        hip.check_cuda(*inputs)
        hip.check_cuda(*self._param_list()[:1])
        for t in inputs:
            # fp32, or the activation storage dtype (Decoder hands raw_features to Merger in it: no fp32 round trip of the largest tensor)
            if t.dtype != torch.float32 and t.dtype != ops._STATE["store"]:
                raise RuntimeError(f"swinvox_amd: expected float32 inputs, got {t.dtype}")
        params = self._param_list()
        save = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(t.requires_grad for t in inputs))
        # module inputs / outputs are fp32 (exception: raw_features, see Decoder._fwd); _fwd / _bwd convert to and from the storage dtype
        return _ModuleFn.apply(self, len(inputs), save, *inputs, *params)

    def _seed(self) -> int:
        return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


class ConvBnAct:
    """Conv/ConvTranspose (+bias) -> BatchNorm (train: batch statistics gathered in the contraction's epilogue)
    -> activation, on channels-last data; forward and backward."""

    def __init__(self, conv: nn.Module, bn: nn.Module, spec: ConvSpec, act: int, slope: float = 0.0):
        self.conv, self.bn, self.spec, self.act, self.slope = conv, bn, spec, act, slope

    def forward(self, x, n, in_grid, training, *, ldi=None, z=None, ldz=None, residual=None, ldr=0):
        sp = self.spec
        og = sp.out_grid(in_grid)
        M = n * og[0] * og[1] * og[2]
        wf = self._pack("f")
        y = empty(M, sp.cout, like=x)
        st = BatchNormState(self.bn, M, training)
        sp.forward(x, n, in_grid, wf, y, ldi=ldi, bias=self.conv.bias, stats=st.sums)
        st.finalize()
        if z is None:
            z, ldz = empty(M, sp.cout, like=x), sp.cout
        st.apply(y, sp.cout, z, ldz, self.act, self.slope, residual, ldr)
        # without a residual the backward recomputes the activation mask from y (scale*y + shift > 0) and never reads z
        return z, og, (x, ldi, y, z if residual is not None else None, ldz, st, n, in_grid, M)

    def backward(self, ctx, dz, lddz, grads, *, need_dx=True, dres=None, lddres=0, dx=None, lddx=None, dx_epi=None):
        x, ldi, y, z, ldz, st, n, in_grid, M = ctx
        sp = self.spec
        dy = (zeros if sp.cout_mem != sp.cout else empty)(M, sp.cout_mem, like=dz)
        st.backward(dz, lddz, z, ldz, y, sp.cout, dy, sp.cout_mem, grads[self.bn.weight], grads[self.bn.bias], self.act, self.slope,
                    dres, lddres)
        sp.wgrad(dy, x, n, in_grid, self._dw(grads), lddy=sp.cout_mem, ldx=ldi,
                 db=grads[self.conv.bias] if self.conv.bias is not None else None)
        if not need_dx:
            return None
        Min = n * in_grid[0] * in_grid[1] * in_grid[2]
        if dx is None:
            dx = (zeros if sp.cin_mem != sp.cin else empty)(Min, sp.cin_mem, like=dz)
            lddx = sp.cin_mem
        sp.dgrad(dy, n, in_grid, self._pack("d"), dx, lddy=sp.cout_mem, lddx=lddx, **(dx_epi or {}))
        return dx

    # ---- the layer followed by MaxPool3d(2): BatchNorm + activation + pool in one pass, the pool's backward inside the BatchNorm backward
    def forward_pool3d(self, x, n, in_grid, training):
        """-> (pooled [n * prod(og // 2), cout], og, ctx); the normalised activation is never stored (sv_bn_act_maxpool3d_fwd)"""
        sp = self.spec
        og = sp.out_grid(in_grid)
        M = n * og[0] * og[1] * og[2]
        y = empty(M, sp.cout, like=x)
        st = BatchNormState(self.bn, M, training)
        sp.forward(x, n, in_grid, self._pack("f"), y, bias=self.conv.bias, stats=st.sums)
        st.finalize()
        st.probe(y, sp.cout)
        pg = (og[0] // 2, og[1] // 2, og[2] // 2)
        p = empty(n * pg[0] * pg[1] * pg[2], sp.cout, like=x)
        idx = torch.empty(p.numel(), dtype=torch.uint8, device=x.device)
        call("sv_bn_act_maxpool3d_fwd", ptr(y), ptr(st.scale), ptr(st.shift), ptr(p), ptr(idx), n, og[0], og[1], og[2], sp.cout, self.act, self.slope)
        return p, og, (x, y, st, n, in_grid, og, M, idx)

    def backward_pool3d(self, ctx, dp, grads, *, need_dx=True):
        x, y, st, n, in_grid, og, M, idx = ctx
        sp = self.spec
        assert sp.cout_mem == sp.cout
        dy = empty(M, sp.cout, like=dp)
        ws = ops.zeros_f64((ops.BN_BWD_SLOTS + 1) * 2 * sp.cout + 2, dp.device)
        call("sv_bn_maxpool3d_bwd", ptr(dp), ptr(idx), ptr(y), ptr(self.bn.weight), ptr(st.mean), ptr(st.rstd), ptr(st.scale), ptr(st.shift),
             n, og[0], og[1], og[2], sp.cout, self.act, self.slope, 1 if st.training else 0, ptr(dy), ptr(grads[self.bn.weight]), ptr(grads[self.bn.bias]),
             ptr(ws))
        sp.wgrad(dy, x, n, in_grid, self._dw(grads), lddy=sp.cout, db=grads[self.conv.bias] if self.conv.bias is not None else None)
        if not need_dx:
            return None
        Min = n * in_grid[0] * in_grid[1] * in_grid[2]
        dx = (zeros if sp.cin_mem != sp.cin else empty)(Min, sp.cin_mem, like=dp)
        sp.dgrad(dy, n, in_grid, self._pack("d"), dx, lddy=sp.cout, lddx=sp.cin_mem)
        return dx

    # the two places a layer that runs on a re-indexed weight (Refiner head) overrides
    def _pack(self, kind):
        return self.spec.pack_fwd(self.conv.weight) if kind == "f" else self.spec.pack_dgrad(self.conv.weight)

    def _dw(self, grads):
        return grads[self.conv.weight]


def conv_spec_of(m: nn.Module, **kw) -> ConvSpec:
    """ConvSpec from a stock nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d / nn.Linear holder."""
    if isinstance(m, nn.Linear):
        return ConvSpec.linear(m.in_features, m.out_features)
    if isinstance(m, nn.Conv2d):
        assert m.groups == 1 and m.dilation == (1, 1)
        return ConvSpec.conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, **kw)
    if isinstance(m, nn.ConvTranspose3d):
        assert m.groups == 1 and m.output_padding == (0, 0, 0)
        return ConvSpec.conv3d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, transposed=True, **kw)
    if isinstance(m, nn.Conv3d):
        assert m.groups == 1
        return ConvSpec.conv3d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, **kw)
    raise TypeError(type(m))
