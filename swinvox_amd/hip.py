"""ctypes binding of libswinvox_hip.so (include/swinvox_hip.h).

The library is the product: there is NO eager/CPU fallback.  Importing this module never touches the
GPU (forked DataLoader workers stay safe, reference core/train.py:64-76); the first call that needs the
library raises RuntimeError if it was not built (python __graft_entry__.py build  /  make -C swinvox_amd/csrc).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SV_HIP_LIB") or os.path.join(_HERE, "libswinvox_hip.so")   # SV_HIP_LIB: A/B builds of the kernels

ACT_NONE, ACT_RELU, ACT_GELU, ACT_LRELU = 0, 1, 2, 3
MATH_F32, MATH_BF16, MATH_FP8 = 0, 1, 2   # MATH_FP8: window-attention forward only (QK^T / PV in OCP e4m3)
F32, BF16 = 0, 1      # SV_F32 / SV_BF16: storage element of the activation tensors of a call


class Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "Ci", "Co", "kd", "kh", "kw",
                                        "sd", "sh", "sw", "pd", "ph", "pw", "ldi")]


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int), ("row_scale", C.c_void_p),
                ("rows_per_scale", C.c_int), ("pre_act", C.c_void_p), ("stats", C.c_void_p), ("act", C.c_int),
                ("slope", C.c_float), ("act_grad_src", C.c_void_p), ("act_grad_kind", C.c_int), ("ldc", C.c_int),
                ("col_off", C.c_int)]


class PackDesc(C.Structure):   # sv_pack_desc
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("A", C.c_int), ("B", C.c_int), ("T", C.c_int), ("swap", C.c_int),
                ("rows_out", C.c_int), ("inner_out", C.c_int), ("block0", C.c_int), ("reserved", C.c_int)]


class AugSample(C.Structure):   # sv_aug_sample
    _fields_ = [("bg", C.c_float * 3), ("jitter_value", C.c_float * 3), ("jitter_order", C.c_int * 3), ("noise", C.c_float * 3),
                ("perm", C.c_int * 3), ("mean", C.c_float * 3), ("std", C.c_float * 3)]


# name -> (restype, argtypes); p = device/host pointer, i = int, l = long long, f = float, u = uint32, z = size_t
_P, _I, _L, _F, _U, _D = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_uint32, C.c_double
# Entry points whose activation tensors are void* + `int act_dtype` (inserted by call() right before the stream argument)
_ACT_TYPED = {
    "sv_conv_gather", "sv_tconv_gather", "sv_conv_wgrad", "sv_stencil3_fwd", "sv_stencil3_wgrad", "sv_colsum",
    "sv_layernorm_fwd", "sv_layernorm_bwd", "sv_ln_image_fwd", "sv_ln_image_bwd", "sv_bn_stats", "sv_scale_shift_act", "sv_bn_bwd", "sv_scale_shift_act_signs", "sv_bn_bwd_signs",
    "sv_window_attention_fwd", "sv_window_attention_bwd", "sv_cross_view_attention_fwd", "sv_cross_view_attention_bwd",
    "sv_transpose", "sv_add_n", "sv_axpby", "sv_relu_bwd", "sv_maxpool2d_fwd", "sv_maxpool2d_bwd", "sv_avgpool2_fwd", "sv_avgpool2_bwd",
    "sv_decoder_seed_fwd", "sv_decoder_seed_bwd", "sv_maxpool3d_fwd", "sv_maxpool3d_bwd", "sv_dropout", "sv_rowscale",
    "sv_dwconv2x2_fwd", "sv_dwconv2x2_bwd", "sv_upsample3to7_add_fwd", "sv_upsample3to7_bwd", "sv_decoder_head_fwd", "sv_decoder_head_bwd",
    "sv_merge_views_fwd", "sv_merge_views_bwd", "sv_stem_space_to_depth", "sv_encoder_prep", "sv_bn_act_maxpool_fwd", "sv_bn_maxpool_bwd", "sv_bn_act_maxpool3d_fwd", "sv_bn_maxpool3d_bwd", "sv_head_pack_x", "sv_head_unpack_dx", "sv_swin_attn_block_fwd", "sv_swin_attn_block_bwd",
}
# argument lists WITHOUT the act_dtype / stream tail (added in load())
_PROTOS = {
    "sv_version": (_I, None),
    "sv_conv_gather": (_I, [_P, _P, _P, C.POINTER(Geom), C.POINTER(Epilogue), _I]),
    "sv_tconv_gather": (_I, [_P, _P, _P, C.POINTER(Geom), C.POINTER(Epilogue), _I]),
    "sv_conv_gather_is_wide": (_I, None, [_P, _P, _P, C.POINTER(Geom), C.POINTER(Epilogue), _I, _I]),
    "sv_conv_wgrad_is_wide": (_I, None, [_P, _I, _P, C.POINTER(Geom), _I, _I, _I]),
    "sv_conv_wgrad_workspace_floats": (C.c_size_t, None, [C.POINTER(Geom)]),
    "sv_conv_wgrad": (_I, [_P, _I, _P, _P, C.POINTER(Geom), _I, _P, _P, _I]),
    "sv_stencil3_fwd": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P, _I, _P, _I, _I, _I, _I, _L, _L]),
    "sv_stencil3_wgrad_workspace_floats": (C.c_size_t, None, [_I, _I]),
    "sv_stencil3_wgrad": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _L]),
    "sv_tconv4s2_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "sv_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _I]),
    "sv_pack_weights_block_elems": (_I, None),
    "sv_pack_weights": (_I, [_P, _I, _I, _I]),
    "sv_colsum": (_I, [_P, _I, _I, _I, _P, _I]),
    "sv_cast": (_I, [_P, _I, _P, _I, _L]),
    "sv_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _I]),
    "sv_layernorm_bwd_workspace_floats": (C.c_size_t, None, [_I]),
    "sv_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I]),
    "sv_bn_bwd_workspace_doubles": (C.c_size_t, None, [_I]),
    "sv_ln_image_workspace_floats": (C.c_size_t, None, [_I, _I]),
    "sv_ln_image_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U, _P]),
    "sv_ln_image_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U, _P]),
    "sv_bn_stats": (_I, [_P, _L, _I, _I, _P]),
    "sv_bn_finalize": (_I, [_P, _L, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _I]),
    "sv_scale_shift_act": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _L, _I, _I, _F]),
    "sv_bn_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _L, _I, _I, _F, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P]),
    "sv_bn_signs_supported": (_I, None, [_I]),
    "sv_scale_shift_act_signs": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _L, _I, _I, _F, _P]),
    "sv_bn_bwd_signs": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _L, _I, _I, _F, _I, _P, _I, _P, _I, _P, _P, _P]),
    "sv_window_attention_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "sv_window_attention_bwd_workspace_floats": (C.c_size_t, None, [_I]),
    "sv_window_attention_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "sv_stem_space_to_depth": (_I, [_P, _P, _I]),
    "sv_bn_act_maxpool_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F]),
    "sv_bn_maxpool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "sv_bn_act_maxpool3d_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F]),
    "sv_bn_maxpool3d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "sv_set_conv_halo": (_I, None, [_I]),
    "sv_conv_halo_mode": (_I, None),
    "sv_set_conv_halo_wgrad": (_I, None, [_I]),
    "sv_conv_halo_launches": (_L, None),
    "sv_encoder_prep": (_I, [_P, _I, _P, _P, _I, _I]),
    "sv_head_pack_x": (_I, [_P, _P, _I, _I]),
    "sv_head_unpack_dx": (_I, [_P, _P, _I, _I]),
    "sv_stem_pack": (_I, [_P, _P, _I]),
    "sv_stem_unpack_grad": (_I, [_P, _P]),
    "sv_merger_pack": (_I, [_P, _P, _I, _I, _I, _I]),
    "sv_swin_mlp_supported": (_I, None, [_I]),
    "sv_swin_mlp_pack": (_I, [_P, _P, _P, _I]),
    "sv_swin_mlp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _F]),
    "sv_swin_mlp_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _F]),
    "sv_swin_mlp_wgrad": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _F]),
    "sv_swin_attn_block_supported": (_I, None, [_I, _I, _I, _I]),
    "sv_swin_attn_block_fwd": (_I, [_P] * 15 + [_I, _I, _I, _I, _I, _I, _F]),
    "sv_swin_attn_block_bwd": (_I, [_P] * 17 + [_I, _I, _I, _I, _I, _I]),
    "sv_cross_view_attention_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I]),
    "sv_cross_view_attention_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    "sv_transpose": (_I, [_P, _P, _I, _I, _I, _I, _I, _L, _L]),
    "sv_add_n": (_I, [_P, _P, _P, _P, _P, _L, _I, _I]),
    "sv_axpby": (_I, [_P, _P, _P, _F, _F, _L]),
    "sv_relu_bwd": (_I, [_P, _P, _P, _L]),
    "sv_maxpool2d_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I]),
    "sv_maxpool2d_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I]),
    "sv_avgpool2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I]),
    "sv_avgpool2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I]),
    "sv_decoder_seed_fwd": (_I, [_P, _P, _I, _I]),
    "sv_decoder_seed_bwd": (_I, [_P, _P, _I, _I]),
    "sv_maxpool3d_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    "sv_maxpool3d_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    "sv_dropout": (_I, [_P, _P, _L, _F, _U, _P]),
    "sv_droppath_scale": (_I, [_P, _I, _F, _U, _P]),
    "sv_rowscale": (_I, [_P, _P, _P, _L, _I, _I]),
    "sv_dwconv2x2_fwd": (_I, [_P, _P, _P, _P, _I, _I]),
    "sv_dwconv2x2_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I]),
    "sv_upsample3to7_add_fwd": (_I, [_P, _P, _I, _P, _I, _I]),
    "sv_upsample3to7_bwd": (_I, [_P, _P, _I, _I]),
    "sv_decoder_head_fwd": (_I, [_P, _P, _P, _P, _P, _L]),
    "sv_decoder_head_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L]),
    "sv_merge_views_fwd": (_I, [_P, _P, _P, _I, _I, _I]),
    "sv_merge_views_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I]),
    "sv_mean_views": (_I, [_P, _P, _I, _I, _I]),
    "sv_bce_logits": (_I, [_P, _P, _L, _P, _P, _P]),
    "sv_iou_counts": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sv_binvox_decode": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "sv_augment_views": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sv_grad_sumsq": (_I, [_P, _L, _F, _P]),
    "sv_adam_step": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _D, _L, _F, _P, _F, _P]),
    "sv_sgd_step": (_I, [_P, _P, _P, _L, _D, _D, _D, _L, _F, _P, _F, _P]),
}


def _argtypes(name):
    """Full ctypes argument list of an entry point: listed arguments (+ act_dtype) + stream; helper queries have neither."""
    spec = _PROTOS[name]
    if spec[1] is None:
        return list(spec[2]) if len(spec) > 2 else []
    return list(spec[1]) + ([_I] if name in _ACT_TYPED else []) + [_P]


def kernel_source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources + the C header: names the build a profile was taken on (bench.py only quotes PMC
    traffic from a committed profile whose hash equals the current tree's)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "swinvox_hip.h"))
    for f in files:
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


EXPORTED_SYMBOLS = sorted(list(_PROTOS.keys()) + ["sv_last_error"])

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """dlopen the HIP library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"swinvox_amd: {LIB_PATH} is missing - the HIP kernels are the product and there is no fallback. "
                    "Build with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950).")
            lib = C.CDLL(LIB_PATH)
            for name, spec in _PROTOS.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = spec[0], _argtypes(name)
            lib.sv_last_error.restype = C.c_char_p
            lib.sv_last_error.argtypes = []
            _lib = lib
    return _lib


class _Dev(threading.local):
    idx = None


_DEV = _Dev()   # per thread: device index of the tensors of the running module (set by check_cuda): the raw-stream query needs it


def _stream() -> int:
    """hipStream_t of torch's current stream (torch.cuda.current_stream() costs ~9 us per call; the raw query ~0.3 us)."""
    d = _DEV.idx
    if d is None:
        d = _DEV.idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(d)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be fp32 (or uint8 for pool indices)."""
    if t is None:
        return None
    return t.data_ptr()


def check_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None:
            if not t.is_cuda:
                raise RuntimeError("swinvox_amd: tensors must live on the GPU (no CPU path exists in the product); got "
                                   f"device={t.device}")
            _DEV.idx = t.device.index


class Tracer:
    """Times selected entry points with HIP events on the launch stream (bench.py's roofline leg).  Events are only
    recorded, never waited for, inside the timed region; summary() synchronises afterwards."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []          # (name, start_event, end_event, algorithmic_flops, algorithmic_bytes)
        self._tls = threading.local()    # the open bracket is per thread (autograd runs a module's backward on its own thread)

    @property
    def _open(self):
        return getattr(self._tls, "open", None)

    @_open.setter
    def _open(self, v):
        self._tls.open = v

    def begin(self, name, flops=0.0, nbytes=0.0, tag=""):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream())
        self._open = (name, ev0, ev1, flops, nbytes, tag)

    def end(self):
        ev1 = self._open[2]
        ev1.record(torch.cuda.current_stream())
        self.records.append(self._open)
        self._open = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, ev0, ev1, flops, nbytes, _ in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += ev0.elapsed_time(ev1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


    def detail(self):
        """per (entry point, geometry tag): launches, total ms, algorithmic TFLOP/s and GB/s - call after summary()"""
        out = {}
        for name, ev0, ev1, flops, nbytes, tag in self.records:
            d = out.setdefault((name, tag), [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += ev0.elapsed_time(ev1); d[2] += flops; d[3] += nbytes
        return out


TRACE = None   # set to a Tracer by bench.py
ACT = F32      # storage dtype code appended to act-typed calls (ops.storage() scopes set it)


def call(name: str, *args, act=None) -> None:
    """Invoke an entry point on the current torch stream; a non-zero return becomes a RuntimeError.  Entry points with
    activation tensors get the current storage dtype code (hip.ACT, or `act` when the caller overrides it)."""
    lib = load()
    if name in _ACT_TYPED:
        args = args + (ACT if act is None else act,)
    tr = TRACE
    if tr is not None and tr._open is None and name in tr.names:   # untimed-by-caller entry point selected for tracing
        tr.begin(name)
        rc = getattr(lib, name)(*args, _stream())
        tr.end()
    else:
        rc = getattr(lib, name)(*args, _stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.sv_last_error().decode()}")
