// Shared device/host helpers for the SwinVox gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/swinvox_hip.h"

namespace sv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- activation element types ---------------------------------------------------------------------------------
// Activations (and activation gradients) live in HBM as fp32 (SV_F32) or bf16 (SV_BF16); arithmetic is always fp32.
// Kernels are templated on the storage element AT and touch memory only through these helpers.
template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }
__device__ __forceinline__ float4 ld4f(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4f(const __bf16* p) {
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]);
}
__device__ __forceinline__ void st4f(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4f(__bf16* p, float4 v) {
  bf16x4 b;
  b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4*>(p) = b;
}
// N = 4 or 8 consecutive activations <-> float registers (8 x bf16 = one 16-byte access)
template <int N, typename T> __device__ __forceinline__ void ldnf(const T* p, float (&v)[N]) {
  if constexpr (N == 8 && sizeof(T) == 2) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    const b8 b = *reinterpret_cast<const b8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)b[j];
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) { const float4 f = ld4f(p + 4 * q); v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w; }
  }
}
template <int N, typename T> __device__ __forceinline__ void stnf(T* p, const float (&v)[N]) {
  if constexpr (N == 8 && sizeof(T) == 2) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    b8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
    *reinterpret_cast<b8*>(p) = b;
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) st4f(p + 4 * q, make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
  }
}
// raw 4-element vectors (no conversion) for the MFMA operand loaders
template <typename T> struct V4;
template <> struct V4<float> {
  typedef float4 type;
  static __device__ __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ type make(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
};
template <> struct V4<__bf16> {
  typedef bf16x4 type;
  static __device__ __forceinline__ type zero() { type z; z[0] = z[1] = z[2] = z[3] = (__bf16)0.f; return z; }
  static __device__ __forceinline__ type load(const __bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static __device__ __forceinline__ type make(float a, float b, float c, float d) {
    type z; z[0] = (__bf16)a; z[1] = (__bf16)b; z[2] = (__bf16)c; z[3] = (__bf16)d; return z;
  }
};

// N-element raw vectors (N = 4: 8 or 16 bytes, N = 8: bf16 only, 16 bytes)
template <typename T, int N> struct VecN;
template <> struct VecN<float, 4> : V4<float> {};
template <> struct VecN<__bf16, 4> : V4<__bf16> {};
template <> struct VecN<__bf16, 8> {
  typedef bf16x8 type;
  static __device__ __forceinline__ type zero() { type z; for (int i = 0; i < 8; ++i) z[i] = (__bf16)0.f; return z; }
  static __device__ __forceinline__ type load(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
};

// host-side dispatch on the activation dtype of a call: BODY sees the element type as AT
#define SV_DISPATCH_ACT(act_dtype, ...)                  \
  do {                                                   \
    if ((act_dtype) == SV_BF16) { typedef __bf16 AT; __VA_ARGS__ } \
    else { typedef float AT; __VA_ARGS__ }               \
  } while (0)
#define SV_REQUIRE_ACT(act_dtype) SV_REQUIRE((act_dtype) == SV_F32 || (act_dtype) == SV_BF16, "bad activation dtype %d", (int)(act_dtype))

// error plumbing: every extern "C" entry returns 0 or a negative code and records a message
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define SV_REQUIRE(cond, ...)                      \
  do {                                             \
    if (!(cond)) {                                 \
      sv::set_error(__VA_ARGS__);                  \
      return SV_ERR_INVALID;                       \
    }                                              \
  } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// all-reduce over the 16 lanes of a DPP row with VALU moves (quad_perm xor 1, xor 2, row_half_mirror, row_mirror); the two steps across rows
// are v_permlane16_swap / v_permlane32_swap exchanges (VALU too: no LDS crossbar round trip as __shfl_xor = ds_bpermute takes).  Every lane
// gets the result.
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// v_permlane16_swap exchanges the odd 16-lane rows of its first register with the even rows of the second, v_permlane32_swap the upper
// half of the first with the lower half of the second: with the same value in both, the registers afterwards hold "mine" and "the
// partner's" (lane ^ 16 / lane ^ 32) in some order - enough for a commutative reduction (scripts/probes/permlane_swap_probe.hip prints the
// mapping).  Inline asm: through __builtin_amdgcn_permlane16_swap hipcc 7.2 propagates the copy it makes for the second operand across the
// instruction - which rewrites BOTH registers - and combines the first result with itself; s_nop 1 covers the VALU-write -> permlane-read
// hazard the compiler would have covered.
__device__ __forceinline__ void permlane16_pair(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void permlane32_pair(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
// one step of an all-reduce (sum) over the lanes that share lane % CW: v + the value of a lane O away.  O = 16 / 32: the exchanges above;
// O < 16: a DPP rotation of the 16-lane row (rotations by CW, 2 CW, ... 8 visit the whole coset, as the xor steps would)
template <int O> __device__ __forceinline__ float lane_step_add(float v) {
  if constexpr (O == 32) { float w = v; permlane32_pair(v, w); return v + w; }
  else if constexpr (O == 16) { float w = v; permlane16_pair(v, w); return v + w; }
  else return v + dpp_move<0x120 + O>(v);   // row_ror:O
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
  v = lane_step_add<16>(v); v = lane_step_add<32>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v)); v = fmaxf(v, dpp_move<0x4E>(v)); v = fmaxf(v, dpp_move<0x141>(v)); v = fmaxf(v, dpp_move<0x140>(v));
  { float w = v; permlane16_pair(v, w); v = fmaxf(v, w); }
  { float w = v; permlane32_pair(v, w); v = fmaxf(v, w); }
  return v;
}

// block-wide sum over 256 threads (4 waves); scratch must hold >= 4 floats; all threads get the result
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += scratch[i];
  return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// GELU of the bf16-MFMA paths: Phi(x) as the logistic of an odd quintic fitted to the normal CDF, sigma(x (c0 + c1 x^2 + c2 x^4)), x^2 clamped at 64:
// |GELU - GELU_erf| <= 2.9e-5, derivative <= 1.1e-4 (1/30 of the bf16 rounding unit of the stored value) - the form the fused MLP kernels use
// (swin_mlp.hip).  5 VALU + v_exp_f32 + v_rcp_f32 against ~30 for the A&S erf above with its IEEE division; the epilogues of fc1 (GELU) and of
// fc2's data gradient (GELU') carry 64 activations per lane and tile.
__device__ __forceinline__ float gelu_fast(float x) {
  const float x2 = fminf(x * x, 64.f);
  const float p = fmaf(x2, fmaf(x2, 9.975397e-04f, -1.0665937e-01f), -2.3012706e+00f);    // -log2(e) * (c0 + c1 x^2 + c2 x^4)
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * p));
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float x2 = fminf(x * x, 64.f);
  const float p = fmaf(x2, fmaf(x2, 9.975397e-04f, -1.0665937e-01f), -2.3012706e+00f);
  const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * p));
  const float qd = fmaf(x2, fmaf(x2, -3.45720919e-03f, 2.21791923e-01f), 1.59511919f);     // d/dx [x (c0 + c1 x^2 + c2 x^4)]
  return s * fmaf(x * (1.f - s), qd, 1.f);
}
// the same on PAIRS of values: the polynomial parts run as packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 do two lanes-worth per
// instruction), only min / exp / rcp stay per element: 17 instructions per pair instead of ~15 per element
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_sigma2(f32x2 x, f32x2& x2) {   // sigma(x (c0 + c1 x^2 + c2 x^4)), x2 = min(x^2, 64)
  x2 = x * x;
  x2[0] = fminf(x2[0], 64.f); x2[1] = fminf(x2[1], 64.f);
  const f32x2 p = __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, (f32x2)(9.975397e-04f), (f32x2)(-1.0665937e-01f)), (f32x2)(-2.3012706e+00f));
  const f32x2 xp = x * p;
  f32x2 e;
  e[0] = __builtin_amdgcn_exp2f(xp[0]); e[1] = __builtin_amdgcn_exp2f(xp[1]);
  e = e + (f32x2)(1.f);
  f32x2 s;
  s[0] = __builtin_amdgcn_rcpf(e[0]); s[1] = __builtin_amdgcn_rcpf(e[1]);
  return s;
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { f32x2 x2; return x * gelu_sigma2(x, x2); }
__device__ __forceinline__ void gelu_both_fast2(f32x2 x, f32x2& g, f32x2& dg) {
  f32x2 x2;
  const f32x2 s = gelu_sigma2(x, x2);
  const f32x2 qd = __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, (f32x2)(-3.45720919e-03f), (f32x2)(2.21791923e-01f)), (f32x2)(1.59511919f));
  g = x * s;
  dg = s * __builtin_elementwise_fma(x * ((f32x2)(1.f) - s), qd, (f32x2)(1.f));
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) { f32x2 g, dg; gelu_both_fast2(x, g, dg); return dg; }
template <bool FAST> __device__ __forceinline__ float gelu_t(float x) {
  if constexpr (FAST) return gelu_fast(x);
  else return gelu_erf(x);
}
template <bool FAST> __device__ __forceinline__ float gelu_grad_t(float x) {
  if constexpr (FAST) return gelu_grad_fast(x);
  else return gelu_erf_grad(x);
}
template <bool FAST> __device__ __forceinline__ float apply_act_t(float v, int act, float slope) {
  switch (act) {
    case SV_ACT_RELU: return v > 0.f ? v : 0.f;
    case SV_ACT_GELU: return gelu_t<FAST>(v);
    case SV_ACT_LRELU: return v > 0.f ? v : v * slope;
    default: return v;
  }
}
template <bool FAST> __device__ __forceinline__ float act_grad_t(float pre, int act, float slope) {
  switch (act) {
    case SV_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
    case SV_ACT_GELU: return gelu_grad_t<FAST>(pre);
    case SV_ACT_LRELU: return pre > 0.f ? 1.f : slope;
    default: return 1.f;
  }
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case SV_ACT_RELU: return v > 0.f ? v : 0.f;
    case SV_ACT_GELU: return gelu_erf(v);
    case SV_ACT_LRELU: return v > 0.f ? v : v * slope;
    default: return v;
  }
}
// derivative of act wrt its input, given the PRE-activation value (GELU) or the sign of pre/post (relu, lrelu)
__device__ __forceinline__ float act_grad(float pre, int act, float slope) {
  switch (act) {
    case SV_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
    case SV_ACT_GELU: return gelu_erf_grad(pre);
    case SV_ACT_LRELU: return pre > 0.f ? 1.f : slope;
    default: return 1.f;
  }
}

// counter-based RNG for dropout / drop-path masks: same (seed, index) -> same bit in fwd and bwd
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint32_t idx_lo, uint32_t idx_hi) {
  uint32_t x = seed ^ (idx_lo * 0x9E3779B1u) ^ (idx_hi * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  x += idx_lo; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// seed of a launch that may be replayed from a hipGraph: the scalar argument is frozen at capture time, so the caller can
// pass a device word (`epoch`, advanced once per replay) that is mixed in on the device - every replay draws new masks,
// the backward of the same replay sees the same word as its forward.  epoch == nullptr: the scalar seed alone.
__device__ __forceinline__ uint32_t eff_seed(uint32_t seed, const uint32_t* epoch) {
  return epoch ? seed + epoch[0] * 0x9E3779B1u : seed;
}
__device__ __forceinline__ float uniform01(uint32_t seed, uint64_t idx) {
  return (hash_u32(seed, (uint32_t)idx, (uint32_t)(idx >> 32)) >> 8) * (1.0f / 16777216.0f);
}

}  // namespace sv
