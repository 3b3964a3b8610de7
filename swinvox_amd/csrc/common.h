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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
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
__device__ __forceinline__ float uniform01(uint32_t seed, uint64_t idx) {
  return (hash_u32(seed, (uint32_t)idx, (uint32_t)(idx >> 32)) >> 8) * (1.0f / 16777216.0f);
}

}  // namespace sv
