// Normalisation kernels (HBM-bound): token LayerNorm (+ fused PatchMerging gather), whole-image
// LayerNorm([C,H,W]) of the Swin stage heads, BatchNorm statistics / apply / backward on channels-last data.
#include "common.h"
#include <stdlib.h>

namespace sv {

// ------------------------------------------------------------------------------------------------
// token LayerNorm.  A row is handled by LPR lanes (16, 32 or 64) holding NV chunks of VEC elements each in registers
// (two-pass variance), so a wave normalises 64/LPR rows at once and narrow rows (C = 96 ... 384) keep 75 % of the lanes
// busy; VEC = 8 (16-byte accesses) with bf16 storage, 4 otherwise.
// MERGE: the row is the PatchMerging gather of 4 tokens (order h0w0,h1w0,h0w1,h1w1) of a [I,H,W,C0] map.
// ------------------------------------------------------------------------------------------------
// Contention-free reductions across workgroups: partial sums go (atomically) into one of NSLOT accumulator images chosen by
// the workgroup index and a tiny second kernel folds the images.  (A "last workgroup folds" ticket would need a device-scope
// __threadfence(), which writes back the whole per-XCD L2 on this chip - measured 100+ us per call.)
constexpr int BN_BWD_SLOTS = 16, LN_BWD_SLOTS = 32;

struct MergeMap { int H, W, C0; };  // H,W of the un-merged map

template <bool MERGE>
__device__ __forceinline__ size_t ln_src_off(long long row, int e, int C, const MergeMap& mm) {   // e = first element of the chunk
  if constexpr (!MERGE) return (size_t)row * C + e;
  const int blk = e / mm.C0, c = e - blk * mm.C0;   // blk = wp*2 + hp
  const int wp = blk >> 1, hp = blk & 1;
  const int W2 = mm.W >> 1, H2 = mm.H >> 1;
  const int x2 = (int)(row % W2); long long t = row / W2;
  const int y2 = (int)(t % H2); const long long i = t / H2;
  return ((size_t)(i * mm.H + 2 * y2 + hp) * mm.W + (2 * x2 + wp)) * mm.C0 + c;
}

template <int VEC, typename AT> struct LnIO;
template <typename AT> struct LnIO<4, AT> {
  static __device__ __forceinline__ void load(const AT* p, float* v) { const float4 q = ld4f(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  static __device__ __forceinline__ void store(AT* p, const float* v) { st4f(p, make_float4(v[0], v[1], v[2], v[3])); }
};
template <> struct LnIO<8, __bf16> {
  static __device__ __forceinline__ void load(const __bf16* p, float* v) {
    const bf16x8 q = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)q[j];
  }
  static __device__ __forceinline__ void store(__bf16* p, const float* v) {
    bf16x8 q;
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = q;
  }
};
__device__ __forceinline__ void ldp(const float* p, float* v, int n) {   // n = 4 or 8 fp32 parameters
  const float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  if (n == 8) { const float4 b = *reinterpret_cast<const float4*>(p + 4); v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
}
template <int CTRL> __device__ __forceinline__ float ln_dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {   // sum over the LPR (16 / 32 / 64) lanes that share a row, every lane gets it
  // inside a 16-lane DPP row: quad_perm xor 1, xor 2, row_half_mirror, row_mirror (VALU moves, no LDS crossbar round trips)
  v += ln_dpp_mov<0xB1>(v); v += ln_dpp_mov<0x4E>(v); v += ln_dpp_mov<0x141>(v); v += ln_dpp_mov<0x140>(v);
  if constexpr (LPR > 16) v = lane_step_add<16>(v);     // across rows: v_permlane16 / 32 swaps (common.h), VALU as well
  if constexpr (LPR > 32) v = lane_step_add<32>(v);
  return v;
}

template <bool MERGE, typename AT, int VEC, int LPR, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const AT* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, AT* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long long rows, int C, float eps, MergeMap mm) {
  constexpr int RPW = 64 / LPR;                       // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane % LPR, gr = lane / LPR;
  const long long row = ((long long)blockIdx.x * 4 + wave) * RPW + gr;
  const bool rok = row < rows;
  const int nchunk = C / VEC;
  float v[NV][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = gl + LPR * i;
    if (rok && ch < nchunk) {
      LnIO<VEC, AT>::load(x + ln_src_off<MERGE>(row, ch * VEC, C, mm), v[i]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += v[i][j];
    }
  }
  const float invC = 1.f / C;                                   // one division per thread, not two per row
  const float mean = group_sum<LPR>(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = gl + LPR * i;
    if (rok && ch < nchunk) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float a = v[i][j] - mean; q += a * a; }
    }
  }
  const float rstd = rsqrtf(group_sum<LPR>(q) * invC + eps);
  if (rok && gl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = gl + LPR * i;
    if (rok && ch < nchunk) {
      float g[VEC], b[VEC], o[VEC];
      ldp(gamma + ch * VEC, g, VEC); ldp(beta + ch * VEC, b, VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      LnIO<VEC, AT>::store(y + (size_t)row * C + ch * VEC, o);
    }
  }
}

// backward: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma; dgamma += dy*xhat, dbeta += dy.
// A workgroup walks rows_per_block rows (each row group a strided subset) with the per-column partial sums in registers;
// they are folded through one LDS image (LDS atomics) and reach HBM as one atomic per column per workgroup.
template <bool MERGE, typename AT, int VEC, int LPR, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const AT* __restrict__ dy, const AT* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, AT* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     long long rows, int C, MergeMap mm, int accumulate_dx, int rows_per_block,
                                                     float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2*C]
  constexpr int RPW = 64 / LPR, RPI = 4 * RPW;   // rows per wave / per workgroup iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane % LPR, gr = lane / LPR;
  const int nchunk = C / VEC;
  for (int c = threadIdx.x; c < 2 * C; c += 256) red[c] = 0.f;
  float dg[NV][VEC], db[NV][VEC], gm[NV][VEC];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = gl + LPR * i;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; gm[i][j] = 0.f; }
    if (ch < nchunk) ldp(gamma + ch * VEC, gm[i], VEC);
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const float invC = 1.f / C;
  // The rows of the NEXT iteration are fetched before this iteration's stores: vmcnt retires in issue order, so loads issued behind a store
  // can only be waited for by draining the store - one store round trip per iteration otherwise.
  float xv[NV][VEC], dv[NV][VEC], ov[NV][VEC], mean = 0.f, rstd = 0.f;
  bool rok = false;
  auto fetch = [&](int rr, float (&xo)[NV][VEC], float (&d_)[NV][VEC], float (&old)[NV][VEC], float& mn, float& rs, bool& ok) {
    const long long row = r0 + rr;
    ok = rr < rows_per_block && row < rows;
    mn = ok ? mean_in[row] : 0.f; rs = ok ? rstd_in[row] : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int ch = gl + LPR * i;
#pragma unroll
      for (int j = 0; j < VEC; ++j) { xo[i][j] = 0.f; d_[i][j] = 0.f; old[i][j] = 0.f; }
      if (ok && ch < nchunk) {
        LnIO<VEC, AT>::load(x + ln_src_off<MERGE>(row, ch * VEC, C, mm), xo[i]);
        LnIO<VEC, AT>::load(dy + (size_t)row * C + ch * VEC, d_[i]);
        if (accumulate_dx) LnIO<VEC, AT>::load(dx + ln_src_off<MERGE>(row, ch * VEC, C, mm), old[i]);
      }
    }
  };
  fetch(wave * RPW + gr, xv, dv, ov, mean, rstd, rok);
  for (int rr = wave * RPW + gr; rr < rows_per_block; rr += RPI) {   // whole row groups stay in the loop: the group reductions need every lane
    const long long row = r0 + rr;
    float xn[NV][VEC], dn[NV][VEC], on[NV][VEC], mean_n, rstd_n;
    bool rok_n;
    fetch(rr + RPI, xn, dn, on, mean_n, rstd_n, rok_n);
    float xh[NV][VEC], g[NV][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int ch = gl + LPR * i;
      if (rok && ch < nchunk) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          xh[i][j] = (xv[i][j] - mean) * rstd;
          g[i][j] = dv[i][j] * gm[i][j];
          s1 += g[i][j]; s2 += g[i][j] * xh[i][j];
          dg[i][j] += dv[i][j] * xh[i][j]; db[i][j] += dv[i][j];
        }
      }
    }
    const float m1 = group_sum<LPR>(s1) * invC, m2 = group_sum<LPR>(s2) * invC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int ch = gl + LPR * i;
      if (rok && ch < nchunk) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = rstd * (g[i][j] - m1 - xh[i][j] * m2) + ov[i][j];
        LnIO<VEC, AT>::store(dx + ln_src_off<MERGE>(row, ch * VEC, C, mm), o);
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) { xv[i][j] = xn[i][j]; dv[i][j] = dn[i][j]; ov[i][j] = on[i][j]; }
    mean = mean_n; rstd = rstd_n; rok = rok_n;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int ch = gl + LPR * i;
    if (ch < nchunk) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { atomicAdd(red + ch * VEC + j, dg[i][j]); atomicAdd(red + C + ch * VEC + j, db[i][j]); }
    }
  }
  __syncthreads();
  // one atomic per column per workgroup into one of LN_BWD_SLOTS images; the last workgroup folds them into dgamma / dbeta
  float* slot = ws + (size_t)(blockIdx.x % LN_BWD_SLOTS) * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256) atomicAdd(slot + c, red[c]);
}
// dgamma[c] += sum_slot ws[slot][c], dbeta[c] += sum_slot ws[slot][C + c]
__global__ __launch_bounds__(256) void ln_bwd_fold_kernel(const float* __restrict__ ws, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * C) return;
  float a = 0.f;
#pragma unroll 8
  for (int sl = 0; sl < LN_BWD_SLOTS; ++sl) a += ws[(size_t)sl * 2 * C + c];
  if (c < C) dgamma[c] += a; else dbeta[c - C] += a;
}

// (LPR, NV) for a row of `chunks` VEC-element chunks; NV is rounded up to an instantiated count
static inline bool ln_shape(int chunks, int& lpr, int& nv) {
  lpr = chunks <= 16 ? 16 : (chunks <= 32 ? 32 : 64);
  const int need = (chunks + lpr - 1) / lpr;
  static const int opts[] = {1, 2, 3, 4, 6, 12};
  for (int o : opts) if (need <= o) { nv = o; return true; }
  return false;
}
#define SV_LN_DISPATCH(LPR_, NV_, ...)                                                       \
  do {                                                                                       \
    if (LPR_ == 16) { constexpr int LPR = 16, NV = 1; __VA_ARGS__ }                          \
    else if (LPR_ == 32) { constexpr int LPR = 32, NV = 1; __VA_ARGS__ }                     \
    else if (NV_ == 1) { constexpr int LPR = 64, NV = 1; __VA_ARGS__ }                       \
    else if (NV_ == 2) { constexpr int LPR = 64, NV = 2; __VA_ARGS__ }                       \
    else if (NV_ == 3) { constexpr int LPR = 64, NV = 3; __VA_ARGS__ }                       \
    else if (NV_ == 4) { constexpr int LPR = 64, NV = 4; __VA_ARGS__ }                       \
    else if (NV_ == 6) { constexpr int LPR = 64, NV = 6; __VA_ARGS__ }                       \
    else { constexpr int LPR = 64, NV = 12; __VA_ARGS__ }                                    \
  } while (0)

// ------------------------------------------------------------------------------------------------
// long-row LayerNorm with a full-size affine (Swin stage heads: nn.LayerNorm([C,H,W]) on NHWC data whose
// affine parameters were transposed to [HW,C] once per step).  L = H*W*C up to 301,056 per image.
// Pass 1: every workgroup reduces a CHUNK of one image to (mean_chunk, M2_chunk); pass 2 merges the
// chunk moments (Chan) and applies the affine (+ optional dropout mask).
// ------------------------------------------------------------------------------------------------
constexpr int LNL_CHUNK = 4096;  // elements per workgroup in pass 1 (16 per thread)

template <typename AT>
__global__ __launch_bounds__(256) void lnl_moments_kernel(const AT* __restrict__ x, float* __restrict__ part, int L, int nchunks) {
  __shared__ float sc[4];
  const int img = blockIdx.y, ck = blockIdx.x;
  const int e0 = ck * LNL_CHUNK;
  int n = L - e0; if (n > LNL_CHUNK) n = LNL_CHUNK;
  const AT* src = x + (size_t)img * L + e0;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (threadIdx.x + 256 * i) * 4;
    if (e < n) { v[i] = ld4f(src + e); s += v[i].x + v[i].y + v[i].z + v[i].w; }
  }
  const float mean = block_sum<4>(s, sc) / n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (threadIdx.x + 256 * i) * 4;
    if (e < n) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
  const float m2 = block_sum<4>(q, sc);
  if (threadIdx.x == 0) {
    part[((size_t)img * nchunks + ck) * 2 + 0] = mean;
    part[((size_t)img * nchunks + ck) * 2 + 1] = m2;
  }
}

__global__ void lnl_finalize_kernel(const float* __restrict__ part, float* __restrict__ meanrstd, int L, int nchunks, float eps, int I) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= I) return;
  float n_a = 0.f, mean_a = 0.f, m2_a = 0.f;
  for (int c = 0; c < nchunks; ++c) {
    int nb = L - c * LNL_CHUNK; if (nb > LNL_CHUNK) nb = LNL_CHUNK;
    const float mean_b = part[((size_t)img * nchunks + c) * 2], m2_b = part[((size_t)img * nchunks + c) * 2 + 1];
    const float n = n_a + nb, delta = mean_b - mean_a;
    mean_a += delta * nb / n;
    m2_a += m2_b + delta * delta * n_a * nb / n;
    n_a = n;
  }
  meanrstd[img * 2] = mean_a;
  meanrstd[img * 2 + 1] = rsqrtf(m2_a / L + eps);
}

template <typename AT>
__global__ __launch_bounds__(256) void lnl_apply_kernel(const AT* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ meanrstd,
                                                        AT* __restrict__ y, int L, float drop_p, uint32_t seed, const uint32_t* __restrict__ epoch) {
  seed = eff_seed(seed, epoch);
  const int img = blockIdx.y;
  const float mean = meanrstd[img * 2], rstd = meanrstd[img * 2 + 1];
  const float keep_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int e = (blockIdx.x * 256 + threadIdx.x) * 4; e < L; e += gridDim.x * 1024) {
    const float4 xv = ld4f(x + (size_t)img * L + e);
    const float4 wv = *reinterpret_cast<const float4*>(w + e), bv = *reinterpret_cast<const float4*>(b + e);
    float o[4] = {(xv.x - mean) * rstd * wv.x + bv.x, (xv.y - mean) * rstd * wv.y + bv.y,
                  (xv.z - mean) * rstd * wv.z + bv.z, (xv.w - mean) * rstd * wv.w + bv.w};
    if (drop_p > 0.f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = uniform01(seed, (uint64_t)img * L + e + j) < drop_p ? 0.f : o[j] * keep_inv;
    }
    st4f(y + (size_t)img * L + e, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// backward pass 1: per-image sums of g and g*xhat (g = dy*mask*w) -> sums[I][2] (atomics)
template <typename AT>
__global__ __launch_bounds__(256) void lnl_bwd_reduce_kernel(const AT* __restrict__ dy, const AT* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ meanrstd,
                                                             double* __restrict__ sums, int L, float drop_p, uint32_t seed, const uint32_t* __restrict__ epoch) {
  seed = eff_seed(seed, epoch);
  __shared__ float sc[4];
  const int img = blockIdx.y;
  const float mean = meanrstd[img * 2], rstd = meanrstd[img * 2 + 1];
  const float keep_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float s1 = 0.f, s2 = 0.f;
  for (int e = (blockIdx.x * 256 + threadIdx.x) * 4; e < L; e += gridDim.x * 1024) {
    const float4 dv = ld4f(dy + (size_t)img * L + e);
    const float4 xv = ld4f(x + (size_t)img * L + e);
    const float4 wv = *reinterpret_cast<const float4*>(w + e);
    const float d[4] = {dv.x, dv.y, dv.z, dv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w}, ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float dd = d[j];
      if (drop_p > 0.f) dd = uniform01(seed, (uint64_t)img * L + e + j) < drop_p ? 0.f : dd * keep_inv;
      const float g = dd * ww[j];
      s1 += g; s2 += g * (xx[j] - mean) * rstd;
    }
  }
  s1 = block_sum<4>(s1, sc);
  s2 = block_sum<4>(s2, sc);
  if (threadIdx.x == 0) { atomicAdd(sums + img * 2, (double)s1); atomicAdd(sums + img * 2 + 1, (double)s2); }
}

// backward pass 2: dx per element; dw/db accumulated over the images by the thread that owns elements e..e+3 (L % 4 == 0).
// The images are split over gridDim.y (each slice ends in one float atomic per element): with a single slice the launch is
// L / 1024 = 294 workgroups at most, each thread walking all I images serially - 1.4 TB/s on the stage-0 head.
template <typename AT>
__global__ __launch_bounds__(256) void lnl_bwd_apply_kernel(const AT* __restrict__ dy, const AT* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ meanrstd,
                                                            const double* __restrict__ sums, AT* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, int L, int I,
                                                            float drop_p, uint32_t seed, const uint32_t* __restrict__ epoch) {
  seed = eff_seed(seed, epoch);
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= L) return;
  const float keep_inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const float4 wq = *reinterpret_cast<const float4*>(w + e);
  const float wv[4] = {wq.x, wq.y, wq.z, wq.w};
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  const int per = (I + gridDim.y - 1) / gridDim.y, i0 = blockIdx.y * per;
  const int i1 = i0 + per < I ? i0 + per : I;
  for (int img = i0; img < i1; ++img) {
    const float mean = meanrstd[img * 2], rstd = meanrstd[img * 2 + 1];
    const double m1 = sums[img * 2] / L, m2 = sums[img * 2 + 1] / L;
    const float4 dq = ld4f(dy + (size_t)img * L + e), xq = ld4f(x + (size_t)img * L + e);
    float dd[4] = {dq.x, dq.y, dq.z, dq.w};
    const float xx[4] = {xq.x, xq.y, xq.z, xq.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (drop_p > 0.f) dd[j] = uniform01(seed, (uint64_t)img * L + e + j) < drop_p ? 0.f : dd[j] * keep_inv;
      const float xh = (xx[j] - mean) * rstd;
      const float g = dd[j] * wv[j];
      o[j] = (float)((double)rstd * ((double)g - m1 - (double)xh * m2));
      aw[j] += dd[j] * xh; ab[j] += dd[j];
    }
    st4f(dx + (size_t)img * L + e, make_float4(o[0], o[1], o[2], o[3]));
  }
  if (gridDim.y == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { dw[e + j] += aw[j]; db[e + j] += ab[j]; }
  } else if (i0 < i1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { atomicAdd(dw + e + j, aw[j]); atomicAdd(db + e + j, ab[j]); }
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm on channels-last [M, C] (row stride ld)
// ------------------------------------------------------------------------------------------------
// per-channel sum and sum of squares (when the producing contraction did not accumulate them itself)
template <typename AT>
__global__ __launch_bounds__(256) void bn_stats_kernel(const AT* __restrict__ x, long long M, int C, int ld,
                                                       double* __restrict__ sums, long long rows_per_block) {
  __shared__ double r1[4][64], r2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1e = r0 + rows_per_block; if (r1e > M) r1e = M;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) for (long long r = r0 + rl; r < r1e; r += 4) { const double v = ldf(x + (size_t)r * ld + c); s1 += v; s2 += v * v; }
  r1[rl][threadIdx.x & 63] = s1; r2[rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    atomicAdd(sums + c, r1[0][threadIdx.x] + r1[1][threadIdx.x] + r1[2][threadIdx.x] + r1[3][threadIdx.x]);
    atomicAdd(sums + C + c, r2[0][threadIdx.x] + r2[1][threadIdx.x] + r2[2][threadIdx.x] + r2[3][threadIdx.x]);
  }
}

// train: batch mean / biased var -> scale, shift, saved mean/rstd, running-stat update (unbiased var).
// eval : scale/shift from the running statistics.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps, int training,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {   // statistics in double: E[x^2]-mean^2 cancels badly in fp32 (torch's CPU path accumulates in double too)
    double t1 = 0.0, t2 = 0.0;
    for (int sl = 0; sl < SV_BN_SLOTS; ++sl) { t1 += sums[(size_t)sl * 2 * C + c]; t2 += sums[(size_t)sl * 2 * C + C + c]; }
    const double m = t1 / count;
    double v = t2 / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m; var = (float)v;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(v * (count > 1.0 ? count / (count - 1.0) : 1.0));
  } else {
    mean = running_mean[c]; var = running_var[c];
  }
  const float rstd = rsqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc; shift[c] = beta[c] - mean * sc;
  save_mean[c] = mean; save_rstd[c] = rstd;
}

// y = act(x*scale[c] + shift[c] (+ residual)): scalar fallback (any C / strides); the vector paths are the *_cg and *_narrow kernels below
template <typename AT>
__global__ __launch_bounds__(256) void scale_shift_act_scalar_kernel(const AT* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, const AT* __restrict__ res,
                                                                     int ldr, AT* __restrict__ y, int ldy, long long M, int C,
                                                                     int act, float slope) {
  const long long total = M * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    float o = __fmaf_rn(ldf(x + (size_t)r * ldx + c), scale[c], shift[c]);
    if (res) o += ldf(res + (size_t)r * ldr + c);
    stf(y + (size_t)r * ldy + c, apply_act(o, act, slope));
  }
}

// fold the BatchNorm backward slots: fin[c] = sum_slot ws[slot][c] (c < 2C), dgamma += fin[C + c], dbeta += fin[c]
__global__ __launch_bounds__(256) void bn_bwd_fold_kernel(double* __restrict__ ws, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * C) return;
  double a = 0.0;
#pragma unroll
  for (int sl = 0; sl < BN_BWD_SLOTS; ++sl) a += ws[(size_t)sl * 2 * C + c];
  ws[(size_t)BN_BWD_SLOTS * 2 * C + c] = a;
  if (c < C) dbeta[c] += (float)a; else dgamma[c - C] += (float)a;
}

// backward pass 1: per channel s1 = sum(dz'), s2 = sum(dz' * xhat), dz' = dz * act'(z) (mask from the OUTPUT z)
template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const AT* __restrict__ dz, int lddz, const AT* __restrict__ z, int ldz,
                                                            const AT* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, long long M, int C, int act, float slope,
                                                            double* __restrict__ sums, long long rows_per_block, int CW,
                                                            const float* __restrict__ fsc, const float* __restrict__ fsh) {
  // CW (a power of two <= 64) lanes across the channels, 256 / CW row lanes: with one or two channels every thread still works
  __shared__ double r1[256], r2[256];
  const int cl = threadIdx.x & (CW - 1), c = blockIdx.x * CW + cl, rl = threadIdx.x / CW, RL = 256 / CW;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1e = r0 + rows_per_block; if (r1e > M) r1e = M;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c];
    const float msc = z ? 0.f : fsc[c], msh = z ? 0.f : fsh[c];     // no saved output: the mask is recomputed from x
    const float neg = act == SV_ACT_LRELU ? slope : 0.f;
    long long r = r0 + rl;
    for (; r + 3 * RL < r1e; r += 4 * RL) {   // 4 rows in flight per thread (independent loads), then the serial tail
      float d[4], xv[4], zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = ldf(dz + (size_t)(r + RL * u) * lddz + c);
        xv[u] = ldf(x + (size_t)(r + RL * u) * ldx + c);
        zv[u] = act != SV_ACT_NONE ? (z ? ldf(z + (size_t)(r + RL * u) * ldz + c) : __fmaf_rn(xv[u], msc, msh)) : 1.f;
      }
      float f1 = 0.f, f2 = 0.f;               // four rows in fp32, then into the double accumulators
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (act != SV_ACT_NONE) d[u] *= (zv[u] > 0.f) ? 1.f : neg;
        f1 += d[u]; f2 += d[u] * (xv[u] - mu) * rs;
      }
      s1 += (double)f1; s2 += (double)f2;
    }
    for (; r < r1e; r += RL) {
      float d = ldf(dz + (size_t)r * lddz + c);
      const float xq = ldf(x + (size_t)r * ldx + c);
      if (act != SV_ACT_NONE) d *= ((z ? ldf(z + (size_t)r * ldz + c) : __fmaf_rn(xq, msc, msh)) > 0.f) ? 1.f : neg;
      s1 += (double)d; s2 += (double)(d * (xq - mu) * rs);
    }
  }
  r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    double a = 0.0, b = 0.0;
    for (int l = 0; l < RL; ++l) { a += r1[l * CW + cl]; b += r2[l * CW + cl]; }
    double* slot = sums + (size_t)(blockIdx.y % BN_BWD_SLOTS) * 2 * C;
    atomicAdd(slot + c, a);
    atomicAdd(slot + C + c, b);
  }
}

// backward pass 2: dx = gamma*rstd*(dz' - s1/M - xhat*s2/M) (train) or dz'*gamma*rstd (eval); dres = dz' (optional);
// block 0 also folds the sums into dgamma/dbeta
template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const AT* __restrict__ dz, int lddz, const AT* __restrict__ z, int ldz,
                                                           const AT* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const double* __restrict__ sums, long long M, int C, int act, float slope,
                                                           int training, AT* __restrict__ dx, int lddx, AT* __restrict__ dres,
                                                           int lddres, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           const float* __restrict__ fsc, const float* __restrict__ fsh) {
  const long long total = M * C;
  const double invM = 1.0 / (double)M;
  sums += (size_t)BN_BWD_SLOTS * 2 * C;     // the folded image written by the reduce pass
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C; const int c = (int)(i - r * C);
    float d = ldf(dz + (size_t)r * lddz + c);
    const float xq = ldf(x + (size_t)r * ldx + c);
    if (act != SV_ACT_NONE) d *= ((z ? ldf(z + (size_t)r * ldz + c) : __fmaf_rn(xq, fsc[c], fsh[c])) > 0.f) ? 1.f : (act == SV_ACT_LRELU ? slope : 0.f);
    if (dres) stf(dres + (size_t)r * lddres + c, d);
    const float rs = rstd[c];
    float o;
    if (training) {
      // double arithmetic for the mean-subtraction terms: sum_r dx must vanish to rounding, otherwise the residual is
      // amplified by every later reduction over positions (torch's CPU kernel evaluates this expression in double too)
      const double xh = ((double)xq - (double)mean[c]) * (double)rs;
      o = (float)((double)gamma[c] * (double)rs * ((double)d - sums[c] * invM - xh * (sums[C + c] * invM)));
    } else {
      o = d * gamma[c] * rs;
    }
    stf(dx + (size_t)r * lddx + c, o);
  }
}

// ------------------------------------------------------------------------------------------------
// narrow BatchNorm (C <= 16, every row stride a multiple of 4: the 9/12-channel merger tail): a thread owns ONE 4-channel
// group of a row (g = tid & 3, 64 rows per workgroup pass), so a wave instruction moves 8/16-byte vectors that tile whole
// rows (21 rows x 24 B with 9 channels) instead of one 2-byte element per lane in the generic scalar kernels, and the
// per-channel constants of the group sit in registers.
// ------------------------------------------------------------------------------------------------
constexpr int BN_NARROW_MAXC = 16;

template <typename AT>
__global__ __launch_bounds__(256) void scale_shift_act_narrow_kernel(const AT* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, const AT* __restrict__ res, int ldr,
                                                                     AT* __restrict__ y, int ldy, long long M, int C, int act, float slope) {
  const int g = threadIdx.x & 3, rl = threadIdx.x >> 2;
  if (4 * g >= C) return;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int c = 4 * g + j; sc[j] = c < C ? scale[c] : 0.f; sh[j] = c < C ? shift[c] : 0.f; }
  for (long long r = (long long)blockIdx.x * 64 + rl; r < M; r += (long long)gridDim.x * 64) {
    const float4 xv = ld4f(x + (size_t)r * ldx + 4 * g);
    float o[4] = {xv.x, xv.y, xv.z, xv.w};
    float rr[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) { const float4 rv = ld4f(res + (size_t)r * ldr + 4 * g); rr[0] = rv.x; rr[1] = rv.y; rr[2] = rv.z; rr[3] = rv.w; }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (4 * g + j < C) ? apply_act(__fmaf_rn(o[j], sc[j], sh[j]) + rr[j], act, slope) : 0.f;
    st4f(y + (size_t)r * ldy + 4 * g, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// pass 1: per-thread fp32 partials over <= rows_per_thread rows, workgroup fold in double, one atomic per channel per workgroup
template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_narrow_kernel(const AT* __restrict__ dz, int lddz, const AT* __restrict__ z, int ldz,
                                                                   const AT* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, long long M, int C, int act, float slope,
                                                                   double* __restrict__ sums, int rows_per_thread,
                                                                   const float* __restrict__ fsc, const float* __restrict__ fsh) {
  __shared__ double red[64][9];     // [row lane][4 x s1, 4 x s2] (+1 pad) per group, folded group by group
  const int g = threadIdx.x & 3, rl = threadIdx.x >> 2;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, mu[4], rs[4], msc[4], msh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * g + j;
    mu[j] = c < C ? mean[c] : 0.f; rs[j] = c < C ? rstd[c] : 0.f;
    msc[j] = (!z && fsc && c < C) ? fsc[c] : 0.f; msh[j] = (!z && fsh && c < C) ? fsh[c] : 0.f;
  }
  const float neg = act == SV_ACT_LRELU ? slope : 0.f;
  if (4 * g < C) {
    const long long r0 = (long long)blockIdx.x * 64 * rows_per_thread + rl;
    for (int k = 0; k < rows_per_thread; ++k) {
      const long long r = r0 + (long long)k * 64;
      if (r >= M) break;
      const float4 dv = ld4f(dz + (size_t)r * lddz + 4 * g), xv = ld4f(x + (size_t)r * ldx + 4 * g);
      float d[4] = {dv.x, dv.y, dv.z, dv.w};
      const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
      if (act != SV_ACT_NONE) {
        float zz[4];
        if (z) { const float4 zv = ld4f(z + (size_t)r * ldz + 4 * g); zz[0] = zv.x; zz[1] = zv.y; zz[2] = zv.z; zz[3] = zv.w; }
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) zz[j] = __fmaf_rn(xx[j], msc[j], msh[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] *= zz[j] > 0.f ? 1.f : neg;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] += d[j]; s2[j] += d[j] * (xx[j] - mu[j]) * rs[j]; }
    }
  }
  for (int gg = 0; gg < 4; ++gg) {           // fold the 64 row lanes of one channel group at a time
    __syncthreads();
    if (g == gg) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[rl][j] = (double)s1[j]; red[rl][4 + j] = (double)s2[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const int c = 4 * gg + (threadIdx.x & 3);
      if (c < C) {
        double a = 0.0;
        for (int l = 0; l < 64; ++l) a += red[l][threadIdx.x];
        atomicAdd(sums + (size_t)(blockIdx.x % BN_BWD_SLOTS) * 2 * C + (threadIdx.x >> 2) * C + c, a);
      }
    }
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_apply_narrow_kernel(const AT* __restrict__ dz, int lddz, const AT* __restrict__ z, int ldz,
                                                                  const AT* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const double* __restrict__ sums, long long M, int C, int act, float slope,
                                                                  int training, AT* __restrict__ dx, int lddx, AT* __restrict__ dres, int lddres,
                                                                  const float* __restrict__ fsc, const float* __restrict__ fsh) {
  sums += (size_t)BN_BWD_SLOTS * 2 * C;     // the folded image
  const int g = threadIdx.x & 3, rl = threadIdx.x >> 2;
  if (4 * g >= C) return;
  float msc[4], msh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int c = 4 * g + j; msc[j] = (!z && fsc && c < C) ? fsc[c] : 0.f; msh[j] = (!z && fsh && c < C) ? fsh[c] : 0.f; }
  // dx = k1*d - k2 - k3*x with per-channel constants (double: the mean-removal terms must cancel to rounding)
  double k1[4], k2[4], k3[4];
  const double invM = 1.0 / (double)M;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * g + j;
    if (c < C) {
      const double gm = gamma[c], rs = rstd[c], mu = mean[c];
      k1[j] = gm * rs;
      if (training) { const double a = sums[c] * invM, b = sums[C + c] * invM; k3[j] = gm * rs * b * rs; k2[j] = gm * rs * a - k3[j] * mu; }
      else { k2[j] = 0.0; k3[j] = 0.0; }
    } else { k1[j] = k2[j] = k3[j] = 0.0; }
  }
  const float neg = act == SV_ACT_LRELU ? slope : 0.f;
  for (long long r = (long long)blockIdx.x * 64 + rl; r < M; r += (long long)gridDim.x * 64) {
    const float4 dv = ld4f(dz + (size_t)r * lddz + 4 * g), xv = ld4f(x + (size_t)r * ldx + 4 * g);
    float d[4] = {dv.x, dv.y, dv.z, dv.w};
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
    if (act != SV_ACT_NONE) {
      float zz[4];
      if (z) { const float4 zv = ld4f(z + (size_t)r * ldz + 4 * g); zz[0] = zv.x; zz[1] = zv.y; zz[2] = zv.z; zz[3] = zv.w; }
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) zz[j] = __fmaf_rn(xx[j], msc[j], msh[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] *= zz[j] > 0.f ? 1.f : neg;
    }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // padding columns: whatever was read (possibly uninitialised memory) is ignored, zero is written
      const bool real = 4 * g + j < C;
      o[j] = real ? (float)(k1[j] * (double)d[j] - k2[j] - k3[j] * (double)xx[j]) : 0.f;
      if (!real) d[j] = 0.f;
    }
    if (dres) st4f(dres + (size_t)r * lddres + 4 * g, make_float4(d[0], d[1], d[2], d[3]));
    st4f(dx + (size_t)r * lddx + 4 * g, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// vector paths of BatchNorm apply / backward apply (C and every row stride multiples of 4): a thread keeps ONE 4-channel group
// (gq = tid % G) and walks rows (rl = tid / G, step RL = 256 / G), so the per-channel constants live in registers and the loop
// has no 64-bit division (a flat element loop spends as many issue slots on index math and constant reloads as on the data).
template <typename AT>
__global__ __launch_bounds__(256) void scale_shift_act_cg_kernel(const AT* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const AT* __restrict__ res, int ldr,
                                                                 AT* __restrict__ y, int ldy, long long M, int C, int act, float slope,
                                                                 long long rows_per_block, int G, unsigned long long* __restrict__ signs) {
  // signs (optional; G == 64 and C % 256 == 0, so that a wave is the 64 column groups of ONE row): bit `lane` of word
  // signs[(row * (C / 256) + blockIdx.x) * 4 + j] = (value of channel 256 blockIdx.x + 4 lane + j before the activation) > 0 - the
  // activation mask the backward needs, 1/16 of the bytes of the output it would otherwise re-read
  const int gq = threadIdx.x % G, rl = threadIdx.x / G, RL = 256 / G;
  const int c = (blockIdx.x * G + gq) * 4;
  if (c >= C || rl >= RL) return;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
#pragma unroll 2
  for (long long r = r0 + rl; r < r1; r += RL) {
    const float4 xv = ld4f(x + (size_t)r * ldx + c);
    float o[4] = {__fmaf_rn(xv.x, sc.x, sh.x), __fmaf_rn(xv.y, sc.y, sh.y), __fmaf_rn(xv.z, sc.z, sh.z), __fmaf_rn(xv.w, sc.w, sh.w)};
    if (res) {
      const float4 rv = ld4f(res + (size_t)r * ldr + c);
      o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
    }
    if (signs) {
      const unsigned long long b0 = __builtin_amdgcn_ballot_w64(o[0] > 0.f), b1 = __builtin_amdgcn_ballot_w64(o[1] > 0.f);
      const unsigned long long b2 = __builtin_amdgcn_ballot_w64(o[2] > 0.f), b3 = __builtin_amdgcn_ballot_w64(o[3] > 0.f);
      if (gq == 0) {
        unsigned long long* w = signs + ((size_t)r * (C >> 8) + blockIdx.x) * 4;
        *reinterpret_cast<ulonglong2*>(w) = make_ulonglong2(b0, b1);
        *reinterpret_cast<ulonglong2*>(w + 2) = make_ulonglong2(b2, b3);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = apply_act(o[j], act, slope);
    st4f(y + (size_t)r * ldy + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_apply_cg_kernel(const AT* __restrict__ dz, int lddz, const AT* __restrict__ z, int ldz,
                                                              const AT* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const double* __restrict__ sums, long long M, int C, int act, float slope,
                                                              int training, AT* __restrict__ dx, int lddx, AT* __restrict__ dres, int lddres,
                                                              const float* __restrict__ fsc, const float* __restrict__ fsh,
                                                              long long rows_per_block, int G) {
  const int gq = threadIdx.x % G, rl = threadIdx.x / G, RL = 256 / G;
  const int c = (blockIdx.x * G + gq) * 4;
  if (c >= C || rl >= RL) return;
  sums += (size_t)BN_BWD_SLOTS * 2 * C;     // the folded image written by the reduce pass
  // dx = k1*d - k2 - k3*x (double: the mean-removal terms must cancel to rounding), see bn_bwd_apply_narrow_kernel
  double k1[4], k2[4], k3[4];
  const double invM = 1.0 / (double)M;
  float msc[4] = {0.f, 0.f, 0.f, 0.f}, msh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double gm = gamma[c + j], rs = rstd[c + j], mu = mean[c + j];
    k1[j] = gm * rs;
    if (training) { const double a = sums[c + j] * invM, b = sums[C + c + j] * invM; k3[j] = gm * rs * b * rs; k2[j] = gm * rs * a - k3[j] * mu; }
    else { k2[j] = 0.0; k3[j] = 0.0; }
    if (!z && act != SV_ACT_NONE) { msc[j] = fsc[c + j]; msh[j] = fsh[c + j]; }
  }
  const float neg = act == SV_ACT_LRELU ? slope : 0.f;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
#pragma unroll 2
  for (long long r = r0 + rl; r < r1; r += RL) {
    const float4 dv = ld4f(dz + (size_t)r * lddz + c), xv = ld4f(x + (size_t)r * ldx + c);
    float d[4] = {dv.x, dv.y, dv.z, dv.w};
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
    if (act != SV_ACT_NONE) {
      float zz[4];
      if (z) { const float4 zv = ld4f(z + (size_t)r * ldz + c); zz[0] = zv.x; zz[1] = zv.y; zz[2] = zv.z; zz[3] = zv.w; }
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) zz[j] = __fmaf_rn(xx[j], msc[j], msh[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] *= zz[j] > 0.f ? 1.f : neg;
    }
    if (dres) st4f(dres + (size_t)r * lddres + c, make_float4(d[0], d[1], d[2], d[3]));
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (float)(k1[j] * (double)d[j] - k2[j] - k3[j] * (double)xx[j]);
    st4f(dx + (size_t)r * lddx + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// vector variant of pass 1 (C and the row strides multiples of 4): a thread owns 4 adjacent channels of a strided row
// subset; G = min(C/4, 64) column groups x 256/G row lanes per workgroup, row lanes folded through LDS
template <typename AT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const AT* __restrict__ dz, int lddz, const AT* __restrict__ z, int ldz,
                                                                const AT* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, long long M, int C, int act, float slope,
                                                                double* __restrict__ sums, long long rows_per_block, int G,
                                                                const float* __restrict__ fsc, const float* __restrict__ fsh,
                                                                AT* __restrict__ dmask, int lddm, const unsigned long long* __restrict__ signs) {
  // signs (optional, instead of z; G == 64, C % 256 == 0): the forward's activation mask, one bit per element (scale_shift_act_cg_kernel)
  // dmask (optional): receives dz' = dz * act'(z) - the gradient of the residual branch AND what pass 2 then reads instead of (dz, z):
  // a BatchNorm in front of a residual sum (bn3 / the down-sampling branch of every bottleneck) saves one read of the widest tensors
  __shared__ double red[256][9];   // [thread][8 partials] (+1 pad)
  const int gq = threadIdx.x % G, rl = threadIdx.x / G, RL = 256 / G;
  const int c = (blockIdx.x * G + gq) * 4;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1e = r0 + rows_per_block; if (r1e > M) r1e = M;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < C && rl < RL) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;          // no saved output: the mask is recomputed from x
    if (!z && !signs && act != SV_ACT_NONE) { msc = *reinterpret_cast<const float4*>(fsc + c); msh = *reinterpret_cast<const float4*>(fsh + c); }
    const float neg = act == SV_ACT_LRELU ? slope : 0.f;
    long long r = r0 + rl;
    for (; r + RL < r1e; r += 2 * RL) {   // two rows in flight
      const float4 d0 = ld4f(dz + (size_t)r * lddz + c), d1 = ld4f(dz + (size_t)(r + RL) * lddz + c);
      const float4 x0 = ld4f(x + (size_t)r * ldx + c), x1 = ld4f(x + (size_t)(r + RL) * ldx + c);
      float a0[4] = {d0.x, d0.y, d0.z, d0.w}, a1[4] = {d1.x, d1.y, d1.z, d1.w};
      if (act != SV_ACT_NONE && signs) {
        const ulonglong2* w0 = reinterpret_cast<const ulonglong2*>(signs + ((size_t)r * (C >> 8) + blockIdx.x) * 4);
        const ulonglong2* w1 = reinterpret_cast<const ulonglong2*>(signs + ((size_t)(r + RL) * (C >> 8) + blockIdx.x) * 4);
        const ulonglong2 p0 = w0[0], p1 = w0[1], q0 = w1[0], q1 = w1[1];       // the same four words for the whole wave (one row each)
        const unsigned long long m0[4] = {p0.x, p0.y, p1.x, p1.y}, m1[4] = {q0.x, q0.y, q1.x, q1.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[j] *= ((m0[j] >> gq) & 1ull) ? 1.f : neg; a1[j] *= ((m1[j] >> gq) & 1ull) ? 1.f : neg; }
      } else if (act != SV_ACT_NONE) {
        float4 z0, z1;
        if (z) { z0 = ld4f(z + (size_t)r * ldz + c); z1 = ld4f(z + (size_t)(r + RL) * ldz + c); }
        else {
          z0 = make_float4(__fmaf_rn(x0.x, msc.x, msh.x), __fmaf_rn(x0.y, msc.y, msh.y), __fmaf_rn(x0.z, msc.z, msh.z), __fmaf_rn(x0.w, msc.w, msh.w));
          z1 = make_float4(__fmaf_rn(x1.x, msc.x, msh.x), __fmaf_rn(x1.y, msc.y, msh.y), __fmaf_rn(x1.z, msc.z, msh.z), __fmaf_rn(x1.w, msc.w, msh.w));
        }
        const float q0[4] = {z0.x, z0.y, z0.z, z0.w}, q1[4] = {z1.x, z1.y, z1.z, z1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { a0[j] *= q0[j] > 0.f ? 1.f : neg; a1[j] *= q1[j] > 0.f ? 1.f : neg; }
      }
      if (dmask) {
        st4f(dmask + (size_t)r * lddm + c, make_float4(a0[0], a0[1], a0[2], a0[3]));
        st4f(dmask + (size_t)(r + RL) * lddm + c, make_float4(a1[0], a1[1], a1[2], a1[3]));
      }
      const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
      const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += (double)a0[j] + (double)a1[j];
        s2[j] += (double)(a0[j] * (xa[j] - mm[j]) * rr[j]) + (double)(a1[j] * (xb[j] - mm[j]) * rr[j]);
      }
    }
    for (; r < r1e; r += RL) {
      const float4 d0 = ld4f(dz + (size_t)r * lddz + c), x0 = ld4f(x + (size_t)r * ldx + c);
      float a0[4] = {d0.x, d0.y, d0.z, d0.w};
      if (act != SV_ACT_NONE && signs) {
        const ulonglong2* w0 = reinterpret_cast<const ulonglong2*>(signs + ((size_t)r * (C >> 8) + blockIdx.x) * 4);
        const ulonglong2 p0 = w0[0], p1 = w0[1];
        const unsigned long long m0[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[j] *= ((m0[j] >> gq) & 1ull) ? 1.f : neg;
      } else if (act != SV_ACT_NONE) {
        const float4 z0 = z ? ld4f(z + (size_t)r * ldz + c)
                            : make_float4(__fmaf_rn(x0.x, msc.x, msh.x), __fmaf_rn(x0.y, msc.y, msh.y), __fmaf_rn(x0.z, msc.z, msh.z), __fmaf_rn(x0.w, msc.w, msh.w));
        const float q0[4] = {z0.x, z0.y, z0.z, z0.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[j] *= q0[j] > 0.f ? 1.f : neg;
      }
      if (dmask) st4f(dmask + (size_t)r * lddm + c, make_float4(a0[0], a0[1], a0[2], a0[3]));
      const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, mm[4] = {mu.x, mu.y, mu.z, mu.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] += (double)a0[j]; s2[j] += (double)(a0[j] * (xa[j] - mm[j]) * rr[j]); }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][4 + j] = s2[j]; }
  __syncthreads();
  // thread t < 8*G folds partial (t % 8) of column group (t / 8) over the row lanes
  for (int t = threadIdx.x; t < 8 * G; t += 256) {
    const int q = t >> 3, k = t & 7;
    const int cc = (blockIdx.x * G + q) * 4 + (k & 3);
    if (cc < C) {
      double a = 0.0;
      for (int l = 0; l < RL; ++l) a += red[l * G + q][k];
      atomicAdd(sums + (size_t)(blockIdx.y % BN_BWD_SLOTS) * 2 * C + (k < 4 ? 0 : C) + cc, a);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// ResNet stem: BatchNorm + ReLU + MaxPool2d(3, stride 2, padding 1) (reference models/encoder.py:22-23: torchvision resnet50's bn1 / relu /
// maxpool) without the 112 x 112 x 64 activation in between - at 512 images that tensor is 822 MB, written by the normalisation pass, read by
// the pool, written again (as its gradient) by the pool's backward and read twice by the BatchNorm backward.
//   forward : pooled[o] = max over the window of relu(scale * y + shift) (each value rounded to the storage type first, as the separate pass
//             stored it), arg-max tap kept (first maximum in scan order, as torch) - reads y once, writes the 56 x 56 map + 1 byte per element;
//   backward: dz[i] = sum over the <= 2 x 2 windows that contain i of dpooled[o] * [argmax(o) == i] (the pool's gather form), computed on the
//             fly in BOTH passes of the BatchNorm backward (reduce: sums of dz' and dz' * xhat; apply: dy = k1 dz' - k2 - k3 y): neither dz nor
//             the activation is ever stored.
// A thread owns 4 adjacent channels; a workgroup walks whole image rows (no per-position divisions).
// ------------------------------------------------------------------------------------------------
// CV = channels per thread: 8 with bf16 storage (16-byte accesses), else 4
template <int CV> struct TapWord;
template <> struct TapWord<4> { typedef unsigned type; };
template <> struct TapWord<8> { typedef unsigned long long type; };
template <int CV> __device__ __forceinline__ void ld_taps(const uint8_t* p, int (&t)[CV]) {
  const typename TapWord<CV>::type w = *reinterpret_cast<const typename TapWord<CV>::type*>(p);
#pragma unroll
  for (int j = 0; j < CV; ++j) t[j] = (int)((w >> (8 * j)) & 0xff);
}
template <typename AT, int CV>
__global__ __launch_bounds__(256) void bn_act_maxpool_fwd_kernel(const AT* __restrict__ x, const float* __restrict__ fsc, const float* __restrict__ fsh,
                                                                 AT* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo,
                                                                 int act, float slope) {
  const int cv = C / CV;
  const long long total = (long long)N * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * CV; long long t = i / cv;
    const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); const int n = (int)(t / Ho);
    float s_[CV], h_[CV], best[CV];
    int bi[CV];
#pragma unroll
    for (int j = 0; j < CV; ++j) { s_[j] = fsc[c + j]; h_[j] = fsh[c + j]; best[j] = -3.4e38f; bi[j] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          float v[CV];
          ldnf<CV>(x + (((size_t)n * H + ih) * W + iw) * C + c, v);
#pragma unroll
          for (int j = 0; j < CV; ++j) {
            const float z = (float)(AT)apply_act(__fmaf_rn(v[j], s_[j], h_[j]), act, slope);   // what scale_shift_act would have stored
            if (z > best[j]) { best[j] = z; bi[j] = kh * 3 + kw; }
          }
        }
      }
    const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c;
    stnf<CV>(y + o, best);
    typename TapWord<CV>::type w = 0;
#pragma unroll
    for (int j = 0; j < CV; ++j) w |= (typename TapWord<CV>::type)bi[j] << (8 * j);
    *reinterpret_cast<typename TapWord<CV>::type*>(idx + o) = w;
  }
}

// The pool's data gradient for the 2 x 2 input block (2a .. 2a + 1, 2b .. 2b + 1), channels c .. c + CV - 1: the block lies in the windows (a, b),
// (a, b + 1), (a + 1, b), (a + 1, b + 1) only, so four pooled gradients + their arg-max taps (tap = 3 kh + kw, input = 2 o - 1 + k) serve four
// positions - and all four loads go out together (the per-position gather form of maxpool2d_bwd_kernel re-reads each pooled element 2.25
// times behind data-dependent loop bounds: 1.57 ms for the two passes against 1.51 for the separate kernels).  d[p][j]: p = 2 * dy + dx.
template <typename AT, int CV>
__device__ __forceinline__ void pool_block_grad(const AT* __restrict__ dmp, const uint8_t* __restrict__ idx, int n, int a, int b, int c, int C, int Ho, int Wo,
                                                float (&d)[4][CV]) {
  const size_t o00 = (((size_t)n * Ho + a) * Wo + b) * C + c;
  const bool vb = b + 1 < Wo, va = a + 1 < Ho;
  const size_t o01 = vb ? o00 + C : o00, o10 = va ? o00 + (size_t)Wo * C : o00, o11 = (va && vb) ? o00 + (size_t)Wo * C + C : o00;
  float g0[CV], g1[CV], g2[CV], g3[CV];
  int i0[CV], i1[CV], i2[CV], i3[CV];
  ldnf<CV>(dmp + o00, g0); ldnf<CV>(dmp + o01, g1); ldnf<CV>(dmp + o10, g2); ldnf<CV>(dmp + o11, g3);
  ld_taps<CV>(idx + o00, i0); ld_taps<CV>(idx + o01, i1); ld_taps<CV>(idx + o10, i2); ld_taps<CV>(idx + o11, i3);
#pragma unroll
  for (int j = 0; j < CV; ++j) {
    const int t1 = vb ? i1[j] : 255, t2 = va ? i2[j] : 255, t3 = (va && vb) ? i3[j] : 255;
    d[0][j] = i0[j] == 4 ? g0[j] : 0.f;
    d[1][j] = (i0[j] == 5 ? g0[j] : 0.f) + (t1 == 3 ? g1[j] : 0.f);
    d[2][j] = (i0[j] == 7 ? g0[j] : 0.f) + (t2 == 1 ? g2[j] : 0.f);
    d[3][j] = (i0[j] == 8 ? g0[j] : 0.f) + (t1 == 6 ? g1[j] : 0.f) + (t2 == 2 ? g2[j] : 0.f) + (t3 == 0 ? g3[j] : 0.f);
  }
}

// pass 1: G = C / CV column groups x 256 / G block lanes; blockIdx.x owns image-row PAIRS [blockIdx.x * rpb, +rpb) of the N * Ho pairs
template <typename AT, int CV>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const AT* __restrict__ dmp, const uint8_t* __restrict__ idx, const AT* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ fsc, const float* __restrict__ fsh, double* __restrict__ sums,
                                                                 int N, int H, int W, int C, int Ho, int Wo, int act, float slope, int rpb) {
  __shared__ double red[256][2 * CV + 1];
  const int G = C / CV, gq = threadIdx.x % G, rl = threadIdx.x / G, RL = 256 / G, c = gq * CV;
  double s1[CV], s2[CV];
#pragma unroll
  for (int j = 0; j < CV; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
  if (rl < RL) {
    float mm[CV], rr[CV], s_[CV], h_[CV];
#pragma unroll
    for (int j = 0; j < CV; ++j) { mm[j] = mean[c + j]; rr[j] = rstd[c + j]; s_[j] = fsc[c + j]; h_[j] = fsh[c + j]; }
    const float neg = act == SV_ACT_LRELU ? slope : 0.f;
    const int rp0 = blockIdx.x * rpb, rp1 = min(rp0 + rpb, N * Ho);
    for (int rp = rp0; rp < rp1; ++rp) {
      const int n = rp / Ho, a = rp - n * Ho;
      float t1[CV], t2[CV];      // one row pair per lane in fp32 (<= 4 Wo / RL terms), then double
#pragma unroll
      for (int j = 0; j < CV; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
      for (int b = rl; b < Wo; b += RL) {
        float d[4][CV];
        pool_block_grad<AT, CV>(dmp, idx, n, a, b, c, C, Ho, Wo, d);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int ih = 2 * a + (p >> 1), iw = 2 * b + (p & 1);
          if (ih < H && iw < W) {
            float xx[CV];
            ldnf<CV>(x + (((size_t)n * H + ih) * W + iw) * C + c, xx);
#pragma unroll
            for (int j = 0; j < CV; ++j) {
              float dd = d[p][j];
              if (act != SV_ACT_NONE) dd *= __fmaf_rn(xx[j], s_[j], h_[j]) > 0.f ? 1.f : neg;
              t1[j] += dd; t2[j] += dd * (xx[j] - mm[j]) * rr[j];
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < CV; ++j) { s1[j] += (double)t1[j]; s2[j] += (double)t2[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < CV; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][CV + j] = s2[j]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * CV * G; t += 256) {
    const int q = t / (2 * CV), k = t % (2 * CV);
    double acc = 0.0;
    for (int l = 0; l < RL; ++l) acc += red[l * G + q][k];
    atomicAdd(sums + (size_t)(blockIdx.x % BN_BWD_SLOTS) * 2 * C + (k < CV ? 0 : C) + q * CV + (k % CV), acc);
  }
}

// pass 2: dx = k1 dz' - k2 - k3 x (double coefficients: the mean-removal terms must cancel to rounding), dz' gathered again
template <typename AT, int CV>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const AT* __restrict__ dmp, const uint8_t* __restrict__ idx, const AT* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ fsc, const float* __restrict__ fsh, const double* __restrict__ sums,
                                                                AT* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo, int act, float slope,
                                                                int training, int rpb) {
  const int G = C / CV, gq = threadIdx.x % G, rl = threadIdx.x / G, RL = 256 / G, c = gq * CV;
  if (rl >= RL) return;
  sums += (size_t)BN_BWD_SLOTS * 2 * C;     // the folded image (bn_bwd_fold_kernel)
  double k1[CV], k2[CV], k3[CV];
  float s_[CV], h_[CV];
  const double invM = 1.0 / ((double)N * H * W);
#pragma unroll
  for (int j = 0; j < CV; ++j) {
    const double gm = gamma[c + j], rs = rstd[c + j], mu = mean[c + j];
    k1[j] = gm * rs;
    if (training) { const double sa = sums[c + j] * invM, sb = sums[C + c + j] * invM; k3[j] = gm * rs * sb * rs; k2[j] = gm * rs * sa - k3[j] * mu; }
    else { k2[j] = 0.0; k3[j] = 0.0; }
    s_[j] = fsc[c + j]; h_[j] = fsh[c + j];
  }
  const float neg = act == SV_ACT_LRELU ? slope : 0.f;
  const int rp0 = blockIdx.x * rpb, rp1 = min(rp0 + rpb, N * Ho);
  for (int rp = rp0; rp < rp1; ++rp) {
    const int n = rp / Ho, a = rp - n * Ho;
    for (int b = rl; b < Wo; b += RL) {
      float d[4][CV];
      pool_block_grad<AT, CV>(dmp, idx, n, a, b, c, C, Ho, Wo, d);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int ih = 2 * a + (p >> 1), iw = 2 * b + (p & 1);
        if (ih < H && iw < W) {
          const size_t o = (((size_t)n * H + ih) * W + iw) * C + c;
          float xx[CV], out[CV];
          ldnf<CV>(x + o, xx);
#pragma unroll
          for (int j = 0; j < CV; ++j) {
            float dd = d[p][j];
            if (act != SV_ACT_NONE) dd *= __fmaf_rn(xx[j], s_[j], h_[j]) > 0.f ? 1.f : neg;
            out[j] = (float)(k1[j] * (double)dd - k2[j] - k3[j] * (double)xx[j]);
          }
          stnf<CV>(dx + o, out);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same fusion around MaxPool3d(2) (floor: 33 -> 16, 17 -> 8, 9 -> 4) for the three down-sampling layers of the Refiner (reference
// models/refiner.py:21-39: Conv3d -> BatchNorm3d -> LeakyReLU -> MaxPool3d): windows do not overlap, an input position lies in at most one
// window and the last plane of an odd grid in none.  tap = 4 dz + 2 dy + dx (sv_maxpool3d_fwd's numbering).  A thread owns CV channels of one
// window (forward) / of one 2 x 2 x 2 input block, partial blocks at the far faces included (backward: their gradient is -k2 - k3 x).
// ------------------------------------------------------------------------------------------------
template <typename AT, int CV>
__global__ __launch_bounds__(256) void bn_act_maxpool3d_fwd_kernel(const AT* __restrict__ x, const float* __restrict__ fsc, const float* __restrict__ fsh,
                                                                   AT* __restrict__ y, uint8_t* __restrict__ idx, int N, int D, int H, int W, int C,
                                                                   int act, float slope) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, cv = C / CV;
  const long long total = (long long)N * Do * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * CV; long long t = i / cv;
    const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); t /= Ho; const int od = (int)(t % Do); const int n = (int)(t / Do);
    float s_[CV], h_[CV], best[CV];
    int bi[CV];
#pragma unroll
    for (int j = 0; j < CV; ++j) { s_[j] = fsc[c + j]; h_[j] = fsh[c + j]; best[j] = -3.4e38f; bi[j] = 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v[CV];
      ldnf<CV>(x + ((((size_t)n * D + od * 2 + (k >> 2)) * H + oh * 2 + ((k >> 1) & 1)) * W + ow * 2 + (k & 1)) * C + c, v);
#pragma unroll
      for (int j = 0; j < CV; ++j) {
        const float z = (float)(AT)apply_act(__fmaf_rn(v[j], s_[j], h_[j]), act, slope);   // what scale_shift_act would have stored
        if (z > best[j]) { best[j] = z; bi[j] = k; }
      }
    }
    const size_t o = ((((size_t)n * Do + od) * Ho + oh) * Wo + ow) * C + c;
    stnf<CV>(y + o, best);
    typename TapWord<CV>::type w = 0;
#pragma unroll
    for (int j = 0; j < CV; ++j) w |= (typename TapWord<CV>::type)bi[j] << (8 * j);
    *reinterpret_cast<typename TapWord<CV>::type*>(idx + o) = w;
  }
}

// APPLY = false: pass 1 (sums of dz' and dz' * xhat into slot images); APPLY = true: pass 2 (dx = k1 dz' - k2 - k3 x)
template <typename AT, int CV, bool APPLY>
__global__ __launch_bounds__(256) void bn_pool3d_bwd_kernel(const AT* __restrict__ dmp, const uint8_t* __restrict__ idx, const AT* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ fsc, const float* __restrict__ fsh, double* __restrict__ sums,
                                                            AT* __restrict__ dx, int N, int D, int H, int W, int C, int act, float slope, int training,
                                                            long long bpw) {
  __shared__ double red[APPLY ? 1 : 256][2 * CV + 1];
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, BD = (D + 1) / 2, BH = (H + 1) / 2, BW = (W + 1) / 2;
  const int G = C / CV, gq = threadIdx.x % G, rl = threadIdx.x / G, RL = 256 / G, c = gq * CV;
  const long long nblk = (long long)N * BD * BH * BW;
  const long long q0 = (long long)blockIdx.x * bpw, q1 = q0 + bpw < nblk ? q0 + bpw : nblk;
  double s1[CV], s2[CV], k1[CV], k2[CV], k3[CV];
  float mm[CV], rr[CV], s_[CV], h_[CV];
#pragma unroll
  for (int j = 0; j < CV; ++j) { s1[j] = 0.0; s2[j] = 0.0; k1[j] = k2[j] = k3[j] = 0.0; mm[j] = rr[j] = s_[j] = h_[j] = 0.f; }
  if (rl < RL) {
#pragma unroll
    for (int j = 0; j < CV; ++j) { mm[j] = mean[c + j]; rr[j] = rstd[c + j]; s_[j] = fsc[c + j]; h_[j] = fsh[c + j]; }
    if constexpr (APPLY) {
      const double* fin = sums + (size_t)BN_BWD_SLOTS * 2 * C;     // the folded image (bn_bwd_fold_kernel)
      const double invM = 1.0 / ((double)N * D * H * W);
#pragma unroll
      for (int j = 0; j < CV; ++j) {
        const double gm = gamma[c + j], rs = rr[j], mu = mm[j];
        k1[j] = gm * rs;
        if (training) { const double sa = fin[c + j] * invM, sb = fin[C + c + j] * invM; k3[j] = gm * rs * sb * rs; k2[j] = gm * rs * sa - k3[j] * mu; }
      }
    }
    const float neg = act == SV_ACT_LRELU ? slope : 0.f;
    for (long long q = q0 + rl; q < q1; q += RL) {
      long long t = q;
      const int bw = (int)(t % BW); t /= BW; const int bh = (int)(t % BH); t /= BH; const int bd = (int)(t % BD); const int n = (int)(t / BD);
      const bool win = bd < Do && bh < Ho && bw < Wo;
      float g[CV];
      int tp[CV];
#pragma unroll
      for (int j = 0; j < CV; ++j) { g[j] = 0.f; tp[j] = 255; }
      if (win) {
        const size_t o = ((((size_t)n * Do + bd) * Ho + bh) * Wo + bw) * C + c;
        ldnf<CV>(dmp + o, g);
        ld_taps<CV>(idx + o, tp);
      }
      float t1[CV], t2[CV];
#pragma unroll
      for (int j = 0; j < CV; ++j) { t1[j] = 0.f; t2[j] = 0.f; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int id = 2 * bd + (k >> 2), ih = 2 * bh + ((k >> 1) & 1), iw = 2 * bw + (k & 1);
        if (id < D && ih < H && iw < W) {
          const size_t o = ((((size_t)n * D + id) * H + ih) * W + iw) * C + c;
          float xx[CV], out[CV];
          ldnf<CV>(x + o, xx);
#pragma unroll
          for (int j = 0; j < CV; ++j) {
            float dd = tp[j] == k ? g[j] : 0.f;
            if (act != SV_ACT_NONE) dd *= __fmaf_rn(xx[j], s_[j], h_[j]) > 0.f ? 1.f : neg;
            if constexpr (APPLY) out[j] = (float)(k1[j] * (double)dd - k2[j] - k3[j] * (double)xx[j]);
            else { t1[j] += dd; t2[j] += dd * (xx[j] - mm[j]) * rr[j]; }
          }
          if constexpr (APPLY) stnf<CV>(dx + o, out);
        }
      }
      if constexpr (!APPLY) {
#pragma unroll
        for (int j = 0; j < CV; ++j) { s1[j] += (double)t1[j]; s2[j] += (double)t2[j]; }
      }
    }
  }
  if constexpr (!APPLY) {
#pragma unroll
    for (int j = 0; j < CV; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][CV + j] = s2[j]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * CV * G; t += 256) {
      const int q = t / (2 * CV), k = t % (2 * CV);
      double acc = 0.0;
      for (int l = 0; l < RL; ++l) acc += red[l * G + q][k];
      atomicAdd(sums + (size_t)(blockIdx.x % BN_BWD_SLOTS) * 2 * C + (k < CV ? 0 : C) + q * CV + (k % CV), acc);
    }
  }
}

}  // namespace sv

using namespace sv;

template <bool MERGE, typename AT, int VEC>
static void ln_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long long rows, int C,
                          float eps, MergeMap mm, hipStream_t s) {
  int lpr, nv;
  ln_shape(C / VEC, lpr, nv);
  const int rpb = 4 * (64 / lpr);   // rows per workgroup
  SV_LN_DISPATCH(lpr, nv, hipLaunchKernelGGL((ln_fwd_kernel<MERGE, AT, VEC, LPR, NV>), dim3(cdiv(rows, rpb)), dim3(256), 0, s,
                                             static_cast<const AT*>(x), gamma, beta, static_cast<AT*>(y), mean, rstd, rows, C, eps, mm););
}
template <bool MERGE, typename AT, int VEC>
static void ln_bwd_launch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                          float* dbeta, long long rows, int C, MergeMap mm, int accumulate_dx, float* ws, hipStream_t s) {
  int lpr, nv;
  ln_shape(C / VEC, lpr, nv);
  const int rpi = 4 * (64 / lpr);                       // rows per workgroup iteration
  long long rpb = rows / 1024; if (rpb < rpi) rpb = rpi; if (rpb > 256) rpb = 256;   // >= ~1024 workgroups where the row count allows
  rpb = (rpb + rpi - 1) / rpi * rpi;
  const size_t lds = sizeof(float) * 2 * C;
  SV_LN_DISPATCH(lpr, nv, hipLaunchKernelGGL((ln_bwd_kernel<MERGE, AT, VEC, LPR, NV>), dim3(cdiv(rows, rpb)), dim3(256), lds, s,
                                             static_cast<const AT*>(dy), static_cast<const AT*>(x), gamma, mean, rstd, static_cast<AT*>(dx),
                                             dgamma, dbeta, rows, C, mm, accumulate_dx, (int)rpb, ws););
  hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, ws, C, dgamma, dbeta);
}
// VEC = 8 needs bf16 storage, 8-element channel groups (also of the un-merged map) and 16-byte aligned tensors
static inline bool ln_vec8(int act_dtype, int C, int merge_H, const void* a, const void* b, const void* c = nullptr) {
  return act_dtype == SV_BF16 && C % 8 == 0 && (merge_H == 0 || (C / 4) % 8 == 0) &&
         (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

extern "C" int sv_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                long long rows, int C, float eps, int merge_H, int merge_W, int act_dtype, void* stream) {
  SV_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0, "layernorm_fwd: null/empty argument");
  SV_REQUIRE(C % 4 == 0 && C <= 3072, "layernorm_fwd: C=%d must be a multiple of 4 and <= 3072", C);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "layernorm_fwd: gamma/beta must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  MergeMap mm{merge_H, merge_W, merge_H > 0 ? C / 4 : 0};
  if (merge_H > 0) SV_REQUIRE(merge_H % 2 == 0 && merge_W % 2 == 0 && C % 16 == 0, "layernorm_fwd(merge): H,W must be even and C a multiple of 16");
  const bool v8 = ln_vec8(act_dtype, C, merge_H, x, y);
  if (merge_H > 0) {
    if (v8) ln_fwd_launch<true, __bf16, 8>(x, gamma, beta, y, mean, rstd, rows, C, eps, mm, s);
    else if (act_dtype == SV_BF16) ln_fwd_launch<true, __bf16, 4>(x, gamma, beta, y, mean, rstd, rows, C, eps, mm, s);
    else ln_fwd_launch<true, float, 4>(x, gamma, beta, y, mean, rstd, rows, C, eps, mm, s);
  } else {
    if (v8) ln_fwd_launch<false, __bf16, 8>(x, gamma, beta, y, mean, rstd, rows, C, eps, mm, s);
    else if (act_dtype == SV_BF16) ln_fwd_launch<false, __bf16, 4>(x, gamma, beta, y, mean, rstd, rows, C, eps, mm, s);
    else ln_fwd_launch<false, float, 4>(x, gamma, beta, y, mean, rstd, rows, C, eps, mm, s);
  }
  return check_launch("sv_layernorm_fwd");
}

extern "C" size_t sv_layernorm_bwd_workspace_floats(int C) { return (size_t)LN_BWD_SLOTS * 2 * C + 2; }

extern "C" int sv_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                void* dx, float* dgamma, float* dbeta, float* workspace, long long rows, int C, int merge_H, int merge_W,
                                int accumulate_dx, int act_dtype, void* stream) {
  SV_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace && rows > 0, "layernorm_bwd: null/empty argument");
  SV_REQUIRE(C % 4 == 0 && C <= 3072, "layernorm_bwd: C=%d unsupported", C);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE(((uintptr_t)gamma & 15) == 0, "layernorm_bwd: gamma must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  MergeMap mm{merge_H, merge_W, merge_H > 0 ? C / 4 : 0};
  const bool v8 = ln_vec8(act_dtype, C, merge_H, dy, x, dx);
  if (merge_H > 0) {
    if (v8) ln_bwd_launch<true, __bf16, 8>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, mm, accumulate_dx, workspace, s);
    else if (act_dtype == SV_BF16) ln_bwd_launch<true, __bf16, 4>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, mm, accumulate_dx, workspace, s);
    else ln_bwd_launch<true, float, 4>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, mm, accumulate_dx, workspace, s);
  } else {
    if (v8) ln_bwd_launch<false, __bf16, 8>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, mm, accumulate_dx, workspace, s);
    else if (act_dtype == SV_BF16) ln_bwd_launch<false, __bf16, 4>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, mm, accumulate_dx, workspace, s);
    else ln_bwd_launch<false, float, 4>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, mm, accumulate_dx, workspace, s);
  }
  return check_launch("sv_layernorm_bwd");
}

// image slices of lnl_bwd_apply_kernel: enough workgroups for ~8 per CU, at least 8 images per slice
static inline int lnl_image_slices(int I, int L) {
  int s = 2048 / cdiv(L, 1024);
  if (s > I / 8) s = I / 8;
  return s < 1 ? 1 : (s > 32 ? 32 : s);
}

extern "C" size_t sv_ln_image_workspace_floats(int I, int L) { return (size_t)I * cdiv(L, LNL_CHUNK) * 2; }

extern "C" int sv_ln_image_fwd(const void* x, const float* w, const float* b, void* y, float* meanrstd, float* workspace,
                               int I, int L, float eps, float drop_p, uint32_t seed, const uint32_t* seed_epoch, int act_dtype, void* stream) {
  SV_REQUIRE(x && w && b && y && meanrstd && workspace && I > 0 && L > 0 && L % 4 == 0, "ln_image_fwd: bad arguments (I=%d L=%d)", I, L);
  SV_REQUIRE_ACT(act_dtype);
  hipStream_t s = (hipStream_t)stream;
  const int nch = cdiv(L, LNL_CHUNK);
  int gx = cdiv(L, 1024); if (gx > 64) gx = 64;
  SV_DISPATCH_ACT(act_dtype,
    hipLaunchKernelGGL(lnl_moments_kernel<AT>, dim3(nch, I), dim3(256), 0, s, static_cast<const AT*>(x), workspace, L, nch);
    hipLaunchKernelGGL(lnl_finalize_kernel, dim3(cdiv(I, 64)), dim3(64), 0, s, workspace, meanrstd, L, nch, eps, I);
    hipLaunchKernelGGL(lnl_apply_kernel<AT>, dim3(gx, I), dim3(256), 0, s, static_cast<const AT*>(x), w, b, meanrstd, static_cast<AT*>(y), L, drop_p, seed, seed_epoch););
  return check_launch("sv_ln_image_fwd");
}

extern "C" int sv_ln_image_bwd(const void* dy, const void* x, const float* w, const float* meanrstd, void* dx, float* dw,
                               float* db, double* sums_ws, int I, int L, float drop_p, uint32_t seed, const uint32_t* seed_epoch, int act_dtype, void* stream) {
  SV_REQUIRE(dy && x && w && meanrstd && dx && dw && db && sums_ws && I > 0 && L > 0 && L % 4 == 0, "ln_image_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(sums_ws, 0, sizeof(double) * 2 * I, s);
  int gx = cdiv(L, 1024); if (gx > 32) gx = 32;
  SV_DISPATCH_ACT(act_dtype,
    hipLaunchKernelGGL(lnl_bwd_reduce_kernel<AT>, dim3(gx, I), dim3(256), 0, s, static_cast<const AT*>(dy), static_cast<const AT*>(x), w, meanrstd, sums_ws, L, drop_p, seed, seed_epoch);
    hipLaunchKernelGGL(lnl_bwd_apply_kernel<AT>, dim3(cdiv(L, 1024), lnl_image_slices(I, L)), dim3(256), 0, s, static_cast<const AT*>(dy), static_cast<const AT*>(x), w, meanrstd, sums_ws,
                       static_cast<AT*>(dx), dw, db, L, I, drop_p, seed, seed_epoch););
  return check_launch("sv_ln_image_bwd");
}

extern "C" int sv_bn_stats(const void* x, long long M, int C, int ld, double* sums, int act_dtype, void* stream) {
  SV_REQUIRE(x && sums && M > 0 && C > 0 && ld >= C, "bn_stats: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  const int cg = cdiv(C, 64);
  long long splits = 2048 / cg; if (splits < 1) splits = 1;
  const long long maxs = (M + 63) / 64; if (splits > maxs) splits = maxs;
  const long long rpb = (M + splits - 1) / splits;
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(bn_stats_kernel<AT>, dim3(cg, cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, static_cast<const AT*>(x), M, C, ld, sums, rpb););
  return check_launch("sv_bn_stats");
}

extern "C" int sv_bn_finalize(const double* sums, long long count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                              float* save_mean, float* save_rstd, int C, void* stream) {
  SV_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && save_mean && save_rstd && C > 0, "bn_finalize: null argument");
  SV_REQUIRE(!training || (sums && count > 0), "bn_finalize: training needs sums and a positive count");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, training, scale, shift, save_mean, save_rstd, C);
  return check_launch("sv_bn_finalize");
}

// vector paths need every activation pointer aligned to 4 elements of the storage type
static inline bool aligned4(int act_dtype, const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr,
                            const void* e = nullptr) {
  const uintptr_t m = act_dtype == SV_BF16 ? 7 : 15;
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & m) == 0;
}

// sign words of sv_scale_shift_act_signs / sv_bn_bwd_signs need the wave-per-row mapping of the vector kernels: G == 64 column groups
static inline bool bn_signs_ok(int C) { return C % 256 == 0; }

static int scale_shift_act_impl(const void* x, int ldx, const float* scale, const float* shift, const void* residual, int ldr,
                                void* y, int ldy, long long M, int C, int act, float slope, unsigned long long* signs, int act_dtype, void* stream) {
  SV_REQUIRE(x && scale && shift && y && M > 0 && C > 0 && ldx >= C && ldy >= C, "scale_shift_act: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (!residual || ldr % 4 == 0) && aligned4(act_dtype, x, y, residual) &&
                   (((uintptr_t)scale | (uintptr_t)shift) & 15) == 0;
  const bool narrow = !vec && C <= BN_NARROW_MAXC && (ldx % 4 == 0) && (ldy % 4 == 0) && ldx >= ((C + 3) & ~3) && ldy >= ((C + 3) & ~3) &&
                      (!residual || (ldr % 4 == 0 && ldr >= ((C + 3) & ~3))) && aligned4(act_dtype, x, y, residual);
  SV_REQUIRE(!signs || (vec && bn_signs_ok(C) && ((uintptr_t)signs & 15) == 0), "scale_shift_act_signs: needs C %% 256 == 0 and 4-aligned rows (C=%d)", C);
  if (narrow) {
    long long blocks = (M + 63) / 64; if (blocks > 8192) blocks = 8192;
    SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(scale_shift_act_narrow_kernel<AT>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const AT*>(x), ldx, scale, shift,
                                                  static_cast<const AT*>(residual), ldr, static_cast<AT*>(y), ldy, M, C, act, slope););
  } else if (vec) {
    const int G = C / 4 < 64 ? C / 4 : 64, RL = 256 / G;
    const int cg = cdiv(C / 4, G);
    long long splits = 4096 / cg; if (splits < 1) splits = 1;
    const long long maxs = (M + 4 * RL - 1) / (4 * RL); if (splits > maxs) splits = maxs;
    const long long rpb = (M + splits - 1) / splits;
    SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(scale_shift_act_cg_kernel<AT>, dim3(cg, cdiv(M, rpb)), dim3(256), 0, s, static_cast<const AT*>(x), ldx, scale, shift,
                                                  static_cast<const AT*>(residual), ldr, static_cast<AT*>(y), ldy, M, C, act, slope, rpb, G, signs););
  } else {
    long long blocks = (M * C + 255) / 256; if (blocks > 8192) blocks = 8192;
    SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(scale_shift_act_scalar_kernel<AT>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const AT*>(x), ldx, scale, shift,
                                                  static_cast<const AT*>(residual), ldr, static_cast<AT*>(y), ldy, M, C, act, slope););
  }
  return check_launch("sv_scale_shift_act");
}

extern "C" int sv_scale_shift_act(const void* x, int ldx, const float* scale, const float* shift, const void* residual, int ldr,
                                  void* y, int ldy, long long M, int C, int act, float slope, int act_dtype, void* stream) {
  return scale_shift_act_impl(x, ldx, scale, shift, residual, ldr, y, ldy, M, C, act, slope, nullptr, act_dtype, stream);
}
extern "C" int sv_bn_signs_supported(int C) { return bn_signs_ok(C) ? 1 : 0; }
extern "C" int sv_scale_shift_act_signs(const void* x, int ldx, const float* scale, const float* shift, const void* residual, int ldr,
                                        void* y, int ldy, long long M, int C, int act, float slope, void* signs, int act_dtype, void* stream) {
  SV_REQUIRE(signs, "scale_shift_act_signs: null sign buffer");
  return scale_shift_act_impl(x, ldx, scale, shift, residual, ldr, y, ldy, M, C, act, slope, static_cast<unsigned long long*>(signs), act_dtype, stream);
}

// ---- ResNet stem, fused with its max-pool (kernels above).  x = the convolution's output [N, H, W, C] (rows of exactly C elements, C % 4 == 0,
// C <= 256 with 256 % (C / 4) == 0), pooled / idx = [N, Ho, Wo, C] with Ho = (H + 1) / 2, Wo = (W + 1) / 2 (kernel 3, stride 2, padding 1).
static bool bn_pool_shape_ok(int N, int H, int W, int C) {
  return N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && (long long)N * H * W * C < (1ll << 40);
}
// 8 channels per thread (16-byte accesses): bf16 storage, C % 8 == 0 with 256 % (C / 8) == 0, 16-byte aligned tensors, 8-byte aligned tap words
static bool bn_pool_cv8(int act_dtype, int C, const void* a, const void* b, const void* c) {
  return act_dtype == SV_BF16 && C % 8 == 0 && 256 % (C / 8) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}
extern "C" int sv_bn_act_maxpool_fwd(const void* x, const float* scale, const float* shift, void* pooled, void* idx, int N, int H, int W, int C,
                                     int act, float slope, int act_dtype, void* stream) {
  SV_REQUIRE(x && scale && shift && pooled && idx && bn_pool_shape_ok(N, H, W, C), "bn_act_maxpool_fwd: bad arguments (N=%d H=%d W=%d C=%d)", N, H, W, C);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE(aligned4(act_dtype, x, pooled, nullptr, nullptr, nullptr) && (((uintptr_t)scale | (uintptr_t)shift) & 15) == 0 && ((uintptr_t)idx & 3) == 0,
             "bn_act_maxpool_fwd: misaligned buffers");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  long long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384;
  if (bn_pool_cv8(act_dtype, C, x, pooled, idx)) {
    const long long t8 = total / 2;
    long long b8 = (t8 + 255) / 256; if (b8 > 16384) b8 = 16384;
    hipLaunchKernelGGL((bn_act_maxpool_fwd_kernel<__bf16, 8>), dim3((unsigned)b8), dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(x),
                       scale, shift, static_cast<__bf16*>(pooled), static_cast<uint8_t*>(idx), N, H, W, C, Ho, Wo, act, slope);
    return check_launch("sv_bn_act_maxpool_fwd");
  }
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL((bn_act_maxpool_fwd_kernel<AT, 4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const AT*>(x),
                                                scale, shift, static_cast<AT*>(pooled), static_cast<uint8_t*>(idx), N, H, W, C, Ho, Wo, act, slope););
  return check_launch("sv_bn_act_maxpool_fwd");
}
/* dpooled [N, Ho, Wo, C] + idx -> dx [N, H, W, C] (gradient w.r.t. the BatchNorm input x), dgamma / dbeta += ; sums_ws as sv_bn_bwd */
extern "C" int sv_bn_maxpool_bwd(const void* dpooled, const void* idx, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                 const float* fwd_scale, const float* fwd_shift, int N, int H, int W, int C, int act, float slope, int training,
                                 void* dx, float* dgamma, float* dbeta, double* sums_ws, int act_dtype, void* stream) {
  SV_REQUIRE(dpooled && idx && x && gamma && save_mean && save_rstd && fwd_scale && fwd_shift && dx && dgamma && dbeta && sums_ws && bn_pool_shape_ok(N, H, W, C),
             "bn_maxpool_bwd: bad arguments (N=%d H=%d W=%d C=%d)", N, H, W, C);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE(aligned4(act_dtype, dpooled, x, dx, nullptr, nullptr) && (((uintptr_t)save_mean | (uintptr_t)save_rstd | (uintptr_t)fwd_scale | (uintptr_t)fwd_shift) & 15) == 0 &&
             ((uintptr_t)idx & 3) == 0, "bn_maxpool_bwd: misaligned buffers");
  hipStream_t s = (hipStream_t)stream;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long long rows = (long long)N * Ho;     // image-row pairs
  SV_REQUIRE(rows < (1ll << 31), "bn_maxpool_bwd: more than 2^31 image rows");
  int rpb = (int)((rows + 2047) / 2048); if (rpb < 1) rpb = 1;
  const unsigned nb = (unsigned)((rows + rpb - 1) / rpb);
  int rpa = (int)((rows + 8191) / 8192); if (rpa < 1) rpa = 1;
  const unsigned na = (unsigned)((rows + rpa - 1) / rpa);
  if (bn_pool_cv8(act_dtype, C, dpooled, x, dx) && ((uintptr_t)idx & 7) == 0) {
    typedef __bf16 AT;
    hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<AT, 8>), dim3(nb), dim3(256), 0, s, static_cast<const AT*>(dpooled), static_cast<const uint8_t*>(idx), static_cast<const AT*>(x),
                       save_mean, save_rstd, fwd_scale, fwd_shift, sums_ws, N, H, W, C, Ho, Wo, act, slope, rpb);
    hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, sums_ws, C, dgamma, dbeta);
    hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<AT, 8>), dim3(na), dim3(256), 0, s, static_cast<const AT*>(dpooled), static_cast<const uint8_t*>(idx), static_cast<const AT*>(x),
                       gamma, save_mean, save_rstd, fwd_scale, fwd_shift, sums_ws, static_cast<AT*>(dx), N, H, W, C, Ho, Wo, act, slope, training, rpa);
    return check_launch("sv_bn_maxpool_bwd");
  }
  SV_DISPATCH_ACT(act_dtype,
    hipLaunchKernelGGL((bn_pool_bwd_reduce_kernel<AT, 4>), dim3(nb), dim3(256), 0, s, static_cast<const AT*>(dpooled), static_cast<const uint8_t*>(idx), static_cast<const AT*>(x),
                       save_mean, save_rstd, fwd_scale, fwd_shift, sums_ws, N, H, W, C, Ho, Wo, act, slope, rpb);
    hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, sums_ws, C, dgamma, dbeta);
    hipLaunchKernelGGL((bn_pool_bwd_apply_kernel<AT, 4>), dim3(na), dim3(256), 0, s, static_cast<const AT*>(dpooled), static_cast<const uint8_t*>(idx), static_cast<const AT*>(x),
                       gamma, save_mean, save_rstd, fwd_scale, fwd_shift, sums_ws, static_cast<AT*>(dx), N, H, W, C, Ho, Wo, act, slope, training, rpa););
  return check_launch("sv_bn_maxpool_bwd");
}

/* Refiner down-sampling layers: BatchNorm3d + activation + MaxPool3d(2) in one pass over the convolution's output x [N, D, H, W, C] (rows of exactly C
 * elements; C % 4 == 0 and 256 % (C / 4) == 0), pooled / idx [N, D/2, H/2, W/2, C] */
extern "C" int sv_bn_act_maxpool3d_fwd(const void* x, const float* scale, const float* shift, void* pooled, void* idx, int N, int D, int H, int W, int C,
                                       int act, float slope, int act_dtype, void* stream) {
  SV_REQUIRE(x && scale && shift && pooled && idx && D > 1 && bn_pool_shape_ok(N, H, W, C) && (long long)N * D * H * W * C < (1ll << 40),
             "bn_act_maxpool3d_fwd: bad arguments (N=%d D=%d H=%d W=%d C=%d)", N, D, H, W, C);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE(aligned4(act_dtype, x, pooled, nullptr, nullptr, nullptr) && ((uintptr_t)idx & 3) == 0, "bn_act_maxpool3d_fwd: misaligned buffers");
  const long long win = (long long)N * (D / 2) * (H / 2) * (W / 2);
  if (win == 0) return SV_OK;
  hipStream_t s = (hipStream_t)stream;
  if (bn_pool_cv8(act_dtype, C, x, pooled, idx)) {
    long long blocks = (win * (C / 8) + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((bn_act_maxpool3d_fwd_kernel<__bf16, 8>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const __bf16*>(x), scale, shift,
                       static_cast<__bf16*>(pooled), static_cast<uint8_t*>(idx), N, D, H, W, C, act, slope);
  } else {
    long long blocks = (win * (C / 4) + 255) / 256; if (blocks > 16384) blocks = 16384;
    SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL((bn_act_maxpool3d_fwd_kernel<AT, 4>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const AT*>(x), scale, shift,
                                                  static_cast<AT*>(pooled), static_cast<uint8_t*>(idx), N, D, H, W, C, act, slope););
  }
  return check_launch("sv_bn_act_maxpool3d_fwd");
}
extern "C" int sv_bn_maxpool3d_bwd(const void* dpooled, const void* idx, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                   const float* fwd_scale, const float* fwd_shift, int N, int D, int H, int W, int C, int act, float slope, int training,
                                   void* dx, float* dgamma, float* dbeta, double* sums_ws, int act_dtype, void* stream) {
  SV_REQUIRE(dpooled && idx && x && gamma && save_mean && save_rstd && fwd_scale && fwd_shift && dx && dgamma && dbeta && sums_ws && D > 1 &&
             bn_pool_shape_ok(N, H, W, C) && (long long)N * D * H * W * C < (1ll << 40), "bn_maxpool3d_bwd: bad arguments (N=%d D=%d H=%d W=%d C=%d)", N, D, H, W, C);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE(aligned4(act_dtype, dpooled, x, dx, nullptr, nullptr) && ((uintptr_t)idx & 3) == 0, "bn_maxpool3d_bwd: misaligned buffers");
  hipStream_t s = (hipStream_t)stream;
  const long long nblk = (long long)N * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
  long long bpr = (nblk + 2047) / 2048, bpa = (nblk + 8191) / 8192;
  const unsigned nr = (unsigned)((nblk + bpr - 1) / bpr), na = (unsigned)((nblk + bpa - 1) / bpa);
#define SV_POOL3D_BWD(AT_, CV_)                                                                                                                             \
  hipLaunchKernelGGL((bn_pool3d_bwd_kernel<AT_, CV_, false>), dim3(nr), dim3(256), 0, s, static_cast<const AT_*>(dpooled), static_cast<const uint8_t*>(idx),   \
                     static_cast<const AT_*>(x), gamma, save_mean, save_rstd, fwd_scale, fwd_shift, sums_ws, static_cast<AT_*>(dx), N, D, H, W, C, act, slope,  \
                     training, bpr);                                                                                                                        \
  hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, sums_ws, C, dgamma, dbeta);                                               \
  hipLaunchKernelGGL((bn_pool3d_bwd_kernel<AT_, CV_, true>), dim3(na), dim3(256), 0, s, static_cast<const AT_*>(dpooled), static_cast<const uint8_t*>(idx),    \
                     static_cast<const AT_*>(x), gamma, save_mean, save_rstd, fwd_scale, fwd_shift, sums_ws, static_cast<AT_*>(dx), N, D, H, W, C, act, slope,  \
                     training, bpa);
  if (bn_pool_cv8(act_dtype, C, dpooled, x, dx) && ((uintptr_t)idx & 7) == 0) { SV_POOL3D_BWD(__bf16, 8) }
  else { SV_DISPATCH_ACT(act_dtype, SV_POOL3D_BWD(AT, 4)); }
#undef SV_POOL3D_BWD
  return check_launch("sv_bn_maxpool3d_bwd");
}

extern "C" size_t sv_bn_bwd_workspace_doubles(int C) { return (size_t)(BN_BWD_SLOTS + 1) * 2 * C + 2; }

static int bn_bwd_impl(const void* dz, int lddz, const void* z, int ldz, const unsigned long long* signs, const void* x, int ldx, const float* gamma,
                       const float* save_mean, const float* save_rstd, long long M, int C, int act, float slope, int training,
                       void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta, double* sums_ws,
                       const float* fwd_scale, const float* fwd_shift, int act_dtype, void* stream) {
  SV_REQUIRE(dz && x && gamma && save_mean && save_rstd && dx && dgamma && dbeta && sums_ws && M > 0 && C > 0, "bn_bwd: null/empty argument");
  SV_REQUIRE(act == SV_ACT_NONE || z || signs || (fwd_scale && fwd_shift), "bn_bwd: the activation mask needs the forward output z, its sign words or the forward scale/shift");
  const float* fsc = fwd_scale; const float* fsh = fwd_shift;
  SV_REQUIRE_ACT(act_dtype);
  hipStream_t s = (hipStream_t)stream;   // sums_ws: sv_bn_bwd_workspace_doubles(C) doubles, ZERO on entry (e.g. a slice of one pre-zeroed arena)
  const bool vec = (C % 4 == 0) && (lddz % 4 == 0) && (ldx % 4 == 0) && (lddx % 4 == 0) && (!z || ldz % 4 == 0) && (!dres || lddres % 4 == 0) &&
                   aligned4(act_dtype, dz, z, x, dx, dres) && (((uintptr_t)save_mean | (uintptr_t)save_rstd) & 15) == 0;
  // sign words: the reduce pass takes the mask from them and hands the masked gradient on through dres (premask) - both vector-path features
  SV_REQUIRE(!signs || (vec && bn_signs_ok(C) && dres && act != SV_ACT_NONE && ((uintptr_t)signs & 15) == 0),
             "bn_bwd_signs: needs C %% 256 == 0, 4-aligned rows, an activation and the residual-branch output dres (C=%d)", C);
  SV_DISPATCH_ACT(act_dtype,
    const AT* dz_ = static_cast<const AT*>(dz); const AT* z_ = static_cast<const AT*>(z); const AT* x_ = static_cast<const AT*>(x);
    AT* dx_ = static_cast<AT*>(dx); AT* dres_ = static_cast<AT*>(dres);
    const int c4 = (C + 3) & ~3;
    const bool narrow = !vec && C <= BN_NARROW_MAXC && (lddz % 4 == 0) && (ldx % 4 == 0) && (lddx % 4 == 0) && lddz >= c4 && ldx >= c4 && lddx >= c4 &&
                        (!z || (ldz % 4 == 0 && ldz >= c4)) && (!dres || (lddres % 4 == 0 && lddres >= c4)) && aligned4(act_dtype, dz, z, x, dx, dres);
    if (narrow) {
      const int rpt = 16;
      const long long nb = (M + 64LL * rpt - 1) / (64LL * rpt);
      hipLaunchKernelGGL(bn_bwd_reduce_narrow_kernel<AT>, dim3((unsigned)nb), dim3(256), 0, s, dz_, lddz, z_, ldz, x_, ldx, save_mean, save_rstd, M, C, act, slope,
                         sums_ws, rpt, fsc, fsh);
      hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, sums_ws, C, dgamma, dbeta);
      long long blocks = (M + 63) / 64; if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(bn_bwd_apply_narrow_kernel<AT>, dim3((unsigned)blocks), dim3(256), 0, s, dz_, lddz, z_, ldz, x_, ldx, gamma, save_mean, save_rstd, sums_ws, M, C,
                         act, slope, training, dx_, lddx, dres_, lddres, fsc, fsh);
    } else if (vec) {
      const int G = C / 4 < 64 ? C / 4 : 64, RL = 256 / G;
      const int cg = cdiv(C / 4, G);
      long long splits = 2048 / cg; if (splits < 1) splits = 1;
      const long long maxs = (M + 2 * RL - 1) / (2 * RL); if (splits > maxs) splits = maxs;
      const long long rpb = (M + splits - 1) / splits;
      // with a residual-branch gradient to produce (dres) the reduce pass stores the masked gradient there and the apply pass reads IT,
      // mask-free, instead of (dz, z): 7 instead of 8 passes over the tensor.  The sums are of the fp32 values, the stored ones are
      // rounded to the storage type - as dres always was.
      static const int premask_on = [] { const char* v = getenv("SV_BN_PREMASK"); return v ? atoi(v) : 1; }();
      const bool premask = (premask_on || signs) && dres_ && act != SV_ACT_NONE;
      hipLaunchKernelGGL(bn_bwd_reduce_vec_kernel<AT>, dim3(cg, cdiv(M, rpb)), dim3(256), 0, s, dz_, lddz, z_, ldz, x_, ldx, save_mean, save_rstd, M, C, act, slope,
                         sums_ws, rpb, G, fsc, fsh, premask ? dres_ : (AT*)nullptr, lddres, signs);
      hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, sums_ws, C, dgamma, dbeta);
      long long asplits = 4096 / cg; if (asplits < 1) asplits = 1;
      const long long amax = (M + 4 * RL - 1) / (4 * RL); if (asplits > amax) asplits = amax;
      const long long arpb = (M + asplits - 1) / asplits;
      if (premask)
        hipLaunchKernelGGL(bn_bwd_apply_cg_kernel<AT>, dim3(cg, cdiv(M, arpb)), dim3(256), 0, s, (const AT*)dres_, lddres, (const AT*)nullptr, 0, x_, ldx, gamma, save_mean,
                           save_rstd, sums_ws, M, C, (int)SV_ACT_NONE, 0.f, training, dx_, lddx, (AT*)nullptr, 0, fsc, fsh, arpb, G);
      else
        hipLaunchKernelGGL(bn_bwd_apply_cg_kernel<AT>, dim3(cg, cdiv(M, arpb)), dim3(256), 0, s, dz_, lddz, z_, ldz, x_, ldx, gamma, save_mean, save_rstd, sums_ws, M, C,
                           act, slope, training, dx_, lddx, dres_, lddres, fsc, fsh, arpb, G);
    } else {
      int CW = 1; while (CW < C && CW < 64) CW <<= 1;         // channel lanes: next power of two of C, at most 64
      const int cg = cdiv(C, CW), RLg = 256 / CW;
      long long splits = 2048 / cg; if (splits < 1) splits = 1;
      const long long maxs = (M + 4 * RLg - 1) / (4 * RLg); if (splits > maxs) splits = maxs;
      const long long rpb = (M + splits - 1) / splits;
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<AT>, dim3(cg, cdiv(M, rpb)), dim3(256), 0, s, dz_, lddz, z_, ldz, x_, ldx, save_mean, save_rstd, M, C, act, slope, sums_ws, rpb, CW, fsc, fsh);
      hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, sums_ws, C, dgamma, dbeta);
      long long blocks = (M * C + 255) / 256; if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(bn_bwd_apply_kernel<AT>, dim3((unsigned)blocks), dim3(256), 0, s, dz_, lddz, z_, ldz, x_, ldx, gamma, save_mean, save_rstd, sums_ws, M, C,
                         act, slope, training, dx_, lddx, dres_, lddres, dgamma, dbeta, fsc, fsh);
    });
  return check_launch("sv_bn_bwd");
}

extern "C" int sv_bn_bwd(const void* dz, int lddz, const void* z, int ldz, const void* x, int ldx, const float* gamma,
                         const float* save_mean, const float* save_rstd, long long M, int C, int act, float slope, int training,
                         void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta, double* sums_ws,
                         const float* fwd_scale, const float* fwd_shift, int act_dtype, void* stream) {
  return bn_bwd_impl(dz, lddz, z, ldz, nullptr, x, ldx, gamma, save_mean, save_rstd, M, C, act, slope, training, dx, lddx, dres, lddres, dgamma, dbeta,
                     sums_ws, fwd_scale, fwd_shift, act_dtype, stream);
}
extern "C" int sv_bn_bwd_signs(const void* dz, int lddz, const void* signs, const void* x, int ldx, const float* gamma,
                               const float* save_mean, const float* save_rstd, long long M, int C, int act, float slope, int training,
                               void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta, double* sums_ws, int act_dtype, void* stream) {
  SV_REQUIRE(signs, "bn_bwd_signs: null sign buffer");
  return bn_bwd_impl(dz, lddz, nullptr, 0, static_cast<const unsigned long long*>(signs), x, ldx, gamma, save_mean, save_rstd, M, C, act, slope, training,
                     dx, lddx, dres, lddres, dgamma, dbeta, sums_ws, nullptr, nullptr, act_dtype, stream);
}
