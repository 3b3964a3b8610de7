// Implicit-GEMM contraction engine for gfx950 (MFMA 16x16x4 f32 / 16x16x32 bf16, fp32 accumulate).
//
// One kernel family serves every Linear / Conv2d / Conv3d / ConvTranspose3d of the SwinVox path:
//   igemm_kernel<BF16, TCONV>  out[row, co] = epilogue( sum_k A[row, k] * W[co, k] )
//       A is never materialised: k = (tap, ci) is decoded to a gathered input position on the fly
//       (channels-last activations, so the ci run of one tap is contiguous -> 16-byte loads).
//       TCONV = stride-s transposed gather, decomposed into s^3 output parity classes (blockIdx.z) so
//       that only taps that really contribute are visited.
//   wgrad_kernel<BF16>         dw[ca, cg, tap] += sum_r anchor[r, ca] * gathered[pos(r, tap), cg]
//       reduction over positions, split across blockIdx.z, fp32 atomics into the native weight layout.
// Tiles: 128x64 outputs per 256-thread workgroup (4 waves as 2x2, each 64x32 = 4x2 MFMA tiles), K-step 32,
// global->register prefetch of tile k+1 under the MFMAs of tile k (register-staged: the gather needs
// per-element predication, which LDS-DMA cannot express).
#include "common.h"

namespace sv {

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int PAD_F32 = 4;   // floats  -> row stride 36 floats (144 B, 16-B aligned)
constexpr int PAD_BF16 = 8;  // bf16    -> row stride 40 bf16 (80 B, 16-B aligned)

struct Geom {
  int N, Di, Hi, Wi, Do, Ho, Wo, Ci, Co, kd, kh, kw, sd, sh, sw, pd, ph, pw, ldi;
};
struct Epi {
  const float* bias; const float* residual; int ldr; const float* row_scale; int rows_per_scale;
  float* pre_act; float* stats; int act; float slope; const float* act_grad_src; int act_grad_kind; int ldc; int col_off;
};
struct ClassInfo {  // one output parity class of a transposed gather
  int o0[3];        // first output index of the class per axis
  int cnt[3];       // outputs of the class per axis
  int ib0[3];       // (o0 + p - r)/s : gathered index for j = 0, t = 0
  int r[3];         // residue = first kernel index of the class
  int T[3];         // taps of the class per axis
};
struct IGemmArgs {
  const float* x; const float* w; float* y;
  Geom g; Epi e;
  int Ktot;         // row length of packed weights = taps_total * Ci
  ClassInfo cls[8];
};

template <bool BF16> struct LdsT { typedef float T; static constexpr int PAD = PAD_F32; };
template <> struct LdsT<true> { typedef __bf16 T; static constexpr int PAD = PAD_BF16; };

__device__ __forceinline__ void store4(float* dst, float4 v) { *reinterpret_cast<float4*>(dst) = v; }
__device__ __forceinline__ void store4(__bf16* dst, float4 v) {
  bf16x4 b;
  b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4*>(dst) = b;
}
__device__ __forceinline__ void store1(float* dst, float v) { *dst = v; }
__device__ __forceinline__ void store1(__bf16* dst, float v) { *dst = (__bf16)v; }

// acc[4][2] += A(64 rows of this wave) x B(32 cols of this wave) over one BK=32 slab in LDS
template <bool BF16>
__device__ __forceinline__ void mma_slab(const typename LdsT<BF16>::T* As, const typename LdsT<BF16>::T* Bs,
                                         int wm, int wn, int lane, f32x4 (&acc)[4][2]) {
  constexpr int LD = BK + LdsT<BF16>::PAD;
  const int lr = lane & 15, lg = lane >> 4;
  if constexpr (BF16) {
    bf16x8 a[4], b[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + mt * 16 + lr) * LD + lg * 8);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 32 + nt * 16 + lr) * LD + lg * 8);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
  } else {
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float a[4], b[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = As[(wm * 64 + mt * 16 + lr) * LD + kk * 4 + lg];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b[nt] = Bs[(wn * 32 + nt * 16 + lr) * LD + kk * 4 + lg];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------------
template <bool BF16, bool TCONV>
__global__ __launch_bounds__(256) void igemm_kernel(const IGemmArgs p) {
  typedef typename LdsT<BF16>::T LT;
  constexpr int LD = BK + LdsT<BF16>::PAD;
  __shared__ __attribute__((aligned(16))) LT As[BM * LD];
  __shared__ __attribute__((aligned(16))) LT Bs[BN * LD];

  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // class setup (TCONV) or the single dense "class"
  int cnt0, cnt1, cnt2, T0, T1, T2;
  ClassInfo ci;
  if constexpr (TCONV) {
    ci = p.cls[blockIdx.z];
    cnt0 = ci.cnt[0]; cnt1 = ci.cnt[1]; cnt2 = ci.cnt[2];
    T0 = ci.T[0]; T1 = ci.T[1]; T2 = ci.T[2];
  } else {
    cnt0 = g.Do; cnt1 = g.Ho; cnt2 = g.Wo;
    T0 = g.kd; T1 = g.kh; T2 = g.kw;
  }
  const long long Mrows = (long long)g.N * cnt0 * cnt1 * cnt2;
  const int K = T0 * T1 * T2 * g.Ci;
  const long long row0 = (long long)blockIdx.x * BM;
  if (row0 >= Mrows) return;
  const int col0 = blockIdx.y * BN;

  // ---- per-thread loader state: 4 A rows + 2 B rows, one fixed 4-wide k chunk -----------------
  const int kq = (tid & 7) * 4;
  const int rb = tid >> 3;  // 0..31
  int a_n[4], a_d[4], a_h[4], a_w[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long m = row0 + rb + 32 * i;
    a_ok[i] = m < Mrows;
    if (!a_ok[i]) m = 0;
    int w_ = (int)(m % cnt2); long long t = m / cnt2;
    int h_ = (int)(t % cnt1); t /= cnt1;
    int d_ = (int)(t % cnt0); int n_ = (int)(t / cnt0);
    a_n[i] = n_;
    if constexpr (TCONV) {  // gathered index for tap t: ib0 + j - t
      a_d[i] = ci.ib0[0] + d_; a_h[i] = ci.ib0[1] + h_; a_w[i] = ci.ib0[2] + w_;
    } else {                // gathered index for tap t: o*s - p + t
      a_d[i] = d_ * g.sd - g.pd; a_h[i] = h_ * g.sh - g.ph; a_w[i] = w_ * g.sw - g.pw;
    }
  }
  const bool vec_ok = (g.Ci & 3) == 0 && (g.ldi & 3) == 0;

  float4 ra[4], rbv[2];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + kq;
    // ---- A (gathered activations)
    if (vec_ok) {
      int tap = k / g.Ci, c = k - tap * g.Ci;
      int tw = tap % T2; int t2 = tap / T2; int th = t2 % T1; int td = t2 / T1;
      const bool kok = k < K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int id, ih, iw;
        if constexpr (TCONV) { id = a_d[i] - td; ih = a_h[i] - th; iw = a_w[i] - tw; }
        else { id = a_d[i] + td; ih = a_h[i] + th; iw = a_w[i] + tw; }
        const bool ok = kok && a_ok[i] && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi;
        if (ok) {
          const size_t off = ((((size_t)a_n[i] * g.Di + id) * g.Hi + ih) * g.Wi + iw) * (size_t)g.ldi + c;
          ra[i] = *reinterpret_cast<const float4*>(p.x + off);
        } else {
          ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {  // scalar path: any Ci (stem Ci=3, refiner head Ci=1 ...)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = k + j;
          int tap = kk / g.Ci, c = kk - tap * g.Ci;
          int tw = tap % T2; int t2 = tap / T2; int th = t2 % T1; int td = t2 / T1;
          int id, ih, iw;
          if constexpr (TCONV) { id = a_d[i] - td; ih = a_h[i] - th; iw = a_w[i] - tw; }
          else { id = a_d[i] + td; ih = a_h[i] + th; iw = a_w[i] + tw; }
          const bool ok = kk < K && a_ok[i] && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi;
          v[j] = ok ? p.x[((((size_t)a_n[i] * g.Di + id) * g.Hi + ih) * g.Wi + iw) * (size_t)g.ldi + c] : 0.f;
        }
        ra[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    // ---- B (packed weights [Co][taps_total][Ci]); TCONV maps class-local taps to kernel indices
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = col0 + rb + 32 * i;
      float v[4];
      if (!TCONV && vec_ok) {
        if (n < g.Co && k < K) rbv[i] = *reinterpret_cast<const float4*>(p.w + (size_t)n * p.Ktot + k);
        else rbv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = k + j;
          float val = 0.f;
          if (n < g.Co && kk < K) {
            int kidx = kk;
            if constexpr (TCONV) {
              int tap = kk / g.Ci, c = kk - tap * g.Ci;
              int tw = tap % T2; int t2 = tap / T2; int th = t2 % T1; int td = t2 / T1;
              const int kd_ = ci.r[0] + g.sd * td, kh_ = ci.r[1] + g.sh * th, kw_ = ci.r[2] + g.sw * tw;
              kidx = ((kd_ * g.kh + kh_) * g.kw + kw_) * g.Ci + c;
            }
            val = p.w[(size_t)n * p.Ktot + kidx];
          }
          v[j] = val;
        }
        rbv[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) store4(As + (rb + 32 * i) * LD + kq, ra[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) store4(Bs + (rb + 32 * i) * LD + kq, rbv[i]);
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  if (nk > 0) {
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load_tile(kt + 1);
      mma_slab<BF16>(As, Bs, wm, wn, lane, acc);
      __syncthreads();
      if (kt + 1 < nk) {
        store_tile();
        __syncthreads();
      }
    }
  }

  // ---- epilogue: C tile element (row = (lane>>4)*4 + j, col = lane&15) ---------------------------
  const Epi& e = p.e;
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = col0 + wn * 32 + nt * 16 + lr;
    const bool nok = n < g.Co;
    const float bias = (nok && e.bias) ? e.bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long m = row0 + wm * 64 + mt * 16 + lg * 4 + j;
        if (m < Mrows && nok) {
          size_t pos;  // output position index (pixel) in the produced tensor
          if constexpr (TCONV) {
            int w_ = (int)(m % cnt2); long long t = m / cnt2;
            int h_ = (int)(t % cnt1); t /= cnt1;
            int d_ = (int)(t % cnt0); int n_ = (int)(t / cnt0);
            const int od = ci.o0[0] + g.sd * d_, oh = ci.o0[1] + g.sh * h_, ow = ci.o0[2] + g.sw * w_;
            pos = (((size_t)n_ * g.Do + od) * g.Ho + oh) * g.Wo + ow;
          } else {
            pos = (size_t)m;
          }
          float v = acc[mt][nt][j] + bias;
          if (e.act_grad_src) v *= act_grad(e.act_grad_src[pos * e.ldc + e.col_off + n], e.act_grad_kind, e.slope);
          if (e.pre_act) e.pre_act[pos * e.ldc + e.col_off + n] = v;
          v = apply_act(v, e.act, e.slope);
          if (e.residual) {
            const float sc = e.row_scale ? e.row_scale[pos / e.rows_per_scale] : 1.f;
            v = e.residual[pos * e.ldr + n] + sc * v;
          }
          p.y[pos * e.ldc + e.col_off + n] = v;
          s1 += v; s2 += v * v;
        }
      }
    }
    if (e.stats) {  // per-channel sum / sumsq of what was stored: reduce the 4 row groups, one atomic per column
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (lg == 0 && nok) {
        atomicAdd(e.stats + n, s1);
        atomicAdd(e.stats + g.Co + n, s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel
// ------------------------------------------------------------------------------------------------
struct WGradArgs {
  const float* anchor; int lda; const float* gathered; float* dw;
  Geom g; int cg_valid; long long rows_per_split; long long Mrows;
};

template <bool BF16>
__global__ __launch_bounds__(256) void wgrad_kernel(const WGradArgs p) {
  typedef typename LdsT<BF16>::T LT;
  constexpr int LD = BK + LdsT<BF16>::PAD;
  __shared__ __attribute__((aligned(16))) LT As[BM * LD];  // [ca][r]
  __shared__ __attribute__((aligned(16))) LT Bs[BN * LD];  // [k_out][r]

  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int taps = g.kd * g.kh * g.kw;
  const int Kout = taps * g.Ci;
  const int ca0 = blockIdx.x * BM, ko0 = blockIdx.y * BN;
  const long long r_begin = (long long)blockIdx.z * p.rows_per_split;
  long long r_end = r_begin + p.rows_per_split;
  if (r_end > p.Mrows) r_end = p.Mrows;
  if (r_begin >= r_end) return;

  // A loader: thread -> 4 consecutive ca, rows rr + 8*i (i<4)
  const int a_c = (tid & 31) * 4, a_r = tid >> 5;
  // B loader: thread -> 4 consecutive k_out (one tap, 4 cg), rows rr + 16*i (i<2)
  const int b_k = ko0 + (tid & 15) * 4, b_r = tid >> 4;
  const bool avec = (p.lda & 3) == 0, bvec = (g.Ci & 3) == 0 && (g.ldi & 3) == 0;
  int b_tap = 0, b_c = 0, b_td = 0, b_th = 0, b_tw = 0;
  if (bvec) {
    b_tap = b_k / g.Ci; b_c = b_k - b_tap * g.Ci;
    b_tw = b_tap % g.kw; int t2 = b_tap / g.kw; b_th = t2 % g.kh; b_td = t2 / g.kh;
  }

  float4 ra[4], rbv[2];
  auto load_tile = [&](long long r0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long r = r0 + a_r + 8 * i;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (r < r_end) {
        const float* src = p.anchor + (size_t)r * p.lda + ca0 + a_c;
        if (avec && ca0 + a_c + 3 < g.Co) {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (ca0 + a_c + j < g.Co) v[j] = src[j];
        }
      }
      ra[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long r = r0 + b_r + 16 * i;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (r < r_end) {
        int w_ = (int)(r % g.Wo); long long t = r / g.Wo;
        int h_ = (int)(t % g.Ho); t /= g.Ho;
        int d_ = (int)(t % g.Do); int n_ = (int)(t / g.Do);
        const int bd = d_ * g.sd - g.pd, bh = h_ * g.sh - g.ph, bw = w_ * g.sw - g.pw;
        if (bvec) {
          const int id = bd + b_td, ih = bh + b_th, iw = bw + b_tw;
          if (b_k < Kout && (unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi) {
            const float4 q = *reinterpret_cast<const float4*>(
                p.gathered + ((((size_t)n_ * g.Di + id) * g.Hi + ih) * g.Wi + iw) * (size_t)g.ldi + b_c);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int kk = b_k + j;
            if (kk < Kout) {
              int tap = kk / g.Ci, c = kk - tap * g.Ci;
              int tw = tap % g.kw; int t2 = tap / g.kw; int th = t2 % g.kh; int td = t2 / g.kh;
              const int id = bd + td, ih = bh + th, iw = bw + tw;
              if ((unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi)
                v[j] = p.gathered[((((size_t)n_ * g.Di + id) * g.Hi + ih) * g.Wi + iw) * (size_t)g.ldi + c];
            }
          }
        }
      }
      rbv[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_tile = [&]() {  // transposing stores: reduction index r becomes the contiguous LDS dimension
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = a_r + 8 * i;
      store1(As + (a_c + 0) * LD + r, ra[i].x); store1(As + (a_c + 1) * LD + r, ra[i].y);
      store1(As + (a_c + 2) * LD + r, ra[i].z); store1(As + (a_c + 3) * LD + r, ra[i].w);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = b_r + 16 * i;
      const int kl = (tid & 15) * 4;
      store1(Bs + (kl + 0) * LD + r, rbv[i].x); store1(Bs + (kl + 1) * LD + r, rbv[i].y);
      store1(Bs + (kl + 2) * LD + r, rbv[i].z); store1(Bs + (kl + 3) * LD + r, rbv[i].w);
    }
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tile(r_begin);
  store_tile();
  __syncthreads();
  for (long long r0 = r_begin; r0 < r_end; r0 += BK) {
    const bool more = r0 + BK < r_end;
    if (more) load_tile(r0 + BK);
    mma_slab<BF16>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }

  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ko = ko0 + wn * 32 + nt * 16 + lr;
    if (ko >= Kout) continue;
    const int tap = ko / g.Ci, cg = ko - tap * g.Ci;
    if (cg >= p.cg_valid) continue;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ca = ca0 + wm * 64 + mt * 16 + lg * 4 + j;
        if (ca < g.Co) atomicAdd(p.dw + ((size_t)ca * p.cg_valid + cg) * taps + tap, acc[mt][nt][j]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// small helpers: weight repack, column sums
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int T, int swap,
                                   int rows_out, int inner_out) {
  // dst[rows_out][T][inner_out]; swap=0: rows=A inner=B ; swap=1: rows=B inner=A ; zero beyond the valid range
  const long long total = (long long)rows_out * T * inner_out;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int in_ = (int)(i % inner_out); long long t2 = i / inner_out;
    const int t = (int)(t2 % T); const int ro = (int)(t2 / T);
    const int a = swap ? in_ : ro, b = swap ? ro : in_;
    dst[i] = (a < A && b < B) ? src[((size_t)a * B + b) * T + t] : 0.f;
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long rows, int cols, int ld,
                                                     float* __restrict__ out, long long rows_per_block) {
  // block = 64 columns x 4 row-lanes; grid.x = column groups, grid.y = row splits
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
  float s = 0.f;
  if (c < cols) for (long long r = r0 + rl; r < r1; r += 4) s += x[(size_t)r * ld + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < cols) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static Geom to_geom(const sv_geom* g) {
  Geom q{g->N, g->Di, g->Hi, g->Wi, g->Do, g->Ho, g->Wo, g->Ci, g->Co, g->kd, g->kh, g->kw,
         g->sd, g->sh, g->sw, g->pd, g->ph, g->pw, g->ldi};
  return q;
}
static Epi to_epi(const sv_epilogue* e) {
  Epi q{e->bias, e->residual, e->ldr, e->row_scale, e->rows_per_scale > 0 ? e->rows_per_scale : 1, e->pre_act, e->stats,
        e->act, e->slope, e->act_grad_src, e->act_grad_kind, e->ldc, e->col_off};
  return q;
}
static int check_common(const sv_geom* g, const sv_epilogue* e, const void* in, const void* w, const void* out) {
  SV_REQUIRE(g && e && in && w && out, "igemm: null argument");
  SV_REQUIRE(g->N > 0 && g->Ci > 0 && g->Co > 0, "igemm: N/Ci/Co must be positive (N=%d Ci=%d Co=%d)", g->N, g->Ci, g->Co);
  SV_REQUIRE(g->Di > 0 && g->Hi > 0 && g->Wi > 0 && g->Do > 0 && g->Ho > 0 && g->Wo > 0, "igemm: empty grid");
  SV_REQUIRE(g->kd > 0 && g->kh > 0 && g->kw > 0 && g->sd > 0 && g->sh > 0 && g->sw > 0, "igemm: kernel/stride must be positive");
  SV_REQUIRE(g->ldi >= g->Ci, "igemm: ldi (%d) < Ci (%d)", g->ldi, g->Ci);
  SV_REQUIRE(e->ldc >= e->col_off + g->Co, "igemm: ldc (%d) < col_off+Co (%d)", e->ldc, e->col_off + g->Co);
  SV_REQUIRE(!e->residual || e->ldr >= g->Co, "igemm: ldr (%d) < Co (%d)", e->ldr, g->Co);
  SV_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)w & 15) == 0, "igemm: in/w must be 16-byte aligned");
  return SV_OK;
}

}  // namespace sv

using namespace sv;

extern "C" int sv_conv_gather(const float* in, const float* w, float* out, const sv_geom* g, const sv_epilogue* e,
                              int math, void* stream) {
  if (int rc = check_common(g, e, in, w, out)) return rc;
  IGemmArgs a{};
  a.x = in; a.w = w; a.y = out; a.g = to_geom(g); a.e = to_epi(e);
  a.Ktot = g->kd * g->kh * g->kw * g->Ci;
  const long long M = (long long)g->N * g->Do * g->Ho * g->Wo;
  dim3 grid(cdiv(M, BM), cdiv(g->Co, BN), 1);
  hipStream_t s = (hipStream_t)stream;
  if (math == SV_MATH_BF16) hipLaunchKernelGGL((igemm_kernel<true, false>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((igemm_kernel<false, false>), grid, dim3(256), 0, s, a);
  return check_launch("sv_conv_gather");
}

extern "C" int sv_tconv_gather(const float* in, const float* w, float* out, const sv_geom* g, const sv_epilogue* e,
                               int math, void* stream) {
  if (int rc = check_common(g, e, in, w, out)) return rc;
  SV_REQUIRE(g->sd <= 2 && g->sh <= 2 && g->sw <= 2, "tconv_gather: stride > 2 unsupported (%d,%d,%d)", g->sd, g->sh, g->sw);
  IGemmArgs a{};
  a.x = in; a.w = w; a.y = out; a.g = to_geom(g); a.e = to_epi(e);
  a.Ktot = g->kd * g->kh * g->kw * g->Ci;
  const int O[3] = {g->Do, g->Ho, g->Wo}, S[3] = {g->sd, g->sh, g->sw}, P[3] = {g->pd, g->ph, g->pw}, Kx[3] = {g->kd, g->kh, g->kw};
  int ncls = 0;
  long long maxM = 0;
  for (int rd = 0; rd < S[0]; ++rd)
    for (int rh = 0; rh < S[1]; ++rh)
      for (int rw = 0; rw < S[2]; ++rw) {
        const int r[3] = {rd, rh, rw};
        ClassInfo c{};
        long long m = g->N;
        for (int ax = 0; ax < 3; ++ax) {
          c.r[ax] = r[ax];
          c.T[ax] = r[ax] < Kx[ax] ? (Kx[ax] - r[ax] + S[ax] - 1) / S[ax] : 0;
          int o0 = ((r[ax] - P[ax]) % S[ax] + S[ax]) % S[ax];
          c.o0[ax] = o0;
          c.cnt[ax] = o0 < O[ax] ? (O[ax] - o0 + S[ax] - 1) / S[ax] : 0;
          c.ib0[ax] = (o0 + P[ax] - r[ax]) / S[ax];
          m *= c.cnt[ax];
        }
        if (m > maxM) maxM = m;
        a.cls[ncls++] = c;
      }
  if (maxM == 0) return SV_OK;
  dim3 grid(cdiv(maxM, BM), cdiv(g->Co, BN), ncls);
  hipStream_t s = (hipStream_t)stream;
  if (math == SV_MATH_BF16) hipLaunchKernelGGL((igemm_kernel<true, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((igemm_kernel<false, true>), grid, dim3(256), 0, s, a);
  return check_launch("sv_tconv_gather");
}

extern "C" int sv_conv_wgrad(const float* anchor, int lda, const float* gathered, float* dw, const sv_geom* g, int cg_valid,
                             int math, void* stream) {
  SV_REQUIRE(anchor && gathered && dw && g, "wgrad: null argument");
  SV_REQUIRE(lda >= g->Co && g->ldi >= g->Ci && cg_valid > 0 && cg_valid <= g->Ci, "wgrad: bad strides (lda=%d Co=%d ldi=%d Ci=%d cg_valid=%d)",
             lda, g->Co, g->ldi, g->Ci, cg_valid);
  SV_REQUIRE(((uintptr_t)anchor & 15) == 0 && ((uintptr_t)gathered & 15) == 0, "wgrad: operands must be 16-byte aligned");
  WGradArgs a{};
  a.anchor = anchor; a.lda = lda; a.gathered = gathered; a.dw = dw; a.g = to_geom(g); a.cg_valid = cg_valid;
  a.Mrows = (long long)g->N * g->Do * g->Ho * g->Wo;
  const int Kout = g->kd * g->kh * g->kw * g->Ci;
  const int tiles = cdiv(g->Co, BM) * cdiv(Kout, BN);
  // enough splits to fill ~4 waves of workgroups per CU, but at least 4 K-steps of work per split
  long long splits = (1024 + tiles - 1) / tiles;
  const long long max_splits = (a.Mrows + 4 * BK - 1) / (4 * BK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  long long rps = (a.Mrows + splits - 1) / splits;
  rps = (rps + BK - 1) / BK * BK;
  splits = (a.Mrows + rps - 1) / rps;
  a.rows_per_split = rps;
  dim3 grid(cdiv(g->Co, BM), cdiv(Kout, BN), (unsigned)splits);
  hipStream_t s = (hipStream_t)stream;
  if (math == SV_MATH_BF16) hipLaunchKernelGGL((wgrad_kernel<true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((wgrad_kernel<false>), grid, dim3(256), 0, s, a);
  return check_launch("sv_conv_wgrad");
}

extern "C" int sv_pack_weight(const float* src, float* dst, int A, int B, int T, int swap, int pad_to, void* stream) {
  SV_REQUIRE(src && dst && A > 0 && B > 0 && T > 0, "pack_weight: bad arguments");
  const int rows_out = swap ? B : A;
  int inner = swap ? A : B;
  if (pad_to > inner) inner = pad_to;
  const long long total = (long long)rows_out * T * inner;
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, A, B, T, swap, rows_out, inner);
  return check_launch("sv_pack_weight");
}

extern "C" int sv_colsum(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* stream) {
  SV_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) hipMemsetAsync(out, 0, sizeof(float) * cols, s);
  const int cg = cdiv(cols, 64);
  int splits = 2048 / cg; if (splits < 1) splits = 1;
  const int maxs = cdiv(rows, 64); if (splits > maxs) splits = maxs;
  const long long rpb = (rows + splits - 1) / splits;
  hipLaunchKernelGGL(colsum_kernel, dim3(cg, cdiv(rows, rpb)), dim3(256), 0, s, x, (long long)rows, cols, ld, out, rpb);
  return check_launch("sv_colsum");
}
