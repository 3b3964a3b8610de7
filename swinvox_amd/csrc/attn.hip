// Attention kernels.
//  * Swin window / shifted-window attention core (timm WindowAttention behind models/swin_transformer.py:78):
//    one wave per (image, window, head); the cyclic shift, the 7x7 window partition/reverse and the 9-region
//    shift mask are folded into the token index math (no materialised roll); q/k/v tiles of the window are
//    staged in LDS, QK^T and PV run on MFMA, the softmax runs on the accumulator layout with 16-lane shuffles.
//  * Cross-view attention core (models/cross_view_attention.py:81-105): one workgroup per (sample, head),
//    V x V scores over 288-long features -- tiny, plain FMA.
#include "common.h"

namespace sv {

constexpr int WT = 49;       // tokens per window
constexpr int HD = 32;       // head dim (Swin-T/B: always 32)
constexpr int LDQ_F = 36;    // fp32 row stride of q/k/v tiles
constexpr int LDP_F = 68;    // fp32 row stride of the P tile
constexpr int LDQ_H = 40;    // bf16 row stride of q/k tiles
constexpr int LDP_H = 72;    // bf16 row stride of P and V^T tiles

template <typename AT>
struct WinArgsT {   // AT = storage element of the activations (qkv, out and their gradients)
  const AT* qkv; const float* table; AT* out;   // fwd
  const AT* dout; AT* dqkv; float* dtable;      // bwd
  int I, H, W, C, heads, shift; float scale; int ntasks; int tasks_per_wave;
};
typedef WinArgsT<float> WinArgs;

struct TokMap {  // window -> token rows of the un-shifted [I,H,W] map
  int img, wy, wx, H, W, shift;
  __device__ __forceinline__ long long row(int t) const {
    const int ty = t / 7, tx = t - ty * 7;
    int ys = wy * 7 + ty + shift; if (ys >= H) ys -= H;
    int xs = wx * 7 + tx + shift; if (xs >= W) xs -= W;
    return ((long long)img * H + ys) * W + xs;
  }
  __device__ __forceinline__ int region(int t) const {  // 9-region id in the rolled frame
    const int ty = t / 7, tx = t - ty * 7;
    const int ry = wy * 7 + ty, rx = wx * 7 + tx;
    const int a = ry < H - 7 ? 0 : (ry < H - shift ? 1 : 2);
    const int b = rx < W - 7 ? 0 : (rx < W - shift ? 1 : 2);
    return a * 3 + b;
  }
};

// all-reduce over the 16 lanes of a DPP row, every lane gets the result: quad_perm xor 1, xor 2, row_half_mirror (i <-> 7 - i: the two quads
// of a half), row_mirror (i <-> 15 - i: the two halves).  Four VALU DPP moves instead of four ds_bpermute round trips through the LDS crossbar.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));     // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov<0x4E>(v));     // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov<0x141>(v));    // row_half_mirror
  v = fmaxf(v, dpp_mov<0x140>(v));    // row_mirror
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
  return v;
}

// all-reduce over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48) on the VALU: the v_permlane16_swap / v_permlane32_swap
// exchanges of common.h (no LDS crossbar round trip as ds_bpermute / __shfl_xor would take)
template <typename OP> __device__ __forceinline__ float lanegroup_allreduce(float v, OP op) {
  float w = v;
  permlane16_pair(v, w);
  v = op(v, w); w = v;
  permlane32_pair(v, w);
  return op(v, w);
}

// scores (+bias, +mask, key padding) -> softmax, all on the 4x4 grid of 16x16 accumulator tiles of one wave.
// element (q = mt*16 + lg*4 + j, key = nt*16 + lr)
__device__ __forceinline__ void bias_mask_softmax(f32x4 (&s)[4][4], const float* bt, const TokMap& tm, int lane, bool shifted) {
  const int lr = lane & 15, lg = lane >> 4;
  int ky[4], kx[4], kreg[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int key = nt * 16 + lr;
    ky[nt] = key / 7; kx[nt] = key - ky[nt] * 7;
    kreg[nt] = (shifted && key < WT) ? tm.region(key) : 0;
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = mt * 16 + lg * 4 + j;
      const int qy = q / 7, qx = q - qy * 7;
      const int qreg = (shifted && q < WT) ? tm.region(q) : 0;
      float mx = -3.0e38f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int key = nt * 16 + lr;
        float v = s[mt][nt][j];
        if (key >= WT) v = -1.0e30f;
        else if (q < WT) {
          v += bt[(qy - ky[nt] + 6) * 13 + (qx - kx[nt] + 6)];
          if (shifted && qreg != kreg[nt]) v += -100.0f;
        }
        s[mt][nt][j] = v;
        mx = fmaxf(mx, v);
      }
      mx = group16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { const float e = expf(s[mt][nt][j] - mx); s[mt][nt][j] = e; sum += e; }
      const float inv = 1.f / group16_sum(sum);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) s[mt][nt][j] *= inv;
    }
}

// load one [49(64) x 32] head slice of q, k or v (or dO) into an fp32 LDS tile, rows >= 49 zeroed
template <typename AT>
__device__ __forceinline__ void load_tile_f32(float* dst, const AT* src, int ld, int col, const TokMap& tm, int lane, float mul) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = (lane >> 3) + 8 * it, ch = (lane & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < WT) {
      v = ld4f(src + (size_t)tm.row(r) * ld + col + ch);
      v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    }
    *reinterpret_cast<float4*>(dst + r * LDQ_F + ch) = v;
  }
}

template <bool BF16, typename AT>
__global__ __launch_bounds__(256) void win_attn_fwd_kernel(const WinArgsT<AT> p) {
  constexpr int WAVE_BYTES = BF16 ? (2 * 64 * LDQ_H + HD * LDP_H) * 2 + 176 * 4 : 3 * 64 * LDQ_F * 4 + 176 * 4;
  __shared__ __attribute__((aligned(16))) char smem[4 * WAVE_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int head = blockIdx.y;
  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  long long task = (long long)blockIdx.x * 4 + wave;
  const bool active = task < p.ntasks;
  if (!active) task = p.ntasks - 1;
  TokMap tm;
  tm.img = (int)(task / nW); const int win = (int)(task - (long long)tm.img * nW);
  tm.wy = win / nWx; tm.wx = win - tm.wy * nWx; tm.H = p.H; tm.W = p.W; tm.shift = p.shift;
  const int ld = 3 * p.C, colq = head * HD;
  char* base = smem + wave * WAVE_BYTES;
  float* bt = reinterpret_cast<float*>(base + WAVE_BYTES - 176 * 4);
  for (int i = lane; i < 169; i += 64) bt[i] = p.table[i * p.heads + head];

  f32x4 s[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 o[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) { o[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  if constexpr (!BF16) {
    float* Qs = reinterpret_cast<float*>(base);
    float* Ks = Qs + 64 * LDQ_F;
    float* Vs = Ks + 64 * LDQ_F;
    float* Ps = Qs;  // aliases Q,K once the scores are in registers
    load_tile_f32(Qs, p.qkv, ld, colq, tm, lane, p.scale);
    load_tile_f32(Ks, p.qkv, ld, p.C + colq, tm, lane, 1.f);
    load_tile_f32(Vs, p.qkv, ld, 2 * p.C + colq, tm, lane, 1.f);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < HD / 4; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = Qs[(t * 16 + lr) * LDQ_F + kk * 4 + lg]; b[t] = Ks[(t * 16 + lr) * LDQ_F + kk * 4 + lg]; }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], s[mt][nt], 0, 0, 0);
    }
    bias_mask_softmax(s, bt, tm, lane, p.shift > 0);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) Ps[(mt * 16 + lg * 4 + j) * LDP_F + nt * 16 + lr] = s[mt][nt][j];
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = Ps[(mt * 16 + lr) * LDP_F + kk * 4 + lg];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b[nt] = Vs[(kk * 4 + lg) * LDQ_F + nt * 16 + lr];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) o[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], o[mt][nt], 0, 0, 0);
    }
  } else {
    __bf16* Qs = reinterpret_cast<__bf16*>(base);
    __bf16* Ks = Qs + 64 * LDQ_H;
    __bf16* Vt = Ks + 64 * LDQ_H;   // [d][key]
    __bf16* Ps = Qs;                // [q][key], aliases Q,K
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = (lane >> 3) + 8 * it, ch = (lane & 7) * 4;
      float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4;
      if (r < WT) {
        const AT* src = p.qkv + (size_t)tm.row(r) * ld + colq + ch;
        q4 = ld4f(src);
        k4 = ld4f(src + p.C);
        v4 = ld4f(src + 2 * p.C);
      }
      bf16x4 qb, kb;
      qb[0] = (__bf16)(q4.x * p.scale); qb[1] = (__bf16)(q4.y * p.scale); qb[2] = (__bf16)(q4.z * p.scale); qb[3] = (__bf16)(q4.w * p.scale);
      kb[0] = (__bf16)k4.x; kb[1] = (__bf16)k4.y; kb[2] = (__bf16)k4.z; kb[3] = (__bf16)k4.w;
      *reinterpret_cast<bf16x4*>(Qs + r * LDQ_H + ch) = qb;
      *reinterpret_cast<bf16x4*>(Ks + r * LDQ_H + ch) = kb;
      Vt[(ch + 0) * LDP_H + r] = (__bf16)v4.x; Vt[(ch + 1) * LDP_H + r] = (__bf16)v4.y;
      Vt[(ch + 2) * LDP_H + r] = (__bf16)v4.z; Vt[(ch + 3) * LDP_H + r] = (__bf16)v4.w;
    }
    __syncthreads();
    {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const bf16x8*>(Qs + (t * 16 + lr) * LDQ_H + lg * 8);
        b[t] = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + lr) * LDQ_H + lg * 8);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], s[mt][nt], 0, 0, 0);
    }
    bias_mask_softmax(s, bt, tm, lane, p.shift > 0);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) Ps[(mt * 16 + lg * 4 + j) * LDP_H + nt * 16 + lr] = (__bf16)s[mt][nt][j];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[4], b[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(Ps + (mt * 16 + lr) * LDP_H + ks * 32 + lg * 8);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(Vt + (nt * 16 + lr) * LDP_H + ks * 32 + lg * 8);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) o[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], o[mt][nt], 0, 0, 0);
    }
  }
  if (active) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = mt * 16 + lg * 4 + j;
        if (q < WT) {
          AT* dst = p.out + (size_t)tm.row(q) * p.C + colq;
          stf(dst + lr, o[mt][0][j]);
          stf(dst + 16 + lr, o[mt][1][j]);
        }
      }
  }
}

// backward (exact fp32 MFMA): recompute P, then dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dP o P)),
// dQ = scale * dS K, dK = dS^T (scale*Q), dtable[idx] += dS.  Two waves per workgroup; a wave walks
// tasks_per_wave windows of ONE head so its relative-position-bias gradient is reduced in LDS first.
constexpr int BWD_WAVE_FLOATS = 4 * 64 * LDQ_F + 64 * LDP_F + 176 + 176;

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void win_attn_bwd_kernel(const WinArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BWD_WAVE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int head = blockIdx.y;
  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  float* Qs = smem + wave * BWD_WAVE_FLOATS;
  float* Ks = Qs + 64 * LDQ_F;
  float* Vs = Ks + 64 * LDQ_F;
  float* Ds = Vs + 64 * LDQ_F;
  float* PS = Ds + 64 * LDQ_F;
  float* bt = PS + 64 * LDP_F;
  float* dbt = bt + 176;
  for (int i = lane; i < 176; i += 64) { bt[i] = i < 169 ? p.table[i * p.heads + head] : 0.f; dbt[i] = 0.f; }
  const int ld = 3 * p.C, colq = head * HD;
  const long long task0 = ((long long)blockIdx.x * 2 + wave) * p.tasks_per_wave;

  for (int tt = 0; tt < p.tasks_per_wave; ++tt) {
    long long task = task0 + tt;
    const bool active = task < p.ntasks;   // inactive waves keep running (barriers) on a clamped task, stores suppressed
    if (!active) task = p.ntasks - 1;
    TokMap tm;
    tm.img = (int)(task / nW); const int win = (int)(task - (long long)tm.img * nW);
    tm.wy = win / nWx; tm.wx = win - tm.wy * nWx; tm.H = p.H; tm.W = p.W; tm.shift = p.shift;

    __syncthreads();  // previous task's LDS reads done
    load_tile_f32(Qs, p.qkv, ld, colq, tm, lane, p.scale);
    load_tile_f32(Ks, p.qkv, ld, p.C + colq, tm, lane, 1.f);
    load_tile_f32(Vs, p.qkv, ld, 2 * p.C + colq, tm, lane, 1.f);
    load_tile_f32(Ds, p.dout, p.C, colq, tm, lane, 1.f);
    __syncthreads();

    f32x4 s[4][4], dp[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { s[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < HD / 4; ++kk) {
      float a[4], b[4], c[4], d[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int off = (t * 16 + lr) * LDQ_F + kk * 4 + lg;
        a[t] = Qs[off]; b[t] = Ks[off]; c[t] = Ds[off]; d[t] = Vs[off];
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          s[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], s[mt][nt], 0, 0, 0);
          dp[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(c[mt], d[nt], dp[mt][nt], 0, 0, 0);
        }
    }
    bias_mask_softmax(s, bt, tm, lane, p.shift > 0);
    // dS = P o (dP - rowsum(dP o P)); kept in dp
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float r = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) r += dp[mt][nt][j] * s[mt][nt][j];
        r = group16_sum(r);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dp[mt][nt][j] = s[mt][nt][j] * (dp[mt][nt][j] - r);
      }
    // relative-position-bias gradient into the wave's LDS table
    if (active) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = mt * 16 + lg * 4 + j;
          if (q < WT) {
            const int qy = q / 7, qx = q - qy * 7;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              const int key = nt * 16 + lr;
              if (key < WT) {
                const int ky = key / 7, kx = key - ky * 7;
                atomicAdd(dbt + (qy - ky + 6) * 13 + (qx - kx + 6), dp[mt][nt][j]);
              }
            }
          }
        }
    }
    // ---- dV = P^T dO
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) PS[(mt * 16 + lg * 4 + j) * LDP_F + nt * 16 + lr] = s[mt][nt][j];
    __syncthreads();
    {
      f32x4 acc[4][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        float a[4], b[2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = PS[(kk * 4 + lg) * LDP_F + mt * 16 + lr];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[nt] = Ds[(kk * 4 + lg) * LDQ_F + nt * 16 + lr];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
      }
      if (active) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int key = mt * 16 + lg * 4 + j;
            if (key < WT) {
              float* dst = p.dqkv + (size_t)tm.row(key) * ld + 2 * p.C + colq;
              dst[lr] = acc[mt][0][j]; dst[16 + lr] = acc[mt][1][j];
            }
          }
      }
    }
    __syncthreads();
    // ---- dQ = scale * dS K ; dK = dS^T (scale*Q)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) PS[(mt * 16 + lg * 4 + j) * LDP_F + nt * 16 + lr] = dp[mt][nt][j];
    __syncthreads();
    {
      f32x4 aq[4][2], ak[4][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        aq[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; aq[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ak[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; ak[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        float a1[4], a2[4], b1[2], b2[2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          a1[mt] = PS[(mt * 16 + lr) * LDP_F + kk * 4 + lg];   // dS[q][key]
          a2[mt] = PS[(kk * 4 + lg) * LDP_F + mt * 16 + lr];   // dS^T[key][q]
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          b1[nt] = Ks[(kk * 4 + lg) * LDQ_F + nt * 16 + lr];
          b2[nt] = Qs[(kk * 4 + lg) * LDQ_F + nt * 16 + lr];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            aq[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mt], b1[nt], aq[mt][nt], 0, 0, 0);
            ak[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[mt], b2[nt], ak[mt][nt], 0, 0, 0);
          }
      }
      if (active) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int t = mt * 16 + lg * 4 + j;
            if (t < WT) {
              float* dst = p.dqkv + (size_t)tm.row(t) * ld + colq;
              dst[lr] = aq[mt][0][j] * p.scale; dst[16 + lr] = aq[mt][1][j] * p.scale;
              dst[p.C + lr] = ak[mt][0][j]; dst[p.C + 16 + lr] = ak[mt][1][j];
            }
          }
      }
    }
  }
  __syncthreads();
  for (int i = lane; i < 169; i += 64) {
    const float v = dbt[i];
    if (v != 0.f) atomicAdd(p.dtable + i * p.heads + head, v);
  }
}

// bf16 backward: same algorithm, bf16 MFMA (16x16x32) with fp32 accumulate; the tiles stay row-major [token][32] /
// [q][key] in LDS and every fragment that needs 8 consecutive TOKENS per lane (P^T, dS^T, and the B operands K, Q, dO of
// dQ / dK / dV) is produced by ds_read_b64_tr_b16, so no transposed copies exist.  15.6 KB + 9.2 KB per wave -> 4 waves.
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_a;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* tile, int ld, int row0, int col0, int lane) {
  // 8 consecutive rows (row0 + 8*(lane>>4) + j) of column (col0 + (lane&15)): two 4x16 transposing reads
  const int lr = lane & 15, lg = lane >> 4, q = lr >> 2, pp = lr & 3;
  const __bf16* src = tile + (row0 + lg * 8 + q) * ld + col0 + pp * 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src + 4 * ld));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// the same fragment with the columns of a 32-wide head tile permuted: MFMA row lr <- column 8 (lr >> 2) + 4 nt + (lr & 3), so that the two
// 16-column blocks nt = 0, 1 of a transposed accumulator leave the 8 CONSECUTIVE columns 8 lg .. 8 lg + 7 of a row in one lane (16-byte stores)
__device__ __forceinline__ bf16x8 tr_frag_perm(const __bf16* tile, int ld, int row0, int nt, int lane) {
  const int lr = lane & 15, lg = lane >> 4;
  const __bf16* src = tile + (row0 + lg * 8 + (lr >> 2)) * ld + (lr & 3) * 8 + nt * 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src + 4 * ld));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ------------------------------------------------------------------------------------------------
// cross-view attention core.  qkv: [B*V*P, 3R] channels-last rows (img, pos), R = heads*hd reduced channels;
// feature of (view, head) = all (pos, c) with c in the head's hd channels -> F = P*hd (288).
// ------------------------------------------------------------------------------------------------
constexpr int CVA_MAXV = 32, CVA_MAXF = 288;

template <typename AT>
struct CvaArgsT { const AT* qkv; AT* out; const AT* dout; AT* dqkv; int B, V, P, R, heads, hd, fc; float scale; };

template <typename AT>
__device__ __forceinline__ size_t cva_off(const CvaArgsT<AT>& p, int b, int v, int f, int head, int which, int ld) {
  const int pos = f / p.hd, c = f - pos * p.hd;
  return ((size_t)(b * p.V + v) * p.P + pos) * ld + which * p.R + head * p.hd + c;
}

// Features are walked in chunks of FC = p.fc <= CVA_MAXF per view (whole positions; one chunk for the 3 x 3 grid of the default
// ATT_SPATIAL_DOWNSAMPLE_RATIO = 2, six for the 7 x 7 grid of ratio 1): the V x V scores accumulate over the chunks, the second sweep
// re-loads only when there was more than one chunk.
template <typename AT>
__global__ __launch_bounds__(256) void cva_attn_fwd_kernel(const CvaArgsT<AT> p) {
  __shared__ float q[CVA_MAXV * CVA_MAXF], k[CVA_MAXV * CVA_MAXF], v[CVA_MAXV * CVA_MAXF];
  __shared__ float a[CVA_MAXV * CVA_MAXV];
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  const int F = p.P * p.hd, V = p.V, FC = p.fc;
  const bool single = F <= FC;
  for (int ij = threadIdx.x; ij < V * V; ij += 256) a[ij] = 0.f;
  for (int f0 = 0; f0 < F; f0 += FC) {
    const int fc = min(FC, F - f0);
    __syncthreads();
    for (int i = threadIdx.x; i < V * fc; i += 256) {
      const int vi = i / fc, f = i - vi * fc;
      q[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 0, 3 * p.R));
      k[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 1, 3 * p.R));
      if (single) v[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 2, 3 * p.R));
    }
    __syncthreads();
    for (int ij = threadIdx.x; ij < V * V; ij += 256) {
      const int i = ij / V, j = ij - i * V;
      float s = 0.f;
      for (int f = 0; f < fc; ++f) s += q[i * fc + f] * k[j * fc + f];
      a[ij] += s;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V; i += 256) {
    float mx = -3.0e38f;
    for (int j = 0; j < V; ++j) { a[i * V + j] *= p.scale; mx = fmaxf(mx, a[i * V + j]); }
    float sum = 0.f;
    for (int j = 0; j < V; ++j) { const float e = expf(a[i * V + j] - mx); a[i * V + j] = e; sum += e; }
    const float inv = 1.f / sum;
    for (int j = 0; j < V; ++j) a[i * V + j] *= inv;
  }
  __syncthreads();
  for (int f0 = 0; f0 < F; f0 += FC) {
    const int fc = min(FC, F - f0);
    if (!single) {
      __syncthreads();
      for (int i = threadIdx.x; i < V * fc; i += 256) {
        const int vi = i / fc, f = i - vi * fc;
        v[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 2, 3 * p.R));
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < V * fc; i += 256) {
      const int vi = i / fc, f = i - vi * fc;
      float o = 0.f;
      for (int j = 0; j < V; ++j) o += a[vi * V + j] * v[j * fc + f];
      stf(p.out + cva_off(p, b, vi, f0 + f, head, 0, p.R), o);
    }
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void cva_attn_bwd_kernel(const CvaArgsT<AT> p) {
  __shared__ float q[CVA_MAXV * CVA_MAXF], k[CVA_MAXV * CVA_MAXF], v[CVA_MAXV * CVA_MAXF], d[CVA_MAXV * CVA_MAXF];
  __shared__ float a[CVA_MAXV * CVA_MAXV], ds[CVA_MAXV * CVA_MAXV];
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  const int F = p.P * p.hd, V = p.V, FC = p.fc;
  const bool single = F <= FC;
  for (int ij = threadIdx.x; ij < V * V; ij += 256) { a[ij] = 0.f; ds[ij] = 0.f; }
  for (int f0 = 0; f0 < F; f0 += FC) {
    const int fc = min(FC, F - f0);
    __syncthreads();
    for (int i = threadIdx.x; i < V * fc; i += 256) {
      const int vi = i / fc, f = i - vi * fc;
      q[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 0, 3 * p.R));
      k[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 1, 3 * p.R));
      v[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 2, 3 * p.R));
      d[i] = ldf(p.dout + cva_off(p, b, vi, f0 + f, head, 0, p.R));
    }
    __syncthreads();
    for (int ij = threadIdx.x; ij < V * V; ij += 256) {
      const int i = ij / V, j = ij - i * V;
      float s = 0.f, da = 0.f;
      for (int f = 0; f < fc; ++f) { s += q[i * fc + f] * k[j * fc + f]; da += d[i * fc + f] * v[j * fc + f]; }
      a[ij] += s; ds[ij] += da;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V; i += 256) {
    float mx = -3.0e38f;
    for (int j = 0; j < V; ++j) { a[i * V + j] *= p.scale; mx = fmaxf(mx, a[i * V + j]); }
    float sum = 0.f;
    for (int j = 0; j < V; ++j) { const float e = expf(a[i * V + j] - mx); a[i * V + j] = e; sum += e; }
    const float inv = 1.f / sum;
    float r = 0.f;
    for (int j = 0; j < V; ++j) { a[i * V + j] *= inv; r += a[i * V + j] * ds[i * V + j]; }
    for (int j = 0; j < V; ++j) ds[i * V + j] = a[i * V + j] * (ds[i * V + j] - r) * p.scale;
  }
  __syncthreads();
  for (int f0 = 0; f0 < F; f0 += FC) {
    const int fc = min(FC, F - f0);
    if (!single) {
      __syncthreads();
      for (int i = threadIdx.x; i < V * fc; i += 256) {
        const int vi = i / fc, f = i - vi * fc;
        q[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 0, 3 * p.R));
        k[i] = ldf(p.qkv + cva_off(p, b, vi, f0 + f, head, 1, 3 * p.R));
        d[i] = ldf(p.dout + cva_off(p, b, vi, f0 + f, head, 0, p.R));
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < V * fc; i += 256) {
      const int vi = i / fc, f = i - vi * fc;
      float dq = 0.f, dk = 0.f, dv = 0.f;
      for (int j = 0; j < V; ++j) {
        dq += ds[vi * V + j] * k[j * fc + f];
        dk += ds[j * V + vi] * q[j * fc + f];
        dv += a[j * V + vi] * d[j * fc + f];
      }
      stf(p.dqkv + cva_off(p, b, vi, f0 + f, head, 0, 3 * p.R), dq);
      stf(p.dqkv + cva_off(p, b, vi, f0 + f, head, 1, 3 * p.R), dk);
      stf(p.dqkv + cva_off(p, b, vi, f0 + f, head, 2, 3 * p.R), dv);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Workgroup-per-window kernels (bf16 MFMA): the 4 waves of a workgroup split the 64 (49 used) query rows of ONE
// (window, head) task - 16 rows each - and the key rows of the transposed products.  Per-wave state is a 16 x 64 score
// strip (16 accumulator registers) instead of the whole 64 x 64 matrix, so 4-6 workgroups fit a CU (LDS 25 / 41 KB) and
// every SIMD holds 4+ waves to hide LDS / HBM latency (the wave-per-window kernels above run 1 wave per SIMD at 256 VGPRs
// and are kept for exact-fp32 MFMA).  A workgroup walks `tasks_per_wave` windows of one head; the relative-position-bias
// gradient is accumulated in REGISTERS across its windows (the (query, key) -> table index map is the same for all) and
// reaches LDS / HBM once per workgroup.
// ------------------------------------------------------------------------------------------------
// bias + mask + softmax over the keys for ONE 16-row strip (query rows mt*16 + lg*4 + j, keys nt*16 + lr).
// The relative-position bias of this lane's 16 (query, key) slots is the same for every window of a head: it is gathered
// once per workgroup into registers (strip_bias); the shifted-window mask only exists in the last window row / column.
__device__ __forceinline__ void strip_bias(float (&bias)[4][4], const float* bt, int lane, int mt, float mul = 1.f) {
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int key = nt * 16 + lr, ky = key / 7, kx = key - ky * 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = mt * 16 + lg * 4 + j, qy = q / 7, qx = q - qy * 7;
      bias[nt][j] = key >= WT ? -1.0e30f : (q < WT ? mul * bt[(qy - ky + 6) * 13 + (qx - kx + 6)] : 0.f);   // key padding: excluded
    }
  }
}
// LOG2: scores and bias arrive multiplied by log2(e) (the forward folds the factor into the scale of Q and into the bias strip), so the
// exponential is one v_exp_f32 without the multiply __expf carries
constexpr float ATTN_LOG2E = 1.4426950408889634f;
template <bool LOG2 = false>
__device__ __forceinline__ void bias_mask_softmax_strip(f32x4 (&s)[4], const float (&bias)[4][4], const TokMap& tm, int lane, bool masked, int mt) {
  const int lr = lane & 15, lg = lane >> 4;
  if (masked) {   // a window touching the rolled seam: tokens of different regions must not attend to each other
    int kreg[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { const int key = nt * 16 + lr; kreg[nt] = key < WT ? tm.region(key) : 0; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = mt * 16 + lg * 4 + j;
      const int qreg = q < WT ? tm.region(q) : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        if (q < WT && nt * 16 + lr < WT && qreg != kreg[nt]) s[nt][j] += LOG2 ? -100.0f * ATTN_LOG2E : -100.0f;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float mx = -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float v = bias[nt][j] <= -1.0e29f ? -1.0e30f : s[nt][j] + bias[nt][j];
      s[nt][j] = v;
      mx = fmaxf(mx, v);
    }
    mx = group16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { const float e = LOG2 ? __builtin_amdgcn_exp2f(s[nt][j] - mx) : __expf(s[nt][j] - mx); s[nt][j] = e; sum += e; }
    const float inv = __builtin_amdgcn_rcpf(group16_sum(sum));   // 1 ulp: the quotient is rounded to bf16 right after (an IEEE division is ~10 instructions)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) s[nt][j] *= inv;
  }
}

__device__ __forceinline__ float4 to_f4(float4 v) { return v; }
__device__ __forceinline__ float4 to_f4(bf16x4 b) { return make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]); }

// cooperative (256 threads) load of one [49(64) x 32] head slice into a bf16 LDS tile [64][LDQ_H]; rows >= 49 zeroed
template <typename AT>
__device__ __forceinline__ void load_tile_wg(__bf16* dst, const AT* src, int ld, int col, const TokMap& tm, int tid, float mul) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = (tid >> 3) + 32 * it, ch = (tid & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < WT) v = ld4f(src + (size_t)tm.row(r) * ld + col + ch);
    bf16x4 b;
    b[0] = (__bf16)(v.x * mul); b[1] = (__bf16)(v.y * mul); b[2] = (__bf16)(v.z * mul); b[3] = (__bf16)(v.w * mul);
    *reinterpret_cast<bf16x4*>(dst + r * LDQ_H + ch) = b;
  }
}

__device__ __forceinline__ TokMap task_map(long long task, int nW, int nWx, int H, int W, int shift) {
  TokMap tm;
  tm.img = (int)(task / nW); const int win = (int)(task - (long long)tm.img * nW);
  tm.wy = win / nWx; tm.wx = win - tm.wy * nWx; tm.H = H; tm.W = W; tm.shift = shift;
  return tm;
}

// (window chunk, head) of a workgroup in the 1-D grids of the workgroup-per-window kernels.  A head's slice of a token row is
// 64 bytes - half a cache line - so the `heads` workgroups that walk the same windows must meet in one L2: consecutive
// linear ids are spread round-robin over the 8 XCDs, ids l, l+8, l+16, ... therefore land on the SAME XCD back to back.
// Those carry the heads of one chunk; the row is fetched from HBM once and the other heads hit in that XCD's L2.
// Measured (I = 256): 8 % faster at 3 heads, 2-4 % at 6, and 4 % slower at 12+ heads (a token row is then >= 6 lines and
// the heads of a chunk crowd one XCD), so wide stages keep the head-major order.
__device__ __forceinline__ bool wg_chunk_head(int nchunks, int heads, int& chunk, int& head) {
  const int l = blockIdx.x;
  if (heads <= 6) {
    const int xcd = l & 7, slot = l >> 3;
    head = slot % heads;
    chunk = (slot / heads) * 8 + xcd;
  } else {
    const int n8 = (nchunks + 7) & ~7;
    head = l / n8;
    chunk = l - head * n8;
  }
  return chunk < nchunks;
}
// Windows per workgroup such that the whole grid is ONE resident wave (256 CUs x the kernel's workgroups per CU): every
// workgroup does the same work, so a partially filled second wave costs as much as a full one.
static inline int wg_tasks_per_block(int ntasks, int heads, int blocks_per_cu) {
  int nchunks = 256 * blocks_per_cu / heads;
  if (nchunks < 1) nchunks = 1;
  if (nchunks > ntasks) nchunks = ntasks;
  return (ntasks + nchunks - 1) / nchunks;
}
static inline unsigned wg_grid(int nchunks, int heads) { return 8u * (unsigned)heads * (unsigned)((nchunks + 7) / 8); }

// A thread's piece of a [64 tokens][32 channels] head tile in the cooperative (256-thread) loads: CV consecutive channels of one token row -
// 16 bytes of bf16 rows (one pass over the tile), 4 floats of fp32 rows (two passes)
template <typename AT> struct HeadChunk {
  static constexpr int CV = sizeof(AT) == 2 ? 8 : 4, TPRW = HD / CV, NIT = 64 * TPRW / 256;
  typedef typename VecN<AT, CV>::type RV;
  static __device__ __forceinline__ int row(int tid, int it) { return tid / TPRW + (256 / TPRW) * it; }
  static __device__ __forceinline__ int ch(int tid) { return (tid % TPRW) * CV; }
};
__device__ __forceinline__ void put_chunk(__bf16* dst, const float4& v, float mul) {
  bf16x4 b;
  b[0] = (__bf16)(v.x * mul); b[1] = (__bf16)(v.y * mul); b[2] = (__bf16)(v.z * mul); b[3] = (__bf16)(v.w * mul);
  *reinterpret_cast<bf16x4*>(dst) = b;
}
__device__ __forceinline__ void put_chunk(__bf16* dst, const bf16x8& v, float mul) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)((float)v[j] * mul);
  *reinterpret_cast<bf16x8*>(dst) = b;
}

template <typename AT>
__global__ __launch_bounds__(256, 4) void win_attn_fwd_wg_kernel(const WinArgsT<AT> p) {
  // q (scaled) | k | v of the window's head, row-major [token][32 (+8)] bf16.  The products are taken with swapped operands (keys / V^T as
  // the first MFMA operand): S^T = K Q^T leaves the 16 keys (16 nt + 4 lg + j) of query lr in one lane, so the softmax of a query is 16
  // in-lane values + two lane-group exchanges, and the probabilities go into P V from the registers they are in (the contraction index of
  // an MFMA may be permuted freely as long as both operands agree: the V^T fragment is fetched with ds_read_b64_tr_b16 from exactly those
  // key rows) - no P tile, no transposed V copy, one barrier less per window than the first version of this kernel.
  __shared__ __attribute__((aligned(16))) __bf16 Qs[64 * LDQ_H], Ks[64 * LDQ_H], Vs[64 * LDQ_H];
  __shared__ float bt[176];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  int chunk, head;
  if (!wg_chunk_head((p.ntasks + p.tasks_per_wave - 1) / p.tasks_per_wave, p.heads, chunk, head)) return;   // uniform over the workgroup
  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  const int ld = 3 * p.C, colq = head * HD;
  for (int i = tid; i < 169; i += 256) bt[i] = p.table[i * p.heads + head] * ATTN_LOG2E;
  __syncthreads();
  const int q = wave * 16 + lr;                      // the query of this lane
  const bool qok = q < WT;
  float bias[4][4];                                  // relative-position bias of this lane's 16 (query, key) slots: the same for every window
  {
    const int qy = q / 7, qx = q - qy * 7;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = nt * 16 + lg * 4 + j, ky = key / 7, kx = key - ky * 7;
        bias[nt][j] = key >= WT ? -1.0e30f : (qok ? bt[(qy - ky + 6) * 13 + (qx - kx + 6)] : 0.f);   // key padding: excluded
      }
  }
  const float qscale = p.scale * ATTN_LOG2E;      // scores come out of the MFMA in log2 units
  const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
  const auto fadd2 = [](float a, float b) { return a + b; };
  const long long task0 = (long long)chunk * p.tasks_per_wave;
  for (int tt = 0; tt < p.tasks_per_wave; ++tt) {
    const long long task = task0 + tt;
    if (task >= p.ntasks) break;                                         // uniform over the workgroup
    const TokMap tm = task_map(task, nW, nWx, p.H, p.W, p.shift);
    __syncthreads();                                                     // previous window's tiles are consumed
    typedef HeadChunk<AT> HC;
#pragma unroll
    for (int it = 0; it < HC::NIT; ++it) {
      const int r = HC::row(tid, it), ch = HC::ch(tid);
      typename HC::RV q4 = VecN<AT, HC::CV>::zero(), k4 = q4, v4 = q4;   // rows >= 49: zeros (finite values under zero probabilities)
      if (r < WT) {
        const AT* src = p.qkv + (size_t)tm.row(r) * ld + colq + ch;
        q4 = VecN<AT, HC::CV>::load(src); k4 = VecN<AT, HC::CV>::load(src + p.C); v4 = VecN<AT, HC::CV>::load(src + 2 * p.C);
      }
      put_chunk(Qs + r * LDQ_H + ch, q4, qscale);
      put_chunk(Ks + r * LDQ_H + ch, k4, 1.f);
      put_chunk(Vs + r * LDQ_H + ch, v4, 1.f);
    }
    __syncthreads();
    f32x4 s[4];
    {
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + q * LDQ_H + lg * 8);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (nt * 16 + lr) * LDQ_H + lg * 8);
        s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // lane: query lr, keys 16 nt + 4 lg + j
      }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[nt][j] += bias[nt][j]; mx = fmaxf(mx, s[nt][j]); }
    if (p.shift > 0 && (tm.wy == p.H / 7 - 1 || tm.wx == nWx - 1)) {    // a window touching the rolled seam: regions must not attend to each other
      const int qreg = qok ? tm.region(q) : 0;
      mx = -3.0e38f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = nt * 16 + lg * 4 + j;
          if (qok && key < WT && tm.region(key) != qreg) s[nt][j] += -100.0f * ATTN_LOG2E;
          mx = fmaxf(mx, s[nt][j]);
        }
    }
    mx = lanegroup_allreduce(mx, fmax2);
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float e = __builtin_amdgcn_exp2f(s[nt][j] - mx); s[nt][j] = e; sum += e; }
    const float inv = __builtin_amdgcn_rcpf(lanegroup_allreduce(sum, fadd2));   // 1 ulp: the quotient is rounded to bf16 right after
    bf16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) { pf[ks][j] = (__bf16)(s[2 * ks][j] * inv); pf[ks][4 + j] = (__bf16)(s[2 * ks + 1][j] * inv); }
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        // V^T fragment of block nt: MFMA row lr <- head channel 8 (lr >> 2) + 4 nt + (lr & 3) (the lanes with lr & 3 = c supply the 4-column
        // group that the lanes with lr >> 2 = c receive), keys 32 ks + 4 lg .. + 3 and 32 ks + 16 + 4 lg .. + 3: the order of the P fragment
        const __bf16* src = Vs + (2 * ks * 16 + lg * 4 + (lr >> 2)) * LDQ_H + (lr & 3) * 8 + nt * 4;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src + 16 * LDQ_H));
        const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks], o[nt], 0, 0, 0);
      }
    if (qok) {   // lane: query lr, head channels 8 lg .. 8 lg + 7 (block nt holds 8 lg + 4 nt + j): one 16-byte vector of bf16 rows
      const float ov[8] = {o[0][0], o[0][1], o[0][2], o[0][3], o[1][0], o[1][1], o[1][2], o[1][3]};
      stnf<8>(p.out + (size_t)tm.row(q) * p.C + colq + lg * 8, ov);
    }
  }
}

// ---- fp8 (OCP e4m3) variant of the forward core: BASELINE configuration 5 ("fp8 MFMA QK/AV") -------------------------------------
// q (already times head_dim^-0.5), k, v and the softmax probabilities go to the matrix pipe as e4m3 (v_mfma_f32_16x16x32_fp8_fp8,
// fp32 accumulate).  Scales are per (window, head) TILE - finer than per tensor and free of an extra pass over HBM: the workgroup takes
// the absolute maximum of its q / k / v tile while the rows travel through registers and maps it to 224 (half of e4m3's 448, head room
// for the rounding); probabilities (<= 1) are scaled by 256, so that everything above 2^-17 survives as a subnormal.  Scores and
// outputs are rescaled in fp32; bias, mask and softmax stay fp32.  The backward keeps bf16 operands and recomputes P from bf16 q / k.
constexpr int LDQ_8 = 40;    // byte row stride of the fp8 q / k tiles (32 + 8: 8-byte fragment reads, rows 10 banks apart)
constexpr int LDP_8 = 72;    // byte row stride of the fp8 P and V^T tiles
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ uint8_t one_fp8(float a) { return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, 0.f, 0, false) & 0xff); }

template <typename AT>
__global__ __launch_bounds__(256, 4) void win_attn_fwd_wg_fp8_kernel(const WinArgsT<AT> p) {
  __shared__ __attribute__((aligned(16))) uint8_t Qs[64 * LDQ_8], Ks[64 * LDQ_8], Vt[HD * LDP_8], Ps[64 * LDP_8];
  __shared__ float bt[176];
  __shared__ float amax[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  int chunk, head;
  if (!wg_chunk_head((p.ntasks + p.tasks_per_wave - 1) / p.tasks_per_wave, p.heads, chunk, head)) return;   // uniform over the workgroup
  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  const int ld = 3 * p.C, colq = head * HD;
  for (int i = tid; i < 169; i += 256) bt[i] = p.table[i * p.heads + head];
  for (int i = tid; i < HD * LDP_8; i += 256) Vt[i] = 0;               // key columns >= 49 stay zero
  __syncthreads();
  float bias[4][4];
  strip_bias(bias, bt, lane, wave);
  const long long task0 = (long long)chunk * p.tasks_per_wave;
  for (int tt = 0; tt < p.tasks_per_wave; ++tt) {
    const long long task = task0 + tt;
    if (task >= p.ntasks) break;                                         // uniform over the workgroup
    const TokMap tm = task_map(task, nW, nWx, p.H, p.W, p.shift);
    float4 q4[2], k4[2], v4[2];
    float mq = 0.f, mk = 0.f, mv = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = (tid >> 3) + 32 * it, ch = (tid & 7) * 4;
      q4[it] = make_float4(0.f, 0.f, 0.f, 0.f); k4[it] = q4[it]; v4[it] = q4[it];
      if (r < WT) {
        const AT* src = p.qkv + (size_t)tm.row(r) * ld + colq + ch;
        q4[it] = ld4f(src); k4[it] = ld4f(src + p.C); v4[it] = ld4f(src + 2 * p.C);
      }
      q4[it].x *= p.scale; q4[it].y *= p.scale; q4[it].z *= p.scale; q4[it].w *= p.scale;
      mq = fmaxf(mq, fmaxf(fmaxf(fabsf(q4[it].x), fabsf(q4[it].y)), fmaxf(fabsf(q4[it].z), fabsf(q4[it].w))));
      mk = fmaxf(mk, fmaxf(fmaxf(fabsf(k4[it].x), fabsf(k4[it].y)), fmaxf(fabsf(k4[it].z), fabsf(k4[it].w))));
      mv = fmaxf(mv, fmaxf(fmaxf(fabsf(v4[it].x), fabsf(v4[it].y)), fmaxf(fabsf(v4[it].z), fabsf(v4[it].w))));
    }
    mq = wave_max(mq); mk = wave_max(mk); mv = wave_max(mv);
    __syncthreads();                                                     // previous window's tiles (and maxima) are consumed
    if (lane == 0) { amax[wave][0] = mq; amax[wave][1] = mk; amax[wave][2] = mv; }
    __syncthreads();
    mq = fmaxf(fmaxf(amax[0][0], amax[1][0]), fmaxf(amax[2][0], amax[3][0]));
    mk = fmaxf(fmaxf(amax[0][1], amax[1][1]), fmaxf(amax[2][1], amax[3][1]));
    mv = fmaxf(fmaxf(amax[0][2], amax[1][2]), fmaxf(amax[2][2], amax[3][2]));
    const float sq = mq > 0.f ? 224.f / mq : 1.f, sk = mk > 0.f ? 224.f / mk : 1.f, sv = mv > 0.f ? 224.f / mv : 1.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = (tid >> 3) + 32 * it, ch = (tid & 7) * 4;
      *reinterpret_cast<uint32_t*>(Qs + r * LDQ_8 + ch) = pack4_fp8(q4[it].x * sq, q4[it].y * sq, q4[it].z * sq, q4[it].w * sq);
      *reinterpret_cast<uint32_t*>(Ks + r * LDQ_8 + ch) = pack4_fp8(k4[it].x * sk, k4[it].y * sk, k4[it].z * sk, k4[it].w * sk);
      if (r < WT) {
        const uint32_t w = pack4_fp8(v4[it].x * sv, v4[it].y * sv, v4[it].z * sv, v4[it].w * sv);
        Vt[(ch + 0) * LDP_8 + r] = (uint8_t)(w & 0xff); Vt[(ch + 1) * LDP_8 + r] = (uint8_t)((w >> 8) & 0xff);
        Vt[(ch + 2) * LDP_8 + r] = (uint8_t)((w >> 16) & 0xff); Vt[(ch + 3) * LDP_8 + r] = (uint8_t)(w >> 24);
      }
    }
    __syncthreads();
    f32x4 s[4];
    {
      const long a = *reinterpret_cast<const long*>(Qs + (wave * 16 + lr) * LDQ_8 + lg * 8);
      const float un = 1.f / (sq * sk);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const long b = *reinterpret_cast<const long*>(Ks + (nt * 16 + lr) * LDQ_8 + lg * 8);
        s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[nt][j] *= un;
      }
    }
    bias_mask_softmax_strip(s, bias, tm, lane, p.shift > 0 && (tm.wy == p.H / 7 - 1 || tm.wx == nWx - 1), wave);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) Ps[(wave * 16 + lg * 4 + j) * LDP_8 + nt * 16 + lr] = one_fp8(s[nt][j] * 256.f);
    __syncthreads();
    f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const long a = *reinterpret_cast<const long*>(Ps + (wave * 16 + lr) * LDP_8 + ks * 32 + lg * 8);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const long b = *reinterpret_cast<const long*>(Vt + (nt * 16 + lr) * LDP_8 + ks * 32 + lg * 8);
        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b, a, o[nt], 0, 0, 0);   // transposed block: O[q = lr][d = lg*4 + j]
      }
    }
    {
      const int q = wave * 16 + lr;
      const float un = 1.f / (256.f * sv);
      if (q < WT) {
        AT* dst = p.out + (size_t)tm.row(q) * p.C + colq + lg * 4;
        st4f(dst, make_float4(o[0][0] * un, o[0][1] * un, o[0][2] * un, o[0][3] * un));
        st4f(dst + 16, make_float4(o[1][0] * un, o[1][1] * un, o[1][2] * un, o[1][3] * un));
      }
    }
  }
}

constexpr int ATTN_DT_SLOTS = 16;   // slot images of the relative-position-bias gradient (see sv_window_attention_bwd)

template <typename AT>
__global__ __launch_bounds__(256, 3) void win_attn_bwd_wg_kernel(const WinArgsT<AT> p, float* __restrict__ dt_ws) {
  __shared__ __attribute__((aligned(16))) __bf16 Qs[64 * LDQ_H], Ks[64 * LDQ_H], Vs[64 * LDQ_H], Ds[64 * LDQ_H];
  __shared__ __attribute__((aligned(16))) __bf16 Ps[64 * LDP_H], Ss[64 * LDP_H];   // P and dS, [query][key]
  __shared__ float bt[176], dbt[176];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  int chunk, head;
  if (!wg_chunk_head((p.ntasks + p.tasks_per_wave - 1) / p.tasks_per_wave, p.heads, chunk, head)) return;   // uniform over the workgroup
  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  const int ld = 3 * p.C, colq = head * HD;
  for (int i = tid; i < 176; i += 256) { bt[i] = i < 169 ? p.table[i * p.heads + head] : 0.f; dbt[i] = 0.f; }
  __syncthreads();
  // Score blocks are taken transposed (keys as the first MFMA operand): lane -> query 16 wave + lr, keys 16 nt + 4 lg + j.  The softmax and
  // the row sum of dP o P of a query are then 16 in-lane values + two lane-group exchanges (the first version reduced every accumulator
  // row over 16 lanes), P and dS reach their LDS tiles as 8-byte row vectors instead of 2-byte scatters, and dQ = dS K takes dS from the
  // registers it is in (K^T fragments by ds_read_b64_tr_b16 in the key order of those registers, as the forward kernel does with V).
  const int qrow = wave * 16 + lr;
  const bool qrow_ok = qrow < WT;
  float bias[4][4];
  {
    const int qy = qrow / 7, qx = qrow - qy * 7;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = nt * 16 + lg * 4 + j, ky = key / 7, kx = key - ky * 7;
        bias[nt][j] = key >= WT ? -1.0e30f : (qrow_ok ? bt[(qy - ky + 6) * 13 + (qx - kx + 6)] : 0.f);   // key padding: excluded
      }
  }
  const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
  const auto fadd2 = [](float a, float b) { return a + b; };
  f32x4 dsum[4];                                   // sum over this workgroup's windows of dS at this lane's (query, key) slots
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) dsum[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long task0 = (long long)chunk * p.tasks_per_wave;
  // the next window's q / k / v / dO vectors travel in registers while this window is contracted: a window is
  // load -> barrier -> MFMA -> softmax -> barrier -> MFMA -> store, and three workgroups per CU do not hide a global round trip
  typedef HeadChunk<AT> HC;
  typedef typename HC::RV RV;
  RV rq[HC::NIT], rk[HC::NIT], rv[HC::NIT], rd[HC::NIT];
  auto fetch = [&](long long task) {
    const TokMap tn = task_map(task, nW, nWx, p.H, p.W, p.shift);
#pragma unroll
    for (int it = 0; it < HC::NIT; ++it) {
      const int r = HC::row(tid, it), ch = HC::ch(tid);
      rq[it] = VecN<AT, HC::CV>::zero(); rk[it] = rq[it]; rv[it] = rq[it]; rd[it] = rq[it];
#ifdef SV_AT_PROBE_NOFETCH
      if (r < 0) {
#else
      if (r < WT) {
#endif
        const size_t row = tn.row(r);
        const AT* src = p.qkv + row * ld + colq + ch;
        rq[it] = VecN<AT, HC::CV>::load(src); rk[it] = VecN<AT, HC::CV>::load(src + p.C); rv[it] = VecN<AT, HC::CV>::load(src + 2 * p.C);
        rd[it] = VecN<AT, HC::CV>::load(p.dout + row * p.C + colq + ch);
      }
    }
  };
  auto put = [&](__bf16* dst, const RV& v, int it, float mul) { put_chunk(dst + HC::row(tid, it) * LDQ_H + HC::ch(tid), v, mul); };
  if (task0 < p.ntasks) fetch(task0);
  for (int tt = 0; tt < p.tasks_per_wave; ++tt) {
    const long long task = task0 + tt;
    if (task >= p.ntasks) break;
    const TokMap tm = task_map(task, nW, nWx, p.H, p.W, p.shift);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < HC::NIT; ++it) { put(Qs, rq[it], it, p.scale); put(Ks, rk[it], it, 1.f); put(Vs, rv[it], it, 1.f); put(Ds, rd[it], it, 1.f); }
    __syncthreads();
    if (tt + 1 < p.tasks_per_wave && task + 1 < p.ntasks) fetch(task + 1);
    // ---- this wave's 16 queries: S^T = K (scale Q)^T, dP^T = V dO^T, P = softmax, dS = P o (dP - rowsum(dP o P))
    f32x4 s[4], dp[4];
    {
      const int off = qrow * LDQ_H + lg * 8;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Qs + off), c = *reinterpret_cast<const bf16x8*>(Ds + off);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int offb = (nt * 16 + lr) * LDQ_H + lg * 8;
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(Ks + offb), d = *reinterpret_cast<const bf16x8*>(Vs + offb);
        s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d, c, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
    }
#ifndef SV_AT_PROBE_NOSOFTMAX
    {
      float mx = -3.0e38f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[nt][j] += bias[nt][j]; mx = fmaxf(mx, s[nt][j]); }
      if (p.shift > 0 && (tm.wy == p.H / 7 - 1 || tm.wx == nWx - 1)) {    // a window touching the rolled seam
        const int qreg = qrow_ok ? tm.region(qrow) : 0;
        mx = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int key = nt * 16 + lg * 4 + j;
            if (qrow_ok && key < WT && tm.region(key) != qreg) s[nt][j] += -100.0f;
            mx = fmaxf(mx, s[nt][j]);
          }
      }
      mx = lanegroup_allreduce(mx, fmax2);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float e = __expf(s[nt][j] - mx); s[nt][j] = e; sum += e; }
      const float inv = __builtin_amdgcn_rcpf(lanegroup_allreduce(sum, fadd2));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[nt][j] *= inv;
    }
#endif
    {
      float r = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) r += dp[nt][j] * s[nt][j];
      r = lanegroup_allreduce(r, fadd2);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dsv = s[nt][j] * (dp[nt][j] - r);
          dp[nt][j] = dsv;
          if (qrow_ok && nt * 16 + lg * 4 + j < WT) dsum[nt][j] += dsv;
        }
    }
    bf16x8 dsf[2];                                   // dS of this query as the second operand of dQ = dS K: keys 32 ks + 4 lg .. + 3, 32 ks + 16 + 4 lg .. + 3
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      bf16x4 pb, sb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { pb[j] = (__bf16)s[nt][j]; sb[j] = (__bf16)dp[nt][j]; dsf[nt >> 1][(nt & 1) * 4 + j] = sb[j]; }
#ifndef SV_AT_PROBE_NOPS
      *reinterpret_cast<bf16x4*>(Ps + qrow * LDP_H + nt * 16 + lg * 4) = pb;     // [query][key] rows: four consecutive keys per lane
      *reinterpret_cast<bf16x4*>(Ss + qrow * LDP_H + nt * 16 + lg * 4) = sb;
#endif
    }
    // ---- dQ = scale dS K for this wave's queries, from the registers: lane -> query lr, head channels 8 lg + 4 nt + j
    f32x4 aq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const __bf16* src = Ks + (2 * ks * 16 + lg * 4 + (lr >> 2)) * LDQ_H + (lr & 3) * 8 + nt * 4;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src + 16 * LDQ_H));
        const bf16x8 kT = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        aq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsf[ks], aq[nt], 0, 0, 0);
      }
    if (qrow_ok) {
      const float qv[8] = {aq[0][0] * p.scale, aq[0][1] * p.scale, aq[0][2] * p.scale, aq[0][3] * p.scale,
                           aq[1][0] * p.scale, aq[1][1] * p.scale, aq[1][2] * p.scale, aq[1][3] * p.scale};
      stnf<8>(p.dqkv + (size_t)tm.row(qrow) * ld + colq + lg * 8, qv);
    }
    __syncthreads();
    // ---- this wave's 16 KEYS: dV = P^T dO and dK = dS^T (scale Q), contractions over all 64 queries of the tiles
    f32x4 av[2], ak[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ak[nt] = av[nt]; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 pT = tr_frag(Ps, LDP_H, ks * 32, wave * 16, lane);        // A[key][q]
      const bf16x8 sT = tr_frag(Ss, LDP_H, ks * 32, wave * 16, lane);        // A[key][q]
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const bf16x8 bd = tr_frag_perm(Ds, LDQ_H, ks * 32, nt, lane);        // dO[q][d]
        const bf16x8 bq = tr_frag_perm(Qs, LDQ_H, ks * 32, nt, lane);        // (scale Q)[q][d]
        // operands swapped: the accumulators are the transposed blocks, X[row = lr][d = 8 lg + 4 nt + j] (16-byte stores below)
        av[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bd, pT, av[nt], 0, 0, 0);
        ak[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, sT, ak[nt], 0, 0, 0);
      }
    }
    {
      const int t = wave * 16 + lr;               // a key row
      if (t < WT) {
        AT* dst = p.dqkv + (size_t)tm.row(t) * ld + colq + lg * 8;
        const float kv[8] = {ak[0][0], ak[0][1], ak[0][2], ak[0][3], ak[1][0], ak[1][1], ak[1][2], ak[1][3]};
        const float vv[8] = {av[0][0], av[0][1], av[0][2], av[0][3], av[1][0], av[1][1], av[1][2], av[1][3]};
        stnf<8>(dst + p.C, kv);
        stnf<8>(dst + 2 * p.C, vv);
      }
    }
  }
  // ---- relative-position-bias gradient: registers -> LDS (once per workgroup) -> one atomic per table entry
  __syncthreads();
  if (qrow_ok) {
    const int qy = qrow / 7, qx = qrow - qy * 7;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = nt * 16 + lg * 4 + j;
        if (key < WT) {
          const int ky = key / 7, kx = key - ky * 7;
          atomicAdd(dbt + (qy - ky + 6) * 13 + (qx - kx + 6), dsum[nt][j]);
        }
      }
  }
  __syncthreads();
  float* dst = dt_ws ? dt_ws + (size_t)(chunk % ATTN_DT_SLOTS) * 169 * p.heads : p.dtable;
  for (int i = tid; i < 169; i += 256) {
    const float v = dbt[i];
    if (v != 0.f) atomicAdd(dst + i * p.heads + head, v);
  }
}
__global__ __launch_bounds__(256) void attn_dtable_fold_kernel(const float* __restrict__ ws, float* __restrict__ dtable, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
#pragma unroll
  for (int sl = 0; sl < ATTN_DT_SLOTS; ++sl) a += ws[(size_t)sl * n + i];
  dtable[i] += a;
}


// ---- fused attention branch of a stage-0 Swin block (C = 96, 3 heads) ----------------------------------------------------------------
// x1 = x + s * proj(window_attention(qkv(LayerNorm(x)))) in ONE kernel (timm SwinTransformerBlock behind models/swin_transformer.py:78:
// norm1 -> roll / partition -> qkv -> per-head softmax(q k^T * scale + bias (+ mask)) v -> proj -> reverse / roll -> drop-path -> + x).
// At C = 96 both weight matrices fit LDS as bf16 (288 x 96 + 96 x 96 = 72 KB), so a workgroup stages them once and then walks windows:
// 8 waves = 2 windows in flight, the 4 waves of a window own 16 token rows each (49 tokens padded to 64).
//  * LayerNorm runs on the MFMA operand layout itself: a lane loads the 16-byte chunks (token lr, channels 32 ks + 8 lg ..) that ARE its
//    fragment of the qkv GEMM, the row sums cross the four lane groups with two shuffles - the normalised rows never touch LDS.
//  * every product is taken with the operands swapped (weights / keys / V^T as the first MFMA operand), so the accumulator of a lane is
//    4 consecutive channels (or keys) of ONE token: q | k | v go to the LDS tile and to HBM as 8-byte vectors, and the softmax of a query
//    is 16 in-lane values + two shuffles across the lane groups instead of 16-lane reductions per accumulator row.
//  * the probabilities never leave registers: S^T = K Q^T leaves keys (16 nt + 4 lg + j) in the lane that, as the second operand of
//    P V, must supply 8 keys per K step - the contraction index of an MFMA may be permuted freely as long as both operands agree, so the
//    V^T fragment is fetched with ds_read_b64_tr_b16 from exactly those key rows (two 4-row blocks of the row-major V tile).
//  * the head outputs overwrite the (wave-private) q columns of the tile and become the operand of the projection; the residual, the
//    projection bias and the per-image drop-path factor are applied on the accumulators.
// Training needs the intermediate tensors of the unfused chain for its backward (LayerNorm output + statistics, qkv, attention output):
// they are written as side outputs when their pointers are given, bit-identical to what the unfused kernels store (q is rounded to bf16
// before the softmax scale is applied, as the core kernel does with the q it reads back), 7 instead of 13 passes over the token map;
// without them (inference) the kernel reads x and writes x1.  Two barriers per window; the next window's rows are prefetched into
// registers before the attention phase.
constexpr int FB_C = 96, FB_HEADS = 3;
constexpr int FB_LDT = 296;   // bf16 row stride of the q | k | v tile: 592 bytes = 37 x 16, odd -> 16-byte fragment reads of 16 rows spread over all bank groups
constexpr int FB_LDW = 104;   // bf16 row stride of the weight images: 208 bytes = 13 x 16
struct BlockFwdArgs {
  const __bf16* x; const float *ln_g, *ln_b, *wqkv, *bqkv, *table, *wproj, *bproj, *row_scale;
  __bf16 *x1, *ln1, *qkv, *att; float *mean, *rstd;
  int I, H, W, shift; float eps, scale; int ntasks, tasks_per_group;
};
constexpr int FB_SMEM = (288 + 96) * FB_LDW * 2 + 2 * 64 * FB_LDT * 2 + (3 * 176 + 288 + 3 * 96) * 4;

__global__ __launch_bounds__(512, 1) void swin_attn_block_fwd_kernel(const BlockFwdArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char fb_smem[FB_SMEM];
  __bf16* Wq = reinterpret_cast<__bf16*>(fb_smem);          // [288][FB_LDW]  qkv.weight rows (out, in), rows permuted inside 32-row chunks (below)
  __bf16* Wp = Wq + 288 * FB_LDW;                            // [96][FB_LDW]   proj.weight rows, same permutation
  __bf16* Tb = Wp + 96 * FB_LDW;                             // 2 x [64][FB_LDT]  q (-> O) | k | v of the two windows in flight
  float* bt = reinterpret_cast<float*>(Tb + 2 * 64 * FB_LDT);   // [3][176] relative-position bias table per head, times log2 e
  float* bq = bt + 3 * 176;                                  // [288] qkv bias
  float* bp = bq + 288;                                      // [96] proj bias
  float* lng = bp + 96;                                      // [96] norm1 weight
  float* lnb = lng + 96;                                     // [96] norm1 bias
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;
  const int lr = lane & 15, lg = lane >> 4;
  // A transposed 16 x 16 accumulator block leaves MFMA row i = 4 lg + j in register j of lane group lg.  Feeding the weight rows of a
  // 32-column chunk in the order  block `half`, MFMA row i  <-  output column 8 (i >> 2) + 4 half + (i & 3)  makes the two blocks of a
  // chunk hold the 8 CONSECUTIVE columns 8 lg .. 8 lg + 7 of a token in one lane: every global access of the kernel is a 16-byte vector
  // in the layout of the LayerNorm loads (4 lanes = 64 contiguous bytes of a token row).  The permutation is applied once, here.
  for (int i = tid; i < (288 + 96) * 24; i += 512) {
    const int R = i / 24, c4 = (i - R * 24) * 4;            // LDS row R = 32 chunk + 16 half + i
    const int ii = R & 15, n = (R & ~31) + 8 * (ii >> 2) + 4 * ((R >> 4) & 1) + (ii & 3);
    const float4 w = ld4f(R < 288 ? p.wqkv + n * FB_C + c4 : p.wproj + (n - 288) * FB_C + c4);
    bf16x4 b; b[0] = (__bf16)w.x; b[1] = (__bf16)w.y; b[2] = (__bf16)w.z; b[3] = (__bf16)w.w;
    *reinterpret_cast<bf16x4*>(Wq + R * FB_LDW + c4) = b;    // Wp follows Wq
  }
  for (int i = tid; i < 3 * 169; i += 512) { const int h = i / 169, e = i - h * 169; bt[h * 176 + e] = p.table[e * FB_HEADS + h] * ATTN_LOG2E; }
  for (int i = tid; i < 288; i += 512) bq[i] = p.bqkv[i];
  if (tid < 96) { bp[tid] = p.bproj[tid]; lng[tid] = p.ln_g[tid]; lnb[tid] = p.ln_b[tid]; }
  __syncthreads();

  const int q = wave * 16 + lr;                      // the token (query) of this lane in every transposed accumulator
  const bool qok = q < WT;
  // relative-position bias of this lane's 16 (query, key) slots per head: the same for every window
  float bias[FB_HEADS][4][4];
  {
    const int qy = q / 7, qx = q - qy * 7;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = nt * 16 + lg * 4 + j, ky = key / 7, kx = key - ky * 7;
#pragma unroll
        for (int h = 0; h < FB_HEADS; ++h)
          bias[h][nt][j] = key >= WT ? -1.0e30f : (qok ? bt[h * 176 + (qy - ky + 6) * 13 + (qx - kx + 6)] : 0.f);
      }
  }
  __bf16* T = Tb + grp * 64 * FB_LDT;
  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  const long long task0 = ((long long)blockIdx.x * 2 + grp) * p.tasks_per_group;
  const float qscale = p.scale * ATTN_LOG2E;
  const bf16x8 zero8 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
  const auto fadd2 = [](float a, float b) { return a + b; };

  // the windows of a group are consecutive: (image, window row, window column) advance by increments, one division at the start
  TokMap tnext = task_map(task0 < p.ntasks ? task0 : 0, nW, nWx, p.H, p.W, p.shift);
  const int qty = (qok ? q : 0) / 7, qtx = (qok ? q : 0) - qty * 7;       // this lane's token inside the window
  auto token_row = [&](const TokMap& t) -> size_t {
    int ys = t.wy * 7 + qty + t.shift; if (ys >= t.H) ys -= t.H;
    int xs = t.wx * 7 + qtx + t.shift; if (xs >= t.W) xs -= t.W;
    return ((size_t)t.img * t.H + ys) * t.W + xs;
  };
  bf16x8 xn[3] = {zero8, zero8, zero8};              // prefetched rows of the NEXT window
  auto fetch = [&](bool in_range) {                   // this lane's three 16-byte chunks of its token row of window `tnext`
    xn[0] = zero8; xn[1] = zero8; xn[2] = zero8;
    if (in_range && qok) {
      const __bf16* src = p.x + token_row(tnext) * FB_C + lg * 8;
      xn[0] = *reinterpret_cast<const bf16x8*>(src); xn[1] = *reinterpret_cast<const bf16x8*>(src + 32); xn[2] = *reinterpret_cast<const bf16x8*>(src + 64);
    }
  };
  fetch(task0 < p.ntasks);
  for (int tt = 0; tt < p.tasks_per_group; ++tt) {
    const long long task = task0 + tt;
    const bool live = task < p.ntasks;                                   // uniform over the 4 waves of a window
    const TokMap tm = tnext;
    const bool valid = live && qok;
    const size_t row = token_row(tm);
    const bf16x8 xr[3] = {xn[0], xn[1], xn[2]};                          // kept for the residual: the projection's columns come out in this layout
    if (++tnext.wx == nWx) { tnext.wx = 0; if (++tnext.wy == p.H / 7) { tnext.wy = 0; ++tnext.img; } }
    fetch(tt + 1 < p.tasks_per_group && task + 1 < p.ntasks);           // issued before any store of this window, lands during its GEMMs
    // ---- LayerNorm of the wave's 16 rows, on the operand layout
    bf16x8 xf[3];
    {
      float xv[3][8];
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[ks][e] = (float)xr[ks][e]; s += xv[ks][e]; }
      const float mean = lanegroup_allreduce(s, fadd2) * (1.f / FB_C);
      float v2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float a = xv[ks][e] - mean; v2 += a * a; }
      const float rstd = rsqrtf(lanegroup_allreduce(v2, fadd2) * (1.f / FB_C) + p.eps);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(lng + ks * 32 + lg * 8), g1 = *reinterpret_cast<const float4*>(lng + ks * 32 + lg * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(lnb + ks * 32 + lg * 8), b1 = *reinterpret_cast<const float4*>(lnb + ks * 32 + lg * 8 + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[ks][e] = (__bf16)((xv[ks][e] - mean) * rstd * g[e] + b[e]);
      }
      if (valid && p.ln1) {
        __bf16* dst = p.ln1 + row * FB_C + lg * 8;
        *reinterpret_cast<bf16x8*>(dst) = xf[0]; *reinterpret_cast<bf16x8*>(dst + 32) = xf[1]; *reinterpret_cast<bf16x8*>(dst + 64) = xf[2];
        if (lg == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
      }
    }
    // ---- qkv = LN(x) W^T + b, one 32-column chunk (= one head of q, k or v) at a time: lane -> token lr, columns 32 ck + 8 lg .. + 7
#pragma unroll 3
    for (int ck = 0; ck < 9; ++ck) {
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Wq + (ck * 32 + hf * 16 + lr) * FB_LDW + ks * 32 + lg * 8);
          acc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xf[ks], acc[hf], 0, 0, 0);
        }
      const int n0 = ck * 32 + lg * 8;
      const float4 b0 = *reinterpret_cast<const float4*>(bq + n0), b1 = *reinterpret_cast<const float4*>(bq + n0 + 4);
      bf16x8 o;
      o[0] = (__bf16)(acc[0][0] + b0.x); o[1] = (__bf16)(acc[0][1] + b0.y); o[2] = (__bf16)(acc[0][2] + b0.z); o[3] = (__bf16)(acc[0][3] + b0.w);
      o[4] = (__bf16)(acc[1][0] + b1.x); o[5] = (__bf16)(acc[1][1] + b1.y); o[6] = (__bf16)(acc[1][2] + b1.z); o[7] = (__bf16)(acc[1][3] + b1.w);
      if (valid && p.qkv) *reinterpret_cast<bf16x8*>(p.qkv + row * (3 * FB_C) + n0) = o;
      if (ck < 3) {   // q: the softmax scale (and log2 e) goes onto the bf16 value, as the core kernel applies it to the q it reads from HBM
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)o[e] * qscale);
      }
      *reinterpret_cast<bf16x8*>(T + q * FB_LDT + n0) = o;
    }
    __syncthreads();                                                     // all key / value rows of both windows are in LDS
    // ---- attention, one head at a time; lane -> query lr, keys nt*16 + lg*4 + j
    const bool masked = p.shift > 0 && (tm.wy == p.H / 7 - 1 || tm.wx == nWx - 1);
    unsigned kdiff = 0;                                                  // bit (nt*4 + j): key in another region than the query
    if (masked) {
      const int qreg = qok ? tm.region(q) : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = nt * 16 + lg * 4 + j;
          if (qok && key < WT && tm.region(key) != qreg) kdiff |= 1u << (nt * 4 + j);
        }
    }
#pragma unroll
    for (int h = 0; h < FB_HEADS; ++h) {
      f32x4 s[4];
      {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(T + q * FB_LDT + h * HD + lg * 8);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(T + (nt * 16 + lr) * FB_LDT + FB_C + h * HD + lg * 8);
          s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = s[nt][j] + bias[h][nt][j];      // key padding: bias = -1e30 and the padded K rows are finite (LayerNorm of a zero row = beta)
          s[nt][j] = v;
          mx = fmaxf(mx, v);
        }
      if (masked) {                                        // uniform over the window: only the last window row / column of a shifted block
        mx = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (kdiff & (1u << (nt * 4 + j))) s[nt][j] += -100.0f * ATTN_LOG2E;
            mx = fmaxf(mx, s[nt][j]);
          }
      }
      mx = lanegroup_allreduce(mx, fmax2);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float e = __builtin_amdgcn_exp2f(s[nt][j] - mx); s[nt][j] = e; sum += e; }
      const float inv = __builtin_amdgcn_rcpf(lanegroup_allreduce(sum, fadd2));
      bf16x8 pf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) { pf[ks][j] = (__bf16)(s[2 * ks][j] * inv); pf[ks][4 + j] = (__bf16)(s[2 * ks + 1][j] * inv); }
      f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          // V^T fragment of block nt: MFMA row lr <- head channel 8 (lr >> 2) + 4 nt + (lr & 3) (the chunk permutation above: the lanes with
          // lr & 3 = c supply the 4-column group that the lanes with lr >> 2 = c receive), the 8 keys of this lane group in the order the P
          // fragment holds them: keys 32 ks + 4 lg .. + 3 and 32 ks + 16 + 4 lg .. + 3
          const __bf16* src = T + (2 * ks * 16 + lg * 4 + (lr >> 2)) * FB_LDT + 2 * FB_C + h * HD + (lr & 3) * 8 + nt * 4;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src + 16 * FB_LDT));
          const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks], o[nt], 0, 0, 0);
        }
      bf16x8 ob;                                                         // lane -> query lr, head channels 8 lg .. 8 lg + 7
#pragma unroll
      for (int j = 0; j < 4; ++j) { ob[j] = (__bf16)o[0][j]; ob[4 + j] = (__bf16)o[1][j]; }
      if (valid && p.att) *reinterpret_cast<bf16x8*>(p.att + row * FB_C + h * HD + lg * 8) = ob;
      *reinterpret_cast<bf16x8*>(T + q * FB_LDT + h * HD + lg * 8) = ob;   // over this wave's own q rows
    }
    __syncthreads();                                                     // keys / values are consumed: the next window may overwrite them
    // ---- x1 = x + s * (O Wp^T + bp), lane -> token lr, channels 32 ck + 8 lg .. + 7: the layout x was loaded in
    {
      const float sc = p.row_scale ? p.row_scale[tm.img] : 1.f;
      bf16x8 of[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) of[ks] = *reinterpret_cast<const bf16x8*>(T + q * FB_LDT + ks * 32 + lg * 8);
#pragma unroll
      for (int ck = 0; ck < 3; ++ck) {
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Wp + (ck * 32 + hf * 16 + lr) * FB_LDW + ks * 32 + lg * 8);
            acc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, of[ks], acc[hf], 0, 0, 0);
          }
        if (valid) {
          const int n0 = ck * 32 + lg * 8;
          const float4 b0 = *reinterpret_cast<const float4*>(bp + n0), b1 = *reinterpret_cast<const float4*>(bp + n0 + 4);
          const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          bf16x8 y;
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = (__bf16)((float)xr[ck][e] + sc * (acc[e >> 2][e & 3] + bb[e]));
          *reinterpret_cast<bf16x8*>(p.x1 + row * FB_C + n0) = y;
        }
      }
    }
  }
}


// ---- fused BACKWARD of the attention branch of a stage-0 Swin block (C = 96, 3 heads) ------------------------------------------------
// x1 = x + s * proj(window_attention(qkv(LayerNorm(x)))).  The unfused chain of this backward is four kernels over the token map - projection
// data gradient, attention core backward, qkv data gradient, LayerNorm backward + residual - with datt, dqkv and dln1 going through HBM in
// between: 17 passes over an [M, 96] tensor.  Here the data path is ONE kernel, 9 passes: it reads dx1, qkv (saved by the forward), x (+ the
// saved LayerNorm statistics) and writes dqkv (the weight-gradient kernels of the engine need it in HBM) and dx.
// One workgroup = 4 waves = one window at a time (49 tokens padded to 64; wave w owns token rows 16 w .. 16 w + 15 wherever rows are
// owned); both weight matrices sit in LDS in DATA-GRADIENT orientation:
//   P1  datt = (s dx1) Wproj          A rows = input channel j of proj, k = output channel   [96][96]
//   P2  per head: the body of win_attn_bwd_wg_kernel (P recomputed) on the q | k | v columns of the window tile; dq / dk / dv leave for HBM
//       as 16-byte row vectors and overwrite the head's q | k | v columns of the tile once every wave is done reading them
//   P3  dln = dqkv Wqkv               A rows = input channel c of qkv, k = output column     [96][288]
//   P4  LayerNorm backward + residual on the accumulators: dx = dx1 + rstd (g - mean(g) - xhat mean(g xhat)), g = dln gamma
// Every product takes the weights as the FIRST MFMA operand with the row permutation of the forward kernel, so a lane's accumulators are
// the 8 consecutive channels 32 ck + 8 lg .. + 7 of ONE token: every global access is a 16-byte vector in one layout (dx1, x, qkv, dqkv, dx),
// and the LayerNorm row sums are 24 in-lane values + two lane-group exchanges.  dgamma / dbeta and the bias-table gradient stay in
// registers over all windows of the workgroup.  The next window's dx1 and qkv rows are prefetched into registers during the current one.
struct BlockBwdArgs {
  const __bf16 *dx1, *qkv, *x; const float *mean, *rstd, *ln_g, *wqkv, *wproj, *table, *row_scale;
  __bf16 *dqkv, *dx, *dbr; float *dgamma, *dbeta, *dt_ws;
  int I, H, W, shift; float scale; int ntasks, tasks_per_group;
};
constexpr int BB_LDT = 296;   // bf16 row stride of the q | k | v (then dq | dk | dv) tile and of the qkv weight image: 592 bytes = 37 x 16
constexpr int BB_LDD = 104;   // bf16 row stride of the s dx1 (then datt) tile and of the proj weight image: 208 bytes = 13 x 16
constexpr int BB_SMEM = 96 * BB_LDT * 2 + 96 * BB_LDD * 2 + 64 * BB_LDT * 2 + 64 * BB_LDD * 2 + 2 * 64 * LDP_H * 2 + (3 * 176 * 2 + 96 + 2 * 96) * 4;

__global__ __launch_bounds__(256, 1) void swin_attn_block_bwd_kernel(const BlockBwdArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char bb_smem[BB_SMEM];
  __bf16* Wq = reinterpret_cast<__bf16*>(bb_smem);          // [96][BB_LDT]  row R (permuted input channel c): Wqkv[k][c], k = 0 .. 287
  __bf16* Wp = Wq + 96 * BB_LDT;                             // [96][BB_LDD]  row R (permuted input channel j): Wproj[i][j], i = 0 .. 95
  __bf16* T = Wp + 96 * BB_LDD;                              // [64][BB_LDT]  scale q | k | v, then dq | dk | dv
  __bf16* D = T + 64 * BB_LDT;                               // [64][BB_LDD]  s dx1, then datt
  __bf16* Ps = D + 64 * BB_LDD;                              // [64][LDP_H]   P  [query][key]
  __bf16* Ss = Ps + 64 * LDP_H;                              // [64][LDP_H]   dS [query][key]
  float* bt = reinterpret_cast<float*>(Ss + 64 * LDP_H);     // [3][176] bias table per head
  float* dbt = bt + 3 * 176;                                 // [3][176] its gradient (filled once, at the end)
  float* lng = dbt + 3 * 176;                                // [96] norm1 weight
  float* red = lng + 96;                                     // [2][96] dgamma / dbeta of the workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // weight images, transposed from the native [out][in] layouts; LDS row R = 32 chunk + 16 half + ii <- channel 32 chunk + 8 (ii >> 2) + 4 half + (ii & 3)
  for (int i = tid; i < 96 * 288; i += 256) {
    const int R = i / 288, k = i - R * 288, ii = R & 15, c = (R & ~31) + 8 * (ii >> 2) + 4 * ((R >> 4) & 1) + (ii & 3);
    Wq[R * BB_LDT + k] = (__bf16)p.wqkv[k * FB_C + c];
  }
  for (int i = tid; i < 96 * 96; i += 256) {
    const int R = i / 96, k = i - R * 96, ii = R & 15, c = (R & ~31) + 8 * (ii >> 2) + 4 * ((R >> 4) & 1) + (ii & 3);
    Wp[R * BB_LDD + k] = (__bf16)p.wproj[k * FB_C + c];
  }
  for (int i = tid; i < 3 * 176; i += 256) { const int h = i / 176, e = i - h * 176; bt[i] = e < 169 ? p.table[e * FB_HEADS + h] : 0.f; dbt[i] = 0.f; }
  if (tid < 96) { lng[tid] = p.ln_g[tid]; red[tid] = 0.f; red[96 + tid] = 0.f; }
  __syncthreads();

  const int q = wave * 16 + lr;                      // the token of this lane wherever rows are owned (queries of P2, keys of its second half)
  const bool qok = q < WT;
  float bias[FB_HEADS][4][4];
  {
    const int qy = q / 7, qx = q - qy * 7;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = nt * 16 + lg * 4 + j, ky = key / 7, kx = key - ky * 7;
#pragma unroll
        for (int h = 0; h < FB_HEADS; ++h)
          bias[h][nt][j] = key >= WT ? -1.0e30f : (qok ? bt[h * 176 + (qy - ky + 6) * 13 + (qx - kx + 6)] * ATTN_LOG2E : 0.f);   // exp2 domain
      }
  }
  const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
  const auto fadd2 = [](float a, float b) { return a + b; };
  const bf16x8 zero8 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  f32x4 dsum[FB_HEADS][4];                           // bias-table gradient at this lane's (query, key) slots, over all windows
#pragma unroll
  for (int h = 0; h < FB_HEADS; ++h)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) dsum[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dgam[3][8], dbet[3][8];                      // LayerNorm parameter gradients of this lane's channels 32 ck + 8 lg + e, over all windows
#pragma unroll
  for (int ck = 0; ck < 3; ++ck)
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgam[ck][e] = 0.f; dbet[ck][e] = 0.f; }

  const int nWx = p.W / 7, nW = (p.H / 7) * nWx;
  const long long task0 = (long long)blockIdx.x * p.tasks_per_group;
  TokMap tnext = task_map(task0 < p.ntasks ? task0 : 0, nW, nWx, p.H, p.W, p.shift);
  const int qty = (qok ? q : 0) / 7, qtx = (qok ? q : 0) - qty * 7;
  auto token_row = [&](const TokMap& t) -> size_t {
    int ys = t.wy * 7 + qty + t.shift; if (ys >= t.H) ys -= t.H;
    int xs = t.wx * 7 + qtx + t.shift; if (xs >= t.W) xs -= t.W;
    return ((size_t)t.img * t.H + ys) * t.W + xs;
  };
  bf16x8 nd[3], nq[9];                               // the NEXT window's dx1 and qkv chunks of this lane's token
  auto fetch = [&](bool in_range) {
#pragma unroll
    for (int i = 0; i < 3; ++i) nd[i] = zero8;
#pragma unroll
    for (int i = 0; i < 9; ++i) nq[i] = zero8;
    if (in_range && qok) {
      const size_t row = token_row(tnext);
      const __bf16* sd = p.dx1 + row * FB_C + lg * 8;
      const __bf16* sq = p.qkv + row * (3 * FB_C) + lg * 8;
#pragma unroll
      for (int i = 0; i < 3; ++i) nd[i] = *reinterpret_cast<const bf16x8*>(sd + 32 * i);
#pragma unroll
      for (int i = 0; i < 9; ++i) nq[i] = *reinterpret_cast<const bf16x8*>(sq + 32 * i);
    }
  };
  fetch(task0 < p.ntasks);
  for (int tt = 0; tt < p.tasks_per_group; ++tt) {
    const long long task = task0 + tt;
    if (task >= p.ntasks) break;                     // uniform over the workgroup
    const TokMap tm = tnext;
    const bool valid = qok;
    const size_t row = token_row(tm);
    bf16x8 dxr[3], xr[3];                            // dx1 (unscaled, for the residual) and x of this lane's token
    const float sc = p.row_scale ? p.row_scale[tm.img] : 1.f;
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { dxr[i] = nd[i]; xr[i] = zero8; }
    if (valid) {
      const __bf16* sx = p.x + row * FB_C + lg * 8;
#pragma unroll
      for (int i = 0; i < 3; ++i) xr[i] = *reinterpret_cast<const bf16x8*>(sx + 32 * i);       // consumed in P4: lands during P1 - P3
      mean = p.mean[row]; rstd = p.rstd[row];
    }
    __syncthreads();                                 // the previous window's tiles are consumed
    // ---- tiles: s dx1 -> D, scale q | k | v -> T (q is rounded to bf16 AFTER the scale goes on, as the unfused core does with the q it reads)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)(sc * (float)dxr[i][e]);
      *reinterpret_cast<bf16x8*>(D + q * BB_LDD + 32 * i + lg * 8) = o;
      if (p.dbr && valid) *reinterpret_cast<bf16x8*>(p.dbr + row * FB_C + 32 * i + lg * 8) = o;   // for the projection's weight gradient
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      bf16x8 o = nq[i];
      if (i < 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)o[e] * p.scale);
      }
      *reinterpret_cast<bf16x8*>(T + q * BB_LDT + 32 * i + lg * 8) = o;
    }
    if (++tnext.wx == nWx) { tnext.wx = 0; if (++tnext.wy == p.H / 7) { tnext.wy = 0; ++tnext.img; } }
    fetch(tt + 1 < p.tasks_per_group && task + 1 < p.ntasks);
    // ---- P1: datt = (s dx1) Wproj for this wave's 16 tokens (its own rows of D: no barrier), written back over them
    {
      bf16x8 df[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) df[ks] = *reinterpret_cast<const bf16x8*>(D + q * BB_LDD + ks * 32 + lg * 8);
#pragma unroll
      for (int ck = 0; ck < 3; ++ck) {
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Wp + (ck * 32 + hf * 16 + lr) * BB_LDD + ks * 32 + lg * 8);
            acc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, df[ks], acc[hf], 0, 0, 0);
          }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e >> 2][e & 3];
        *reinterpret_cast<bf16x8*>(D + q * BB_LDD + ck * 32 + lg * 8) = o;
      }
    }
    __syncthreads();                                 // every row of T and of D (= datt) is in LDS
    const bool masked = p.shift > 0 && (tm.wy == p.H / 7 - 1 || tm.wx == nWx - 1);
    unsigned kdiff = 0;                              // bit (nt * 4 + j): key in another region of the rolled map than the query
    if (masked) {
      const int qreg = qok ? tm.region(q) : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = nt * 16 + lg * 4 + j;
          if (qok && key < WT && tm.region(key) != qreg) kdiff |= 1u << (nt * 4 + j);
        }
    }
    // ---- P2: attention backward, one head at a time (win_attn_bwd_wg_kernel's body on the tile columns of the head)
#pragma unroll
    for (int h = 0; h < FB_HEADS; ++h) {
      const __bf16* Qs = T + h * HD;
      const __bf16* Ks = T + FB_C + h * HD;
      const __bf16* Vs = T + 2 * FB_C + h * HD;
      const __bf16* Ds = D + h * HD;
      f32x4 s[4], dp[4];
      {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Qs + q * BB_LDT + lg * 8), c = *reinterpret_cast<const bf16x8*>(Ds + q * BB_LDD + lg * 8);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(Ks + (nt * 16 + lr) * BB_LDT + lg * 8);
          const bf16x8 d = *reinterpret_cast<const bf16x8*>(Vs + (nt * 16 + lr) * BB_LDT + lg * 8);
          s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          dp[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d, c, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
      }
      {
        // scores in the exp2 domain: log2 e rides on the bias (above) and on the raw product here (Q carries the plain softmax scale, as the
        // dK product needs it)
        float mx = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = __builtin_fmaf(s[nt][j], ATTN_LOG2E, bias[h][nt][j]);
            s[nt][j] = v;
            mx = fmaxf(mx, v);
          }
        if (masked) {                                // uniform over the workgroup: only windows on the rolled seam
          mx = -3.0e38f;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (kdiff & (1u << (nt * 4 + j))) s[nt][j] += -100.0f * ATTN_LOG2E;
              mx = fmaxf(mx, s[nt][j]);
            }
        }
        mx = lanegroup_allreduce(mx, fmax2);
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float e = __builtin_amdgcn_exp2f(s[nt][j] - mx); s[nt][j] = e; sum += e; }
        const float inv = __builtin_amdgcn_rcpf(lanegroup_allreduce(sum, fadd2));
        float r = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) { s[nt][j] *= inv; r += dp[nt][j] * s[nt][j]; }
        r = lanegroup_allreduce(r, fadd2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float dsv = s[nt][j] * (dp[nt][j] - r);
            dp[nt][j] = dsv;
            if (qok && nt * 16 + lg * 4 + j < WT) dsum[h][nt][j] += dsv;
          }
      }
      bf16x8 dsf[2];                                 // dS of this query as the second operand of dQ = dS K
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bf16x4 pb, sb;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pb[j] = (__bf16)s[nt][j]; sb[j] = (__bf16)dp[nt][j]; dsf[nt >> 1][(nt & 1) * 4 + j] = sb[j]; }
        *reinterpret_cast<bf16x4*>(Ps + q * LDP_H + nt * 16 + lg * 4) = pb;
        *reinterpret_cast<bf16x4*>(Ss + q * LDP_H + nt * 16 + lg * 4) = sb;
      }
      f32x4 aq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};   // dQ = scale dS K: lane -> query lr, head channels 8 lg + 4 nt + j
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const __bf16* src = Ks + (2 * ks * 16 + lg * 4 + (lr >> 2)) * BB_LDT + (lr & 3) * 8 + nt * 4;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_a*)(src + 16 * BB_LDT));
          const bf16x8 kT = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          aq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsf[ks], aq[nt], 0, 0, 0);
        }
      __syncthreads();                               // P and dS of all 64 queries are in LDS
      f32x4 av[2], ak[2];                            // this wave's 16 KEYS: dV = P^T dO, dK = dS^T (scale Q)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) { av[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ak[nt] = av[nt]; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 pT = tr_frag(Ps, LDP_H, ks * 32, wave * 16, lane);
        const bf16x8 sT = tr_frag(Ss, LDP_H, ks * 32, wave * 16, lane);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16x8 bd = tr_frag_perm(Ds, BB_LDD, ks * 32, nt, lane);
          const bf16x8 bq = tr_frag_perm(Qs, BB_LDT, ks * 32, nt, lane);
          av[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bd, pT, av[nt], 0, 0, 0);
          ak[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, sT, ak[nt], 0, 0, 0);
        }
      }
      bf16x8 oq, ok_, ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) { oq[e] = (__bf16)(aq[e >> 2][e & 3] * p.scale); ok_[e] = (__bf16)ak[e >> 2][e & 3]; ov[e] = (__bf16)av[e >> 2][e & 3]; }
      if (valid) {
        __bf16* dst = p.dqkv + row * (3 * FB_C) + h * HD + lg * 8;
        *reinterpret_cast<bf16x8*>(dst) = oq;
        *reinterpret_cast<bf16x8*>(dst + FB_C) = ok_;
        *reinterpret_cast<bf16x8*>(dst + 2 * FB_C) = ov;
      }
      __syncthreads();                               // every wave is done with this head's q | k | v, P and dS
      {
        __bf16* dstl = T + q * BB_LDT + h * HD + lg * 8;     // this wave's own rows: dq | dk | dv become the operand of P3 (zero rows for the padding)
        *reinterpret_cast<bf16x8*>(dstl) = valid ? oq : zero8;
        *reinterpret_cast<bf16x8*>(dstl + FB_C) = valid ? ok_ : zero8;
        *reinterpret_cast<bf16x8*>(dstl + 2 * FB_C) = valid ? ov : zero8;
      }
    }
    // ---- P3: dln = dqkv Wqkv for this wave's 16 tokens (its own rows of T), P4: LayerNorm backward + residual on the accumulators
    {
      bf16x8 gf[9];
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) gf[ks] = *reinterpret_cast<const bf16x8*>(T + q * BB_LDT + ks * 32 + lg * 8);
      float dl[3][8], xh[3][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ck = 0; ck < 3; ++ck) {
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 9; ++ks)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Wq + (ck * 32 + hf * 16 + lr) * BB_LDT + ks * 32 + lg * 8);
            acc[hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, gf[ks], acc[hf], 0, 0, 0);
          }
        const float4 g0 = *reinterpret_cast<const float4*>(lng + ck * 32 + lg * 8), g1 = *reinterpret_cast<const float4*>(lng + ck * 32 + lg * 8 + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // the unfused chain stores dln1 as bf16 between the qkv data gradient and the LayerNorm backward: the same rounding here
          const float d = valid ? (float)(__bf16)acc[e >> 2][e & 3] : 0.f;
          const float xv = valid ? ((float)xr[ck][e] - mean) * rstd : 0.f;
          dbet[ck][e] += d; dgam[ck][e] += d * xv;
          const float g = d * gm[e];
          dl[ck][e] = g; xh[ck][e] = xv;
          s1 += g; s2 += g * xv;
        }
      }
      s1 = lanegroup_allreduce(s1, fadd2) * (1.f / FB_C);
      s2 = lanegroup_allreduce(s2, fadd2) * (1.f / FB_C);
      if (valid) {
#pragma unroll
        for (int ck = 0; ck < 3; ++ck) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)dxr[ck][e] + rstd * (dl[ck][e] - s1 - xh[ck][e] * s2));
          *reinterpret_cast<bf16x8*>(p.dx + row * FB_C + ck * 32 + lg * 8) = o;
        }
      }
    }
  }
  // ---- parameter gradients: registers -> LDS (once per workgroup) -> atomics
  __syncthreads();
#pragma unroll
  for (int ck = 0; ck < 3; ++ck)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = group16_sum(dgam[ck][e]), b = group16_sum(dbet[ck][e]);      // the 16 tokens of a lane group
      if (lr == 0) { atomicAdd(red + ck * 32 + lg * 8 + e, a); atomicAdd(red + 96 + ck * 32 + lg * 8 + e, b); }
    }
  if (qok) {
    const int qy = q / 7, qx = q - qy * 7;
#pragma unroll
    for (int h = 0; h < FB_HEADS; ++h)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = nt * 16 + lg * 4 + j;
          if (key < WT) {
            const int ky = key / 7, kx = key - ky * 7;
            atomicAdd(dbt + h * 176 + (qy - ky + 6) * 13 + (qx - kx + 6), dsum[h][nt][j]);
          }
        }
  }
  __syncthreads();
  if (tid < 96) { atomicAdd(p.dgamma + tid, red[tid]); atomicAdd(p.dbeta + tid, red[96 + tid]); }
  float* dst = p.dt_ws + (size_t)(blockIdx.x % ATTN_DT_SLOTS) * 169 * FB_HEADS;
  for (int i = tid; i < 3 * 169; i += 256) {
    const int h = i / 169, e = i - h * 169;
    const float v = dbt[h * 176 + e];
    if (v != 0.f) atomicAdd(dst + e * FB_HEADS + h, v);
  }
}

}  // namespace sv

using namespace sv;

static int win_check(const void* qkv, const float* table, int I, int H, int W, int C, int heads, int shift, int math, int act_dtype) {
  SV_REQUIRE(qkv && table && I > 0, "window_attention: null/empty argument");
  SV_REQUIRE(H % 7 == 0 && W % 7 == 0 && H >= 7 && W >= 7, "window_attention: map %dx%d is not a multiple of the 7x7 window", H, W);
  SV_REQUIRE(C == heads * HD, "window_attention: C (%d) must equal heads (%d) * 32", C, heads);
  SV_REQUIRE(shift >= 0 && shift < 7 && (shift == 0 || (H > 7 && W > 7)), "window_attention: bad shift %d for map %dx%d", shift, H, W);
  SV_REQUIRE(((uintptr_t)qkv & 15) == 0, "window_attention: qkv must be 16-byte aligned");
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE(act_dtype == SV_F32 || math == SV_MATH_BF16 || math == SV_MATH_FP8, "window_attention: bf16 activations require SV_MATH_BF16 / SV_MATH_FP8");
  return SV_OK;
}

template <typename AT>
static WinArgsT<AT> win_args(const void* qkv, const float* table, void* out, const void* dout, void* dqkv, float* dtable, int I, int H, int W,
                             int C, int heads, int shift) {
  WinArgsT<AT> a{};
  a.qkv = static_cast<const AT*>(qkv); a.table = table; a.out = static_cast<AT*>(out);
  a.dout = static_cast<const AT*>(dout); a.dqkv = static_cast<AT*>(dqkv); a.dtable = dtable;
  a.I = I; a.H = H; a.W = W; a.C = C; a.heads = heads; a.shift = shift;
  a.scale = 1.0f / sqrtf((float)HD);
  a.ntasks = I * (H / 7) * (W / 7);
  return a;
}

extern "C" int sv_window_attention_fwd(const void* qkv, const float* table, void* out, int I, int H, int W, int C, int heads,
                                       int shift, int math, int act_dtype, void* stream) {
  if (int rc = win_check(qkv, table, I, H, W, C, heads, shift, math, act_dtype)) return rc;
  SV_REQUIRE(out, "window_attention_fwd: null out");
  const int ntasks = I * (H / 7) * (W / 7);
  hipStream_t s = (hipStream_t)stream;
  if (math == SV_MATH_FP8) {    // e4m3 operands for QK^T and PV (forward only; the backward entry point treats SV_MATH_FP8 as bf16)
    int tpb = (int)((long long)ntasks * heads / 2048); if (tpb < 1) tpb = 1; if (tpb > 8) tpb = 8;
    dim3 grid(wg_grid(cdiv(ntasks, tpb), heads));
    if (act_dtype == SV_BF16) {
      WinArgsT<__bf16> a = win_args<__bf16>(qkv, table, out, nullptr, nullptr, nullptr, I, H, W, C, heads, shift);
      a.tasks_per_wave = tpb;
      hipLaunchKernelGGL(win_attn_fwd_wg_fp8_kernel<__bf16>, grid, dim3(256), 0, s, a);
    } else {
      WinArgs a = win_args<float>(qkv, table, out, nullptr, nullptr, nullptr, I, H, W, C, heads, shift);
      a.tasks_per_wave = tpb;
      hipLaunchKernelGGL(win_attn_fwd_wg_fp8_kernel<float>, grid, dim3(256), 0, s, a);
    }
    return check_launch("sv_window_attention_fwd");
  }
  if (math == SV_MATH_BF16) {   // workgroup per window; a workgroup walks tpb windows of one head (>= ~2048 short workgroups:
    // the forward task is brief, oversubscribing the CUs balances better than one exact wave - measured)
    int tpb = (int)((long long)ntasks * heads / 2048); if (tpb < 1) tpb = 1; if (tpb > 8) tpb = 8;
    dim3 grid(wg_grid(cdiv(ntasks, tpb), heads));
    if (act_dtype == SV_BF16) {
      WinArgsT<__bf16> a = win_args<__bf16>(qkv, table, out, nullptr, nullptr, nullptr, I, H, W, C, heads, shift);
      a.tasks_per_wave = tpb;
      hipLaunchKernelGGL(win_attn_fwd_wg_kernel<__bf16>, grid, dim3(256), 0, s, a);
    } else {
      WinArgs a = win_args<float>(qkv, table, out, nullptr, nullptr, nullptr, I, H, W, C, heads, shift);
      a.tasks_per_wave = tpb;
      hipLaunchKernelGGL(win_attn_fwd_wg_kernel<float>, grid, dim3(256), 0, s, a);
    }
  } else {
    dim3 grid(cdiv(ntasks, 4), heads);
    hipLaunchKernelGGL((win_attn_fwd_kernel<false, float>), grid, dim3(256), 0, s, win_args<float>(qkv, table, out, nullptr, nullptr, nullptr, I, H, W, C, heads, shift));
  }
  return check_launch("sv_window_attention_fwd");
}

extern "C" size_t sv_window_attention_bwd_workspace_floats(int heads) { return (size_t)ATTN_DT_SLOTS * 169 * heads; }

extern "C" int sv_window_attention_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, float* workspace,
                                       int I, int H, int W, int C, int heads, int shift, int math, int act_dtype, void* stream) {
  if (int rc = win_check(qkv, table, I, H, W, C, heads, shift, math, act_dtype)) return rc;
  SV_REQUIRE(dout && dqkv && dtable && ((uintptr_t)dout & 15) == 0, "window_attention_bwd: null/unaligned argument");
  const int ntasks = I * (H / 7) * (W / 7);
  hipStream_t s = (hipStream_t)stream;
  if (math == SV_MATH_BF16 || math == SV_MATH_FP8) {   // workgroup per window, tpb windows of one head per workgroup
    const int tpb = wg_tasks_per_block(ntasks, heads, 3);
    dim3 grid(wg_grid(cdiv(ntasks, tpb), heads));
    if (act_dtype == SV_BF16) {
      WinArgsT<__bf16> a = win_args<__bf16>(qkv, table, nullptr, dout, dqkv, dtable, I, H, W, C, heads, shift);
      a.tasks_per_wave = tpb;
      hipLaunchKernelGGL(win_attn_bwd_wg_kernel<__bf16>, grid, dim3(256), 0, s, a, workspace);
    } else {
      WinArgs a = win_args<float>(qkv, table, nullptr, dout, dqkv, dtable, I, H, W, C, heads, shift);
      a.tasks_per_wave = tpb;
      hipLaunchKernelGGL(win_attn_bwd_wg_kernel<float>, grid, dim3(256), 0, s, a, workspace);
    }
    if (workspace) hipLaunchKernelGGL(attn_dtable_fold_kernel, dim3(cdiv(169 * heads, 256)), dim3(256), 0, s, workspace, dtable, 169 * heads);
  } else {
    // exact-fp32 MFMA: wave per window, several windows per wave (same head) so the bias gradient is reduced on chip
    int tpw = 1;
    while (tpw < 8 && (long long)ntasks * heads / (tpw * 2) > 2048) tpw *= 2;
    dim3 grid(cdiv(ntasks, 2 * tpw), heads);
    WinArgs a = win_args<float>(qkv, table, nullptr, dout, dqkv, dtable, I, H, W, C, heads, shift);
    a.tasks_per_wave = tpw;
    hipLaunchKernelGGL(win_attn_bwd_kernel, grid, dim3(128), 0, s, a);
  }
  return check_launch("sv_window_attention_bwd");
}

static int cva_check(int B, int V, int P, int R, int heads) {
  SV_REQUIRE(B > 0 && V > 0 && V <= CVA_MAXV, "cross_view_attention: n_views=%d unsupported (max %d)", V, CVA_MAXV);
  SV_REQUIRE(heads > 0 && R % heads == 0 && P > 0 && R / heads <= CVA_MAXF, "cross_view_attention: head_dim %d exceeds %d",
             R / (heads > 0 ? heads : 1), CVA_MAXF);
  return SV_OK;
}
// features per chunk: whole positions, as many as the LDS arrays hold
static int cva_chunk(int P, int R, int heads) { const int hd = R / heads; return min(P, CVA_MAXF / hd) * hd; }

extern "C" int sv_cross_view_attention_fwd(const void* qkv, void* out, int B, int V, int P, int R, int heads, int act_dtype, void* stream) {
  SV_REQUIRE(qkv && out, "cross_view_attention_fwd: null argument");
  SV_REQUIRE_ACT(act_dtype);
  if (int rc = cva_check(B, V, P, R, heads)) return rc;
  SV_DISPATCH_ACT(act_dtype,
    CvaArgsT<AT> a{static_cast<const AT*>(qkv), static_cast<AT*>(out), nullptr, nullptr, B, V, P, R, heads, R / heads, cva_chunk(P, R, heads), 1.0f / sqrtf((float)(R / heads) * (float)V)};
    hipLaunchKernelGGL(cva_attn_fwd_kernel<AT>, dim3(B * heads), dim3(256), 0, (hipStream_t)stream, a););
  return check_launch("sv_cross_view_attention_fwd");
}

extern "C" int sv_cross_view_attention_bwd(const void* qkv, const void* dout, void* dqkv, int B, int V, int P, int R, int heads,
                                           int act_dtype, void* stream) {
  SV_REQUIRE(qkv && dout && dqkv, "cross_view_attention_bwd: null argument");
  SV_REQUIRE_ACT(act_dtype);
  if (int rc = cva_check(B, V, P, R, heads)) return rc;
  SV_DISPATCH_ACT(act_dtype,
    CvaArgsT<AT> a{static_cast<const AT*>(qkv), nullptr, static_cast<const AT*>(dout), static_cast<AT*>(dqkv), B, V, P, R, heads, R / heads,
                   cva_chunk(P, R, heads), 1.0f / sqrtf((float)(R / heads) * (float)V)};
    hipLaunchKernelGGL(cva_attn_bwd_kernel<AT>, dim3(B * heads), dim3(256), 0, (hipStream_t)stream, a););
  return check_launch("sv_cross_view_attention_bwd");
}

// ---- fused attention branch of a stage-0 Swin block ------------------------------------------------------------------------------------
extern "C" int sv_swin_attn_block_supported(int C, int heads, int act_dtype, int math) {
  return C == FB_C && heads == FB_HEADS && act_dtype == SV_BF16 && math == SV_MATH_BF16;
}

extern "C" int sv_swin_attn_block_fwd(const void* x, const float* ln_g, const float* ln_b, const float* wqkv, const float* bqkv, const float* table,
                                      const float* wproj, const float* bproj, const float* row_scale, void* x1, void* ln1, float* mean,
                                      float* rstd, void* qkv, void* att, int I, int H, int W, int C, int heads, int shift, float eps,
                                      int act_dtype, void* stream) {
  SV_REQUIRE(x && ln_g && ln_b && wqkv && bqkv && table && wproj && bproj && x1 && I > 0, "swin_attn_block_fwd: null/empty argument");
  SV_REQUIRE(C == FB_C && heads == FB_HEADS && act_dtype == SV_BF16, "swin_attn_block_fwd: built for C = 96, 3 heads, bf16 token rows (got C=%d heads=%d dtype=%d)",
             C, heads, act_dtype);
  SV_REQUIRE(H % 7 == 0 && W % 7 == 0 && H >= 7 && W >= 7, "swin_attn_block_fwd: map %dx%d is not a multiple of the 7x7 window", H, W);
  SV_REQUIRE(shift >= 0 && shift < 7 && (shift == 0 || (H > 7 && W > 7)), "swin_attn_block_fwd: bad shift %d for map %dx%d", shift, H, W);
  const bool side = ln1 || mean || rstd || qkv || att;
  SV_REQUIRE(!side || (ln1 && mean && rstd && qkv && att), "swin_attn_block_fwd: the side outputs (ln1, mean, rstd, qkv, att) come all or none");
  SV_REQUIRE((((uintptr_t)x | (uintptr_t)x1 | (uintptr_t)ln1 | (uintptr_t)qkv | (uintptr_t)att | (uintptr_t)wqkv | (uintptr_t)wproj) & 15) == 0,
             "swin_attn_block_fwd: tensors must be 16-byte aligned");
  BlockFwdArgs a{};
  a.x = static_cast<const __bf16*>(x); a.ln_g = ln_g; a.ln_b = ln_b; a.wqkv = wqkv; a.bqkv = bqkv; a.table = table; a.wproj = wproj; a.bproj = bproj;
  a.row_scale = row_scale; a.x1 = static_cast<__bf16*>(x1); a.ln1 = static_cast<__bf16*>(ln1); a.qkv = static_cast<__bf16*>(qkv);
  a.att = static_cast<__bf16*>(att); a.mean = mean; a.rstd = rstd;
  a.I = I; a.H = H; a.W = W; a.shift = shift; a.eps = eps; a.scale = 1.0f / sqrtf((float)HD);
  a.ntasks = I * (H / 7) * (W / 7);
  a.tasks_per_group = cdiv(a.ntasks, 512);                  // one workgroup (two window groups) per CU, every group the same share
  const int nblocks = cdiv(a.ntasks, 2 * a.tasks_per_group);
  hipLaunchKernelGGL(swin_attn_block_fwd_kernel, dim3(nblocks), dim3(512), 0, (hipStream_t)stream, a);
  return check_launch("sv_swin_attn_block_fwd");
}

/* Fused backward of the attention branch of a stage-0 block (see swin_attn_block_bwd_kernel).  dgamma / dbeta / dtable are accumulated into;
 * workspace = sv_window_attention_bwd_workspace_floats(heads) floats, zero on entry. */
extern "C" int sv_swin_attn_block_bwd(const void* dx1, const void* qkv, const void* x, const float* mean, const float* rstd, const float* ln_g,
                                      const float* wqkv, const float* wproj, const float* table, const float* row_scale, void* dqkv, void* dx,
                                      void* dbr, float* dgamma, float* dbeta, float* dtable, float* workspace, int I, int H, int W, int C,
                                      int heads, int shift, int act_dtype, void* stream) {
  SV_REQUIRE(dx1 && qkv && x && mean && rstd && ln_g && wqkv && wproj && table && dqkv && dx && dgamma && dbeta && dtable && workspace && I > 0,
             "swin_attn_block_bwd: null/empty argument");
  SV_REQUIRE(C == FB_C && heads == FB_HEADS && act_dtype == SV_BF16, "swin_attn_block_bwd: built for C = 96, 3 heads, bf16 token rows (got C=%d heads=%d dtype=%d)",
             C, heads, act_dtype);
  SV_REQUIRE(H % 7 == 0 && W % 7 == 0 && H >= 7 && W >= 7, "swin_attn_block_bwd: map %dx%d is not a multiple of the 7x7 window", H, W);
  SV_REQUIRE(shift >= 0 && shift < 7 && (shift == 0 || (H > 7 && W > 7)), "swin_attn_block_bwd: bad shift %d for map %dx%d", shift, H, W);
  SV_REQUIRE((((uintptr_t)dx1 | (uintptr_t)qkv | (uintptr_t)x | (uintptr_t)dqkv | (uintptr_t)dx | (uintptr_t)dbr) & 15) == 0,
             "swin_attn_block_bwd: tensors must be 16-byte aligned");
  BlockBwdArgs a{};
  a.dx1 = static_cast<const __bf16*>(dx1); a.qkv = static_cast<const __bf16*>(qkv); a.x = static_cast<const __bf16*>(x);
  a.mean = mean; a.rstd = rstd; a.ln_g = ln_g; a.wqkv = wqkv; a.wproj = wproj; a.table = table; a.row_scale = row_scale;
  a.dqkv = static_cast<__bf16*>(dqkv); a.dx = static_cast<__bf16*>(dx); a.dbr = static_cast<__bf16*>(dbr);
  a.dgamma = dgamma; a.dbeta = dbeta; a.dt_ws = workspace;
  a.I = I; a.H = H; a.W = W; a.shift = shift; a.scale = 1.0f / sqrtf((float)HD);
  a.ntasks = I * (H / 7) * (W / 7);
  a.tasks_per_group = cdiv(a.ntasks, 256);                  // one workgroup per CU, every workgroup the same share of consecutive windows
  const int nblocks = cdiv(a.ntasks, a.tasks_per_group);
  hipLaunchKernelGGL(swin_attn_block_bwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(attn_dtable_fold_kernel, dim3(cdiv(169 * heads, 256)), dim3(256), 0, (hipStream_t)stream, workspace, dtable, 169 * heads);
  return check_launch("sv_swin_attn_block_bwd");
}
