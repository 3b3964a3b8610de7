// Fused Swin MLP branch for gfx950:  x2 = x1 + s * fc2(GELU(fc1(LayerNorm(x1))))   (timm Mlp + norm2 + DropPath behind
// reference models/swin_transformer.py:78), forward, data gradient and weight gradients, with NO hidden activation in HBM.
//
// Unfused, a stage-0 block moves 7.5 GB per step for this branch (the 4C-wide hidden tensor is written twice in the forward -
// pre-activation and GELU output - and read / written four more times in the backward) through kernels that are bound by the
// HBM write stream.  Here the hidden tile never leaves the CU:
//   swin_mlp_fwd_kernel   one wave owns 32 tokens.  H^T chunk [32 hidden x 32 tokens] = W1 chunk . LN(x)^T on MFMA 32x32x16, bias +
//                         GELU on the accumulator, and the accumulator is fed straight back as the B operand of the second product
//                         Y^T += W2 chunk . H^T (an accumulator tile is a valid B fragment when the other operand's k order is permuted to
//                         match - cdna_hip_programming.md "An accumulator tile as the next MFMA's operand").  Weight chunks stream from L2
//                         through a double-buffered LDS ring in FRAGMENT ORDER (sv_swin_mlp_pack), so every A fragment is one
//                         conflict-free ds_read_b128.  Reads x1, writes x2: 4 bytes per element instead of 26.
//   swin_mlp_bwd_kernel   same skeleton, recomputes LN and the pre-activation, dH^T = (W2^T chunk . dy^T) * GELU'(hpre), dLN^T += W1^T chunk . dH^T,
//                         then the LayerNorm backward and the residual add on the accumulator; dgamma / dbeta by DPP row reductions.
//   swin_mlp_wgrad_kernel a workgroup owns a slice of the hidden units (one 16/32-wide chunk per wave) and walks a range of tokens: the
//                         x / dy tiles go to LDS once per 128 tokens and serve every wave; H and dH are recomputed for the wave's chunk
//                         (MFMA 16x16x32, tokens on the accumulator rows) and contracted over the tokens with the transposed tiles read by
//                         ds_read_b64_tr_b16: dW2 += dy^T . H, dW1^T += LN(x)^T . dH.  Accumulators stay in registers for the whole range.
// bf16 storage + bf16 MFMA only (the exact-fp32 parity mode keeps the unfused chain); C = 96, 128, 192 (Swin-T stages 0-1, Swin-B stage 0).
#include "common.h"

namespace sv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

struct MlpArgs {
  const __bf16* x1; const __bf16* dx2; __bf16* out;
  const float* ln_g; const float* ln_b; float eps;
  const __bf16* packs;                 // W1F | W2F | W2TF | W1TF, 4C*C elements each (sv_swin_mlp_pack)
  const float* b1; const float* b2;
  const float* row_scale; int rows_per_scale;
  float* dgamma; float* dbeta;
  long long M; int ntiles;
};

// ---- weight packs in MFMA fragment order ------------------------------------------------------------------------------------
// chunk = 32 hidden units.  "row-k" images (A = [32 hidden rows][k = c]):       [chunk][f = c/16][lane][8]   lane = (r, h): hid = 32 chunk + r, c = 16 f + 8 h + j
//                           "k-perm" images (A = [32 c rows][k = hidden, permuted]): [chunk][blk = c/32][s][lane][8] lane = (r, h): c = 32 blk + r,
//                           hid = 32 chunk + 16 s + 8 (j >> 2) + 4 h + (j & 3)  - the k order in which a 32x32 accumulator tile is a B fragment.
__global__ __launch_bounds__(256) void swin_mlp_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, __bf16* __restrict__ packs, int C) {
  const int HID = 4 * C;
  const long long n = (long long)HID * C;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < 4 * n; idx += (long long)gridDim.x * 256) {
    const int img = (int)(idx / n);
    const int e = (int)(idx % n);
    const int chunk = e / (32 * C), rem = e % (32 * C);
    const int lane = (rem % 512) / 8, j = rem % 8, r = lane & 31, h = lane >> 5;
    float v;
    if (img == 0 || img == 2) {
      const int f = rem / 512;
      const int hid = 32 * chunk + r, c = 16 * f + 8 * h + j;
      v = img == 0 ? w1[(size_t)hid * C + c] : w2[(size_t)c * HID + hid];
    } else {
      const int blk = rem / 1024, s = (rem % 1024) / 512;
      const int c = 32 * blk + r, hid = 32 * chunk + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      v = img == 1 ? w2[(size_t)c * HID + hid] : w1[(size_t)hid * C + c];
    }
    packs[idx] = (__bf16)v;
  }
}

// GELU and its derivative for the bf16 path: Phi(x) = logistic(x (c0 + c1 x^2 + c2 x^4)) (odd quintic in the logit, minimax-fitted to the
// normal CDF over |x| <= 8; x^2 is clamped at 64, where the logistic has saturated to 1 - 7e-13).  |x Phi(x) - GELU_erf(x)| <= 2.9e-5 and
// |d/dx - GELU_erf'(x)| <= 1.1e-4 everywhere: 1/30 of the bf16 rounding unit of the values that are stored or fed to the matrix pipe.  Cost:
// 6 VALU + v_exp_f32 + v_rcp_f32 against 16 + 2 for erf by Abramowitz-Stegun 7.1.26 - with 16 activations per lane behind every 12 MFMAs
// (C = 96) the GELU stream, not the matrix pipe, paced these kernels (probe: forward 0.36 ms with A&S, 0.22 ms with GELU compiled out).
// The exact-fp32 parity mode never comes here (erff in the unfused chain).
__device__ __forceinline__ f32x2 gelu_pair(f32x2 x) {
#if defined(SV_PROBE_NOGELU)
  return x;
#else
  return gelu_fast2(x);
#endif
}
__device__ __forceinline__ void gelu_both_pair(f32x2 x, f32x2& g, f32x2& dg) {
#if defined(SV_PROBE_NOGELU)
  g = x; dg = (f32x2)(1.f); return;
#else
  gelu_both_fast2(x, g, dg);
#endif
}
__device__ __forceinline__ float gelu_fwd(float x) {
#if defined(SV_PROBE_NOGELU)   // measurement probe only (never built into the library)
  return x;
#else
  const float x2 = fminf(x * x, 64.f);
  const float p = fmaf(x2, fmaf(x2, 9.975397e-04f, -1.0665937e-01f), -2.3012706e+00f);    // -log2(e) * (c0 + c1 x^2 + c2 x^4)
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * p));
#endif
}
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
#if defined(SV_PROBE_NOGELU)
  g = x; dg = 1.f; return;
#endif
  const float x2 = fminf(x * x, 64.f);
  const float p = fmaf(x2, fmaf(x2, 9.975397e-04f, -1.0665937e-01f), -2.3012706e+00f);
  const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * p));
  const float qd = fmaf(x2, fmaf(x2, -3.45720919e-03f, 2.21791923e-01f), 1.59511919f);     // d/dx [x (c0 + c1 x^2 + c2 x^4)]
  g = x * s;
  dg = s * fmaf(x * (1.f - s), qd, 1.f);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {   // registers 8s .. 8s+7 -> one bf16 fragment
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
  return r;
}

// sum over the 32 lanes of each wave half (lanes 0-31 -> lane 31, lanes 32-63 -> lane 63) on the VALU (DPP row shifts + row broadcast)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float half_wave_total(float v) {
  v += dpp_shift<0x111, 0xf>(v);   // row_shr:1
  v += dpp_shift<0x112, 0xf>(v);   // row_shr:2
  v += dpp_shift<0x114, 0xf>(v);   // row_shr:4
  v += dpp_shift<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of every 16-lane row holds the row total
  v += dpp_shift<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the totals of lanes 0-31 / 32-63
  return v;
}

// One 64-hidden-unit weight stage (NIMG images, 64 C elements each) HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): a wave
// instruction moves 1 KB, lane l's 16 bytes landing at (wave-uniform LDS base) + 16 l - exactly the fragment order of the packs, so
// the copy needs no staging registers and runs under the MFMAs of the previous stage.  Completion: the vmcnt(0) that
// __syncthreads() carries while a DMA is in flight.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int C, int NW, int NIMG>
__device__ __forceinline__ void stage_dma(const __bf16* const (&img)[NIMG], int pair, __bf16* stage, int wave, int lane) {
  constexpr int PIECES = NIMG * 8 * C / 64;          // 1-KB pieces per stage (8 C / 64 per image)
#pragma unroll
  for (int pc = wave; pc < PIECES; pc += NW) {
    const int im = pc / (8 * C / 64), o = pc % (8 * C / 64);
    const __bf16* src = img[im] + (size_t)pair * 64 * C + o * 512 + lane * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(stage + (size_t)pc * 512), 16, 0, 0);
  }
}

// ---- forward ---------------------------------------------------------------------------------------------------------------
// Workgroups of 4 waves (128 tokens), two per CU: the two run out of phase, so one's prologue (HBM latency of its rows, LayerNorm)
// and store tail hide under the other's MFMA / GELU loop; with one 8-wave workgroup per CU every wave sat in the same phase.
template <int C, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void swin_mlp_fwd_kernel(const MlpArgs p) {
  constexpr int NF = C / 16, NB = C / 32, HID = 4 * C, NPAIR = HID / 64, NT = NW * 64;
  constexpr int STAGE = 2 * 64 * C;                      // bf16 elements: W1F pair | W2F pair
  // ONE shared object: with a second one beside the LDS-DMA ring hipcc puts s_waitcnt vmcnt(0) in front of the first ds_read of every
  // stage (it must assume the read aliases the DMA in flight) and the copy no longer runs under the MFMAs (cdna_hip_programming.md 5)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE * 2 + HID * 4];
  __bf16* wbuf = reinterpret_cast<__bf16*>(smem);
  float* sb1 = reinterpret_cast<float*>(smem + 2 * STAGE * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long long HC = (long long)HID * C;
  const __bf16* const imgs[2] = {p.packs, p.packs + HC};
  for (int i = tid; i < HID; i += NT) sb1[i] = p.b1[i];

  const long long tok = (long long)blockIdx.x * (NW * 32) + wave * 32 + r;
  const bool valid = tok < p.M;
  // ---- LayerNorm of the wave's 32 token rows, straight into the B fragments of the first product (lane (r, h): 8 consecutive channels)
  bf16x8 xn[NF];
  {
    float xv[NF][8];
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      bf16x8 raw = VecN<__bf16, 8>::zero();
      if (valid) raw = *reinterpret_cast<const bf16x8*>(p.x1 + tok * C + 16 * f + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) { xv[f][j] = (float)raw[j]; s += xv[f][j]; }
    }
    s = lane_step_add<32>(s);
    const float mean = s * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[f][j] - mean; q += d * d; }
    q = lane_step_add<32>(q);
    const float rstd = rsqrtf(q * (1.f / C) + p.eps);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int c0 = 16 * f + 8 * h;
      const float4 g0 = *reinterpret_cast<const float4*>(p.ln_g + c0), g1 = *reinterpret_cast<const float4*>(p.ln_g + c0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.ln_b + c0), b1v = *reinterpret_cast<const float4*>(p.ln_b + c0 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[f][j] = (__bf16)((xv[f][j] - mean) * rstd * gg[j] + bb[j]);
    }
  }

  f32x16 acc2[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[b][i] = 0.f;

  stage_dma<C, NW, 2>(imgs, 0, wbuf, wave, lane);
#pragma unroll 1
  for (int pr = 0; pr < NPAIR; ++pr) {
    const __bf16* st = wbuf + (pr & 1) * STAGE;
#ifndef SV_PROBE_NODMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of stage pr have landed ...
    __syncthreads();                                     // ... and everybody's; everybody is done with the other buffer
    if (pr + 1 < NPAIR) stage_dma<C, NW, 2>(imgs, pr + 1, wbuf + ((pr + 1) & 1) * STAGE, wave, lane);   // lands under the MFMAs below
#endif
    // the two 32-unit sub-chunks of the stage run interleaved: two independent accumulator chains keep the matrix pipe issuing
    f32x16 acc1[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc1[sub][i] = 0.f;
    // every A fragment of the first product is requested before the first MFMA (ds_read latency ~ 4 MFMAs: fetched two at a time in
    // front of their MFMA the matrix pipe idles); the fragments of the second product are requested in front of the GELU block
    bf16x8 wa[2][NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) wa[sub][f] = *reinterpret_cast<const bf16x8*>(st + sub * 32 * C + f * 512 + lane * 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
        acc1[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[sub][f], xn[f], acc1[sub], 0, 0, 0);   // H^T chunk: rows = hidden, columns = tokens
    bf16x8 wb[2][2][NB];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int b = 0; b < NB; ++b) wb[sub][s2][b] = *reinterpret_cast<const bf16x8*>(st + 64 * C + sub * 32 * C + (b * 2 + s2) * 512 + lane * 8);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 hb[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int hid0 = (2 * pr + sub) * 32;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {      // pairs: packed fp32 math (common.h gelu_fast2)
        const int hb = hid0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        const f32x2 gl = gelu_pair((f32x2){acc1[sub][i] + sb1[hb], acc1[sub][i + 1] + sb1[hb + 1]});
        acc1[sub][i] = gl[0]; acc1[sub][i + 1] = gl[1];
      }
      hb[sub][0] = pack8(acc1[sub], 0); hb[sub][1] = pack8(acc1[sub], 1);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[sub][s2][b], hb[sub][s2], acc2[b], 0, 0, 0);   // Y^T: rows = output channels, columns = tokens
  }
  // ---- epilogue: + bias, drop-path scale, + residual; lane (r, h) owns channels 32 b + 8 g + 4 h .. + 3 of token r
  if (valid) {
    const float sc = p.row_scale ? p.row_scale[tok / p.rows_per_scale] : 1.f;
    bf16x4 resv[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) resv[b][g] = *reinterpret_cast<const bf16x4*>(p.x1 + tok * C + 32 * b + 8 * g + 4 * h);   // all in flight at once
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * b + 8 * g + 4 * h;
        const float4 bi = *reinterpret_cast<const float4*>(p.b2 + c0);
        const bf16x4 res = resv[b][g];
        bf16x4 o;
        o[0] = (__bf16)((float)res[0] + sc * (acc2[b][4 * g + 0] + bi.x));
        o[1] = (__bf16)((float)res[1] + sc * (acc2[b][4 * g + 1] + bi.y));
        o[2] = (__bf16)((float)res[2] + sc * (acc2[b][4 * g + 2] + bi.z));
        o[3] = (__bf16)((float)res[3] + sc * (acc2[b][4 * g + 3] + bi.w));
#ifdef SV_PROBE_NOSTORE
        if (o[0] == (__bf16)123.f) *reinterpret_cast<bf16x4*>(p.out + tok * C + c0) = o;
#else
        *reinterpret_cast<bf16x4*>(p.out + tok * C + c0) = o;
#endif
      }
  }
}

// ---- data gradient ------------------------------------------------------------------------------------------------------
// dx1 = dx2 + LayerNormBackward( (s * dx2 . W2) * GELU'(hpre) . W1 ),  dgamma / dbeta of the LayerNorm accumulated per workgroup
template <int C, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void swin_mlp_bwd_kernel(const MlpArgs p) {
  constexpr int NF = C / 16, NB = C / 32, HID = 4 * C, NPAIR = HID / 64, NT = NW * 64;
  constexpr int STAGE = 3 * 64 * C;                      // W1F pair | W2TF pair | W1TF pair
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE * 2 + (HID + 4 * C) * 4];   // one object (see the forward kernel)
  __bf16* wbuf = reinterpret_cast<__bf16*>(smem);
  float* sb1 = reinterpret_cast<float*>(smem + 2 * STAGE * 2);
  float* sdg = sb1 + HID;
  float* sdb = sdg + C;
  // LayerNorm parameters from LDS: read from global memory they are invariant over the persistent tile loop, hipcc hoists all 3 C / 2
  // per-lane values out of it and spills them around the main loop
  float* sgam = sdb + C;
  float* sbet = sgam + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long long HC = (long long)HID * C;
  const __bf16* const imgs[3] = {p.packs, p.packs + 2 * HC, p.packs + 3 * HC};
  for (int i = tid; i < HID; i += NT) sb1[i] = p.b1[i];
  for (int i = tid; i < C; i += NT) { sdg[i] = 0.f; sdb[i] = 0.f; sgam[i] = p.ln_g[i]; sbet[i] = p.ln_b[i]; }
  __syncthreads();

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const long long tok = (long long)tile * (NW * 32) + wave * 32 + r;
    const bool valid = tok < p.M;
    const float sc = (valid && p.row_scale) ? p.row_scale[tok / p.rows_per_scale] : 1.f;
    bf16x8 xn[NF], dyb[NF];
    float mean, rstd;
    {
      float xv[NF][8];
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        bf16x8 raw = VecN<__bf16, 8>::zero(), dr = VecN<__bf16, 8>::zero();
        if (valid) {
          raw = *reinterpret_cast<const bf16x8*>(p.x1 + tok * C + 16 * f + 8 * h);
          dr = *reinterpret_cast<const bf16x8*>(p.dx2 + tok * C + 16 * f + 8 * h);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[f][j] = (float)raw[j]; s += xv[f][j]; dyb[f][j] = (__bf16)(sc * (float)dr[j]); }
      }
      s = lane_step_add<32>(s);
      mean = s * (1.f / C);
      float q = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = xv[f][j] - mean; q += d * d; }
      q = lane_step_add<32>(q);
      rstd = rsqrtf(q * (1.f / C) + p.eps);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int c0 = 16 * f + 8 * h;
        const float4 g0 = *reinterpret_cast<const float4*>(sgam + c0), g1 = *reinterpret_cast<const float4*>(sgam + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(sbet + c0), b1v = *reinterpret_cast<const float4*>(sbet + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) xn[f][j] = (__bf16)((xv[f][j] - mean) * rstd * gg[j] + bb[j]);
      }
    }
    f32x16 acc3[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc3[b][i] = 0.f;

    __syncthreads();                                     // previous tile: every wave has left its last stage
    stage_dma<C, NW, 3>(imgs, 0, wbuf, wave, lane);
#pragma unroll 1
    for (int pr = 0; pr < NPAIR; ++pr) {
      const __bf16* st = wbuf + (pr & 1) * STAGE;
#ifndef SV_PROBE_NODMA
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (pr + 1 < NPAIR) stage_dma<C, NW, 3>(imgs, pr + 1, wbuf + ((pr + 1) & 1) * STAGE, wave, lane);
#endif
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        f32x16 acc1, accd;                               // two independent chains (pre-activation, dy . W2) interleave on the matrix pipe
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[i] = 0.f; accd[i] = 0.f; }
        const __bf16* w1f = st + sub * 32 * C;
        const __bf16* w2tf = st + 64 * C + sub * 32 * C;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1f + f * 512 + lane * 8);
          const bf16x8 at = *reinterpret_cast<const bf16x8*>(w2tf + f * 512 + lane * 8);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xn[f], acc1, 0, 0, 0);     // pre-activation^T (recomputed)
          accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, dyb[f], accd, 0, 0, 0);   // (dy . W2)^T
        }
        const int hid0 = (2 * pr + sub) * 32;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int hb = hid0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          f32x2 g, dg;
          gelu_both_pair((f32x2){acc1[i] + sb1[hb], acc1[i + 1] + sb1[hb + 1]}, g, dg);
          accd[i] *= dg[0]; accd[i + 1] *= dg[1];
        }
        const bf16x8 d0 = pack8(accd, 0), d1 = pack8(accd, 1);
        __builtin_amdgcn_sched_barrier(0);               // do not hoist the next fragments above the GELU block (register pressure)
        const __bf16* w1tf = st + 128 * C + sub * 32 * C;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(w1tf + (b * 2 + 0) * 512 + lane * 8);
          acc3[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, d0, acc3[b], 0, 0, 0);  // dLN^T: rows = channels, columns = tokens
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(w1tf + (b * 2 + 1) * 512 + lane * 8);
          acc3[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d1, acc3[b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- LayerNorm backward on the accumulator (lane (r, h): channels 32 b + 8 g + 4 h + k of token r)
    // lane offset laundered per tile: otherwise hipcc materialises all 3 C / 2 per-lane LDS addresses of the epilogue once, outside the
    // tile loop, and spills them around the main loop; with an opaque base they stay immediate offsets of the ds instructions
    int lane_off = 4 * h;
    asm volatile("" : "+v"(lane_off));
    float* const lg_ = sgam + lane_off;
    float* const ldg_ = sdg + lane_off;
    float* const ldb_ = sdb + lane_off;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * b + 8 * g + 4 * h;
        bf16x4 xr = V4<__bf16>::zero();
        if (valid) xr = *reinterpret_cast<const bf16x4*>(p.x1 + tok * C + c0);
        const float4 gm = *reinterpret_cast<const float4*>(lg_ + 32 * b + 8 * g);
        const float gv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = valid ? ((float)xr[k] - mean) * rstd : 0.f;
          const float d = valid ? acc3[b][4 * g + k] : 0.f;
          const float gg = d * gv[k];
          s1 += gg; s2 += gg * xh;
          // dgamma / dbeta: totals over the 32 tokens of this lane half arrive in lanes 31 and 63
#ifdef SV_PROBE_NODPP
          const float tb = d, tg = d * xh;
#else
          const float tb = half_wave_total(d), tg = half_wave_total(d * xh);
#endif
          if (r == 31) { atomicAdd(ldb_ + 32 * b + 8 * g + k, tb); atomicAdd(ldg_ + 32 * b + 8 * g + k, tg); }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the 8 reduction chains of one channel group together (interleaving all 96 spills)
      }
    s1 = lane_step_add<32>(s1);
    s2 = lane_step_add<32>(s2);
    s1 *= (1.f / C); s2 *= (1.f / C);
    if (valid) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = 32 * b + 8 * g + 4 * h;
          const bf16x4 xr = *reinterpret_cast<const bf16x4*>(p.x1 + tok * C + c0);
          const bf16x4 dr = *reinterpret_cast<const bf16x4*>(p.dx2 + tok * C + c0);
          const float4 gm = *reinterpret_cast<const float4*>(lg_ + 32 * b + 8 * g);
          const float gv[4] = {gm.x, gm.y, gm.z, gm.w};
          bf16x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xh = ((float)xr[k] - mean) * rstd;
            o[k] = (__bf16)((float)dr[k] + rstd * (acc3[b][4 * g + k] * gv[k] - s1 - xh * s2));
          }
          *reinterpret_cast<bf16x4*>(p.out + tok * C + c0) = o;
        }
    }
  }
  __syncthreads();
  for (int i = tid; i < C; i += NT) { atomicAdd(p.dgamma + i, sdg[i]); atomicAdd(p.dbeta + i, sdb[i]); }
}

// ---- weight gradients --------------------------------------------------------------------------------------------------------
struct MlpWArgs {
  const __bf16* x1; const __bf16* dx2;
  const float* ln_g; const float* ln_b; float eps;
  const __bf16* w1r;   // [4C][C]  fc1 weight rows (forward pack of fc1)
  const __bf16* w2tr;  // [4C][C]  fc2 weight transposed (data-gradient pack of fc2)
  const float* b1; const float* row_scale; int rows_per_scale;
  float* dw1; float* db1; float* dw2; float* db2;     // native layouts [4C][C], [4C], [C][4C], [C]
  long long M; int tok_per_split; int HG;
};

// Tiles: TT token rows of x1 and dx2 per step, DOUBLE-BUFFERED in LDS and filled by LDS-DMA while the previous tile is being
// contracted.  A row record is the C/8 16-byte units of the row plus one pad unit (row stride = 8 banks mod 64: the fragment reads
// below are conflict-free); one DMA instruction moves 64 / (C/8 + 1) whole records (the pad unit re-reads the row's last unit), so
// the records stay contiguous as LDS-DMA requires.  LayerNorm (4 lanes per row) and the drop-path scale of dy then run in place.
template <int C, int NW, int HCW, int TT>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void swin_mlp_wgrad_kernel(const MlpWArgs p) {
  constexpr int NF = C / 32, NCB = C / 16, NHB = HCW / 16, HID = 4 * C, NT = NW * 64;
  constexpr int UPR = C / 8, RU = UPR + 1, LD = RU * 8, RPI = 64 / RU, NGRP = (TT + RPI - 1) / RPI, TILE = TT * LD;
  // ONE shared object (see the forward kernel), and no ordinary global load inside the token loop: hipcc would drain the DMA of the next
  // tile in front of its first use - LayerNorm parameters sit in LDS, the drop-path factors of a tile are fetched before its DMA wait
  __shared__ __attribute__((aligned(16))) char smem_raw[4 * TILE * 2 + 2 * C * 4];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);            // [buffer][x | dy][TT][LD]
  float* sgam = reinterpret_cast<float*>(smem_raw + 4 * TILE * 2);
  float* sbet = sgam + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4, q = lr >> 2, pp = lr & 3;
  for (int i = tid; i < C; i += NT) { sgam[i] = p.ln_g[i]; sbet[i] = p.ln_b[i]; }
  const int hg = blockIdx.x % p.HG, split = blockIdx.x / p.HG;
  const int h0 = (hg * NW + wave) * HCW;                         // first hidden unit of this wave
  const long long t_begin = (long long)split * p.tok_per_split;
  long long t_end = t_begin + p.tok_per_split;
  if (t_end > p.M) t_end = p.M;

  auto tile_dma = [&](long long t0, int buf) {
    const int rr = lane / RU, uu = lane % RU;
#pragma unroll 1
    for (int g = wave; g < NGRP; g += NW) {
      const int row = g * RPI + rr;
      if (rr < RPI && row < TT) {
        long long tok = t0 + row;
        if (tok >= t_end) tok = t_end - 1;                       // rows past the end are zeroed by the LayerNorm pass
        const size_t off = (size_t)tok * C + (uu < UPR ? uu : UPR - 1) * 8;
        __bf16* dst = smem + (size_t)(2 * buf) * TILE + (size_t)g * RPI * LD;
        __builtin_amdgcn_global_load_lds((gptr_t)(p.x1 + off), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(p.dx2 + off), (lptr_t)(dst + TILE), 16, 0, 0);
      }
    }
  };

  // per-wave constant B fragments: W1 rows / W2^T rows of the wave's hidden units (k = channels, 8 consecutive per lane)
  bf16x8 w1b[NHB][NF], w2b[NHB][NF];
  float b1v[NHB];
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) {
    const int hid = h0 + 16 * hb + lr;
    b1v[hb] = p.b1[hid];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      w1b[hb][f] = *reinterpret_cast<const bf16x8*>(p.w1r + (size_t)hid * C + 32 * f + 8 * lg);
      w2b[hb][f] = *reinterpret_cast<const bf16x8*>(p.w2tr + (size_t)hid * C + 32 * f + 8 * lg);
    }
  }
  f32x4 G2[NCB][NHB], G1[NCB][NHB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb) { G2[cb][hb] = (f32x4){0.f, 0.f, 0.f, 0.f}; G1[cb][hb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float db1acc[NHB];
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) db1acc[hb] = 0.f;
  float db2acc = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the fragment loads above are ordinary loads: retire them before any DMA
  tile_dma(t_begin, 0);
  int buf = 0;
#pragma unroll 1
  for (long long t0 = t_begin; t0 < t_end; t0 += TT, buf ^= 1) {
    __bf16* xs = smem + (size_t)(2 * buf) * TILE;
    __bf16* ds = xs + TILE;
    // drop-path factors of this tile: it spans at most two images (rows_per_scale >= TT is checked on the host)
    float sc_lo = 1.f, sc_hi = 1.f;
    long long t_img = t_end;                                     // first token of the second image inside the tile
    if (p.row_scale) {
      const long long i_lo = t0 / p.rows_per_scale;
      long long t_last = t0 + TT - 1; if (t_last >= t_end) t_last = t_end - 1;
      sc_lo = p.row_scale[i_lo]; sc_hi = p.row_scale[t_last / p.rows_per_scale];
      t_img = (i_lo + 1) * (long long)p.rows_per_scale;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's records of tile t0 have landed ...
    __syncthreads();                                             // ... and everybody's; the other buffer has been consumed
    // ---- in place: LayerNorm of x1 rows (4 lanes per row), drop-path scale of dy rows, zero rows past the end
    for (int row = tid >> 2; row < TT; row += NT / 4) {
      const int part = tid & 3;
      const bool live = t0 + row < t_end;
      __bf16* xr = xs + row * LD + part * (C / 4);
      __bf16* dr = ds + row * LD + part * (C / 4);
      bf16x8 v[C / 32];
      float s1 = 0.f;
#pragma unroll
      for (int u = 0; u < C / 32; ++u) {
        v[u] = *reinterpret_cast<const bf16x8*>(xr + 8 * u);
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += (float)v[u][j];
      }
      s1 += dpp_move<0xB1>(s1); s1 += dpp_move<0x4E>(s1);
      const float mean = s1 * (1.f / C);
      float s2 = 0.f;
#pragma unroll
      for (int u = 0; u < C / 32; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (float)v[u][j] - mean; s2 += d * d; }
      s2 += dpp_move<0xB1>(s2); s2 += dpp_move<0x4E>(s2);
      const float rstd = rsqrtf(s2 * (1.f / C) + p.eps);
      const float sc = t0 + row < t_img ? sc_lo : sc_hi;
#pragma unroll
      for (int u = 0; u < C / 32; ++u) {
        const int c0 = part * (C / 4) + 8 * u;
        const float4 g0 = *reinterpret_cast<const float4*>(sgam + c0), g1 = *reinterpret_cast<const float4*>(sgam + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(sbet + c0), b1q = *reinterpret_cast<const float4*>(sbet + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1q.x, b1q.y, b1q.z, b1q.w};
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = live ? (__bf16)(((float)v[u][j] - mean) * rstd * gg[j] + bb[j]) : (__bf16)0.f;
        *reinterpret_cast<bf16x8*>(xr + 8 * u) = o;
        if (!live || p.row_scale) {
          bf16x8 d = *reinterpret_cast<const bf16x8*>(dr + 8 * u);
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = live ? (__bf16)(sc * (float)d[j]) : (__bf16)0.f;
          *reinterpret_cast<bf16x8*>(dr + 8 * u) = d;
        }
      }
    }
    __syncthreads();
    if (t0 + TT < t_end) tile_dma(t0 + TT, buf ^ 1);             // the next tile lands under the contraction below
    if (hg == 0 && tid < C) {                                    // bias gradient of fc2: column sums of the (scaled) dy tile
      float s = 0.f;
#pragma unroll 8
      for (int row = 0; row < TT; ++row) s += (float)ds[row * LD + tid];
      db2acc += s;
    }
    // ---- the wave's hidden chunk against the 32-token sub-tiles
#pragma unroll 1
    for (int sub = 0; sub < TT / 32; ++sub) {
      f32x4 Hh[2][NHB], Dh[2][NHB];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        bf16x8 xa[NF], da[NF];
        const int row = 32 * sub + 16 * tb + lr;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          xa[f] = *reinterpret_cast<const bf16x8*>(xs + row * LD + 32 * f + 8 * lg);
          da[f] = *reinterpret_cast<const bf16x8*>(ds + row * LD + 32 * f + 8 * lg);
        }
#pragma unroll
        for (int hb = 0; hb < NHB; ++hb) {
          f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[f], w1b[hb][f], a, 0, 0, 0);   // rows = tokens, columns = hidden units
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[f], w2b[hb][f], d, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; i += 2) {
            f32x2 g, dg;
            gelu_both_pair((f32x2){a[i] + b1v[hb], a[i + 1] + b1v[hb]}, g, dg);
            a[i] = g[0]; a[i + 1] = g[1];
            d[i] *= dg[0]; d[i + 1] *= dg[1];
            db1acc[hb] += d[i] + d[i + 1];
          }
          Hh[tb][hb] = a; Dh[tb][hb] = d;
        }
      }
      // B fragments of the token contraction: the stacked accumulator blocks (k order {4 lg + i} U {16 + 4 lg + i})
      bf16x8 hB[NHB], dB[NHB];
#pragma unroll
      for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          hB[hb][i] = (__bf16)Hh[0][hb][i]; hB[hb][4 + i] = (__bf16)Hh[1][hb][i];
          dB[hb][i] = (__bf16)Dh[0][hb][i]; dB[hb][4 + i] = (__bf16)Dh[1][hb][i];
        }
      // A fragments: transposed tiles (rows = channels, k = tokens in the same order) by ds_read_b64_tr_b16
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const __bf16* sx = xs + (32 * sub + 4 * lg + q) * LD + 16 * cb + 4 * pp;
        const __bf16* sd = ds + (32 * sub + 4 * lg + q) * LD + 16 * cb + 4 * pp;
        const bf16x4 xlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(sx));
        const bf16x4 xhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(sx + 16 * LD));
        const bf16x4 dlo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(sd));
        const bf16x4 dhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(sd + 16 * LD));
        const bf16x8 xT = __builtin_shufflevector(xlo, xhi, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 dT = __builtin_shufflevector(dlo, dhi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int hb = 0; hb < NHB; ++hb) {
          G2[cb][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT, hB[hb], G2[cb][hb], 0, 0, 0);   // dW2[c][hid]
          G1[cb][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xT, dB[hb], G1[cb][hb], 0, 0, 0);   // dW1[hid][c], held transposed
        }
      }
    }
  }
  // ---- write out (accumulator: row = channel 16 cb + 4 lg + i, column = hidden unit h0 + 16 hb + lr)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 16 * cb + 4 * lg + i, hid = h0 + 16 * hb + lr;
        atomicAdd(p.dw2 + (size_t)c * HID + hid, G2[cb][hb][i]);
        atomicAdd(p.dw1 + (size_t)hid * C + c, G1[cb][hb][i]);
      }
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) {
    float v = db1acc[hb];
    v = lane_step_add<16>(v);
    v = lane_step_add<32>(v);
    if (lg == 0) atomicAdd(p.db1 + h0 + 16 * hb + lr, v);
  }
  if (hg == 0 && tid < C) atomicAdd(p.db2 + tid, db2acc);
}

static inline bool mlp_supported(int C) { return C == 96 || C == 128 || C == 192; }

}  // namespace sv

using namespace sv;
#define STREAM static_cast<hipStream_t>(stream)

extern "C" int sv_swin_mlp_supported(int C) { return mlp_supported(C) ? 1 : 0; }

extern "C" int sv_swin_mlp_pack(const float* w1, const float* w2, void* packs, int C, void* stream) {
  SV_REQUIRE(w1 && w2 && packs && mlp_supported(C), "swin_mlp_pack: bad arguments (C=%d)", C);
  const long long total = 16ll * C * C;
  hipLaunchKernelGGL(swin_mlp_pack_kernel, dim3(cdiv(total, 256 * 4)), dim3(256), 0, STREAM, w1, w2, static_cast<__bf16*>(packs), C);
  return check_launch("sv_swin_mlp_pack");
}

static int mlp_check(const void* x1, const void* out, const float* g, const float* b, const void* packs, long long M, int C, int rps) {
  SV_REQUIRE(x1 && out && g && b && packs && M > 0 && M < (1ll << 31) && mlp_supported(C), "swin_mlp: bad arguments (M=%lld C=%d)", M, C);
  SV_REQUIRE((((uintptr_t)x1 | (uintptr_t)out | (uintptr_t)packs) & 15) == 0 && (((uintptr_t)g | (uintptr_t)b) & 15) == 0, "swin_mlp: operands must be 16-byte aligned");
  SV_REQUIRE(rps > 0, "swin_mlp: rows_per_scale must be positive");
  return SV_OK;
}

extern "C" int sv_swin_mlp_fwd(const void* x1, void* x2, const float* ln_g, const float* ln_b, const void* packs, const float* b1,
                               const float* b2, const float* row_scale, int rows_per_scale, long long M, int C, float eps, void* stream) {
  if (int rc = mlp_check(x1, x2, ln_g, ln_b, packs, M, C, rows_per_scale)) return rc;
  SV_REQUIRE(b1 && b2 && (((uintptr_t)b2) & 15) == 0, "swin_mlp_fwd: biases required (16-byte aligned)");
  MlpArgs a{};
  a.x1 = static_cast<const __bf16*>(x1); a.out = static_cast<__bf16*>(x2); a.ln_g = ln_g; a.ln_b = ln_b; a.eps = eps;
  a.packs = static_cast<const __bf16*>(packs); a.b1 = b1; a.b2 = b2; a.row_scale = row_scale; a.rows_per_scale = rows_per_scale; a.M = M;
#define SV_MLP_FWD(CC, NW, WPS) hipLaunchKernelGGL((swin_mlp_fwd_kernel<CC, NW, WPS>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), 0, STREAM, a)
  if (C == 96) SV_MLP_FWD(96, 4, 2);
  else if (C == 128) SV_MLP_FWD(128, 4, 2);
  else SV_MLP_FWD(192, 4, 1);
#undef SV_MLP_FWD
  return check_launch("sv_swin_mlp_fwd");
}

extern "C" int sv_swin_mlp_bwd(const void* x1, const void* dx2, void* dx1, const float* ln_g, const float* ln_b, const void* packs,
                               const float* b1, const float* row_scale, int rows_per_scale, float* dgamma, float* dbeta, long long M, int C,
                               float eps, void* stream) {
  if (int rc = mlp_check(x1, dx1, ln_g, ln_b, packs, M, C, rows_per_scale)) return rc;
  SV_REQUIRE(dx2 && b1 && dgamma && dbeta && (((uintptr_t)dx2) & 15) == 0, "swin_mlp_bwd: bad arguments");
  MlpArgs a{};
  a.x1 = static_cast<const __bf16*>(x1); a.dx2 = static_cast<const __bf16*>(dx2); a.out = static_cast<__bf16*>(dx1);
  a.ln_g = ln_g; a.ln_b = ln_b; a.eps = eps; a.packs = static_cast<const __bf16*>(packs); a.b1 = b1;
  a.row_scale = row_scale; a.rows_per_scale = rows_per_scale; a.dgamma = dgamma; a.dbeta = dbeta; a.M = M;
#define SV_MLP_BWD(CC, NW, WPS)                                                                           \
  do {                                                                                                    \
    a.ntiles = cdiv(M, NW * 32);                                                                          \
    int grid = 256 * WPS; if (grid > a.ntiles) grid = a.ntiles;                                           \
    hipLaunchKernelGGL((swin_mlp_bwd_kernel<CC, NW, WPS>), dim3(grid), dim3(NW * 64), 0, STREAM, a);      \
  } while (0)
  if (C == 96) SV_MLP_BWD(96, 4, 2);
  else if (C == 128) SV_MLP_BWD(128, 4, 1);
  else SV_MLP_BWD(192, 4, 1);
#undef SV_MLP_BWD
  return check_launch("sv_swin_mlp_bwd");
}

extern "C" int sv_swin_mlp_wgrad(const void* x1, const void* dx2, const float* ln_g, const float* ln_b, const void* w1_rows,
                                 const void* w2t_rows, const float* b1, const float* row_scale, int rows_per_scale, float* dw1, float* db1,
                                 float* dw2, float* db2, long long M, int C, float eps, void* stream) {
  SV_REQUIRE(x1 && dx2 && ln_g && ln_b && w1_rows && w2t_rows && b1 && dw1 && db1 && dw2 && db2 && M > 0 && M < (1ll << 31) && mlp_supported(C),
             "swin_mlp_wgrad: bad arguments (M=%lld C=%d)", M, C);
  SV_REQUIRE((((uintptr_t)x1 | (uintptr_t)dx2 | (uintptr_t)w1_rows | (uintptr_t)w2t_rows) & 15) == 0 && rows_per_scale > 0, "swin_mlp_wgrad: alignment");
  SV_REQUIRE(!row_scale || rows_per_scale >= 128, "swin_mlp_wgrad: rows_per_scale (%d) must be >= 128 (a token tile spans at most two scale groups)", rows_per_scale);
  MlpWArgs a{};
  a.x1 = static_cast<const __bf16*>(x1); a.dx2 = static_cast<const __bf16*>(dx2); a.ln_g = ln_g; a.ln_b = ln_b; a.eps = eps;
  a.w1r = static_cast<const __bf16*>(w1_rows); a.w2tr = static_cast<const __bf16*>(w2t_rows); a.b1 = b1;
  a.row_scale = row_scale; a.rows_per_scale = rows_per_scale; a.dw1 = dw1; a.db1 = db1; a.dw2 = dw2; a.db2 = db2; a.M = M;
#define SV_MLP_WG(CC, NW, HCW, TT)                                                                              \
  do {                                                                                                          \
    a.HG = (4 * CC) / (NW * HCW);                                                                               \
    long long splits = 256 / a.HG; if (splits < 1) splits = 1;                                                  \
    long long tps = (M + splits - 1) / splits; tps = (tps + TT - 1) / TT * TT;                                  \
    splits = (M + tps - 1) / tps;                                                                               \
    a.tok_per_split = (int)tps;                                                                                 \
    hipLaunchKernelGGL((swin_mlp_wgrad_kernel<CC, NW, HCW, TT>), dim3((unsigned)(splits * a.HG)), dim3(NW * 64), 0, STREAM, a); \
  } while (0)
  if (C == 96) SV_MLP_WG(96, 6, 32, 128);
  else if (C == 128) SV_MLP_WG(128, 8, 16, 128);
  else SV_MLP_WG(192, 8, 16, 64);
#undef SV_MLP_WG
  return check_launch("sv_swin_mlp_wgrad");
}
