// LDS-halo MFMA stencils for the <=16-channel 3x3x3 convolutions of the 32^3 tail (merger, models/merger.py:20-54).
//
// The generic implicit-GEMM engine gathers every (voxel, tap) operand row from L2 (27 x 48 B per voxel) and pads the
// 9 output channels to a 64-wide tile; here a workgroup owns a 4x8x8 brick of voxels, stages the brick + halo ONCE in LDS
// as bf16 [position][16 channels] (32 B rows) and feeds the MFMA straight from it:
//   forward / data-gradient:  out[vox, n] = sum_{tap, c} x[vox + tap, c] * w[n, tap, c]
//       A fragment (voxel rows, 8 consecutive channels of one tap) = ONE 16-byte LDS read; weights [16n][27][16G] in LDS.
//   weight gradient:          dw[co, tap, c] += sum_vox dy[vox, co] * x[vox + tap, c]
//       the contraction runs over VOXELS: both fragments (8 consecutive x-positions per lane) come from the
//       [position][channel] images through ds_read_b64_tr_b16; the four waves split the 27 taps; workgroups are
//       persistent over bricks so the 27 x 81 partial sums reach HBM once per workgroup.
// bf16 operands, fp32 accumulate (these kernels serve set_math("bf16"); exact-fp32 parity runs use the generic engine).
#include "common.h"

namespace sv {

constexpr int TZ = 4, TY = 8, TX = 8;                 // brick of output voxels per workgroup
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;  // with halo
constexpr int HPOS = HZ * HY * HX;                    // 600 positions
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

template <typename AT>
struct StencilArgsT {                         // AT = storage element of the activations (x, out, residual)
  const AT* x; int ldx; int cin_load;         // input positions [I*D*H*W][ldx], cin_load (multiple of 4, <= 16*G) elements loaded per position
  const __bf16* w;                            // packed weights [NT*16][27][16*G]
  const float* bias; AT* out; int ldc; int col_off; int cout;   // columns written: n < cout
  const AT* residual; int ldr;                // optional: out = residual + val (same column window)
  double* stats;                              // optional [SV_BN_SLOTS][2*cout]
  int I, D, H, W;
};

// stage the halo brick of one tile into LDS as bf16 [HPOS][16*G]; out-of-volume positions and channels >= cin_load are zero
template <int G, typename AT>
__device__ __forceinline__ void load_halo(__bf16* Xs, const AT* x, int ldx, int cin_load, int img, int z0, int y0, int x0,
                                          int D, int H, int W, int tid) {
  constexpr int C = 16 * G, V4 = C / 4;
  for (int i = tid; i < HPOS * V4; i += 256) {
    const int h = i / V4, v = i - h * V4;
    const int hx = h % HX; const int t = h / HX; const int hy = t % HY; const int hz = t / HY;
    const int z = z0 - 1 + hz, y = y0 - 1 + hy, xx = x0 - 1 + hx;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v * 4 < cin_load && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
      q = ld4f(x + ((((size_t)img * D + z) * H + y) * W + xx) * (size_t)ldx + v * 4);
    st4f(Xs + h * C + v * 4, q);
  }
}

template <int G, int NT, typename AT>
__global__ __launch_bounds__(256) void stencil3_fwd_kernel(const StencilArgsT<AT> p) {
  constexpr int C = 16 * G, KTOT = 27 * C, KPAD = (KTOT + 31) / 32 * 32, NSTEP = KPAD / 32;
  __shared__ __attribute__((aligned(16))) __bf16 Xs[HPOS * C];
  __shared__ __attribute__((aligned(16))) __bf16 Ws[NT * 16 * KPAD];
  __shared__ float red[4 * NT * 16 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int tz = p.D / TZ, ty = p.H / TY, tx = p.W / TX;
  int t = blockIdx.x;
  const int bx = t % tx; t /= tx; const int by = t % ty; t /= ty; const int bz = t % tz; const int img = t / tz;
  const int z0 = bz * TZ, y0 = by * TY, x0 = bx * TX;

  // weights -> LDS (rows padded with zeros to a multiple of 32 k)
  for (int i = tid; i < NT * 16 * KPAD / 8; i += 256) {
    const int n = (i * 8) / KPAD, k = (i * 8) - n * KPAD;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k + j < KTOT) ? p.w[(size_t)n * KTOT + k + j] : (__bf16)0.f;
    *reinterpret_cast<bf16x8*>(Ws + n * KPAD + k) = v;
  }
  load_halo<G, AT>(Xs, p.x, p.ldx, p.cin_load, img, z0, y0, x0, p.D, p.H, p.W, tid);
  __syncthreads();

  // this wave: z-slice `wave`; M-tile mt = rows y = 2mt, 2mt+1; fragment row r = lane&15 -> (yy = r>>3, xx = r&7)
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int yy = lr >> 3, xx = lr & 7;
  int abase[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) abase[mt] = (wave * HY + 2 * mt + yy) * HX + xx;   // halo index of (z, y, x) shifted by tap (0,0,0)
#pragma unroll 2
  for (int s = 0; s < NSTEP; ++s) {
    const int kb = s * 32 + lg * 8;
    int tap = kb / C; const int c = kb - tap * C;
    if (tap > 26) tap = 26;                       // padded k: weights are zero there, any valid address will do
    const int dz = tap / 9, dy = (tap - dz * 9) / 3, dx = tap - dz * 9 - dy * 3;
    const int off = (dz * HY + dy) * HX + dx;
    bf16x8 b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(Ws + (nt * 16 + lr) * KPAD + kb);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Xs + (abase[mt] + off) * C + c);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], acc[mt][nt], 0, 0, 0);
    }
  }

  // epilogue: C element (row = lg*4 + j -> voxel, col = lr -> output channel)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + lr;
    const bool nok = n < p.cout;
    const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = lg * 4 + j;
        const int z = z0 + wave, y = y0 + 2 * mt + (r >> 3), x = x0 + (r & 7);
        if (nok) {
          const size_t pos = (((size_t)img * p.D + z) * p.H + y) * p.W + x;
          float v = acc[mt][nt][j] + bias;
          if (p.residual) v += ldf(p.residual + pos * p.ldr + n);
          stf(p.out + pos * p.ldc + p.col_off + n, v);
          s1 += v; s2 += v * v;
        }
      }
    if (p.stats) {
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (lg == 0) { red[((wave * NT + nt) * 16 + lr) * 2] = s1; red[((wave * NT + nt) * 16 + lr) * 2 + 1] = s2; }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < NT * 16 && tid < p.cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[((w * NT * 16) + tid) * 2]; b += red[((w * NT * 16) + tid) * 2 + 1]; }
      double* st = p.stats + (size_t)(blockIdx.x % SV_BN_SLOTS) * 2 * p.cout;
      atomicAdd(st + tid, (double)a);
      atomicAdd(st + p.cout + tid, (double)b);
    }
  }
}

template <typename AT>
struct StencilWArgsT {
  const AT* x; int ldx; int cin_load;        // gathered operand (conv input), memory channels = 16*G (zero-padded)
  const AT* dy; int lddy; int cout_load;     // anchor operand (output gradient), <= 16 memory channels
  float* dw;                                 // native [cout][cin][27], accumulated with atomics
  int cout, cin;                             // real channel counts
  int c_stride, c_valid;                     // memory channel c -> real channel (c / c_stride) * c_valid + c % c_stride, valid if c % c_stride < c_valid
  int I, D, H, W, ntiles;
};

template <int G, typename AT>
__global__ __launch_bounds__(256) void stencil3_wgrad_kernel(const StencilWArgsT<AT> p) {
  constexpr int C = 16 * G;
  __shared__ __attribute__((aligned(16))) __bf16 Xs[HPOS * C];
  __shared__ __attribute__((aligned(16))) __bf16 Ds[TZ * TY * TX * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4, q = lr >> 2, pp = lr & 3;
  const int tz = p.D / TZ, ty = p.H / TY, tx = p.W / TX;
  // taps of this wave: t = wave + 4*i, i < NTAP
  const int ntap = (27 - wave + 3) / 4;
  f32x4 acc[7][G];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int gg = 0; gg < G; ++gg) acc[i][gg] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int t = tile;
    const int bx = t % tx; t /= tx; const int by = t % ty; t /= ty; const int bz = t % tz; const int img = t / tz;
    const int z0 = bz * TZ, y0 = by * TY, x0 = bx * TX;
    __syncthreads();   // previous tile's LDS reads are done
    load_halo<G, AT>(Xs, p.x, p.ldx, p.cin_load, img, z0, y0, x0, p.D, p.H, p.W, tid);
    for (int i = tid; i < TZ * TY * TX * 4; i += 256) {     // dy brick -> [voxel][16] bf16
      const int v = i >> 2, c4 = i & 3;
      const int xx = v % TX; const int t2 = v / TX; const int yy = t2 % TY; const int zz = t2 / TY;
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 * 4 < p.cout_load)
        qv = ld4f(p.dy + ((((size_t)img * p.D + z0 + zz) * p.H + y0 + yy) * p.W + x0 + xx) * (size_t)p.lddy + c4 * 4);
      st4f(Ds + v * 16 + c4 * 4, qv);
    }
    __syncthreads();
    // 8 chunks of 32 voxels: chunk = (z, half); k = 8*g + j  <->  (y = 4*half + g, x = j)
#pragma unroll 1
    for (int ch = 0; ch < 8; ++ch) {
      const int z = ch >> 1, yb = (ch & 1) * 4 + lg;
      const __bf16* asrc = Ds + (((z * TY + yb) * TX) + q) * 16 + pp * 4;       // rows = 4 consecutive x positions
      const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(asrc));
      const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(asrc + 4 * 16));
      const bf16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i < ntap) {
          const int tap = wave + 4 * i;
          const int dz = tap / 9, dy = (tap - dz * 9) / 3, dx = tap - dz * 9 - dy * 3;
          const int hb = ((z + dz) * HY + yb + dy) * HX + dx + q;
#pragma unroll
          for (int gg = 0; gg < G; ++gg) {
            const __bf16* bsrc = Xs + hb * C + gg * 16 + pp * 4;
            const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(bsrc));
            const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(bsrc + 4 * C));
            const bf16x8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[i][gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i][gg], 0, 0, 0);
          }
        }
      }
    }
  }
  // C element: row = lg*4 + j -> co, col = lr -> memory channel 16*gg + lr
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    if (i < ntap) {
      const int tap = wave + 4 * i;
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const int cm = gg * 16 + lr;
        const int grp = cm / p.c_stride, within = cm - grp * p.c_stride;
        const int ci = grp * p.c_valid + within;
        if (within < p.c_valid && ci < p.cin) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = lg * 4 + j;
            if (co < p.cout) atomicAdd(p.dw + ((size_t)co * p.cin + ci) * 27 + tap, acc[i][gg][j]);
          }
        }
      }
    }
  }
}

}  // namespace sv

using namespace sv;

static int stencil_check(int I, int D, int H, int W) {
  SV_REQUIRE(I > 0 && D % TZ == 0 && H % TY == 0 && W % TX == 0, "stencil3: grid %dx%dx%d must be a multiple of the 4x8x8 brick", D, H, W);
  return SV_OK;
}

extern "C" int sv_stencil3_fwd(const void* x, int ldx, int cin_load, int groups, const void* w_bf16, int ntiles16,
                               const float* bias, void* out, int ldc, int col_off, int cout, const void* residual, int ldr,
                               double* stats, int I, int D, int H, int W, int act_dtype, void* stream) {
  SV_REQUIRE(x && w_bf16 && out, "stencil3_fwd: null argument");
  SV_REQUIRE_ACT(act_dtype);
  if (int rc = stencil_check(I, D, H, W)) return rc;
  SV_REQUIRE(cin_load % 4 == 0 && cin_load <= 16 * groups && ldx % 4 == 0 && ldx >= cin_load, "stencil3_fwd: bad input channels (cin_load=%d ldx=%d groups=%d)", cin_load, ldx, groups);
  SV_REQUIRE(cout > 0 && cout <= 16 * ntiles16 && ldc >= col_off + cout, "stencil3_fwd: bad output window");
  SV_REQUIRE(((uintptr_t)x & (act_dtype == SV_BF16 ? 7 : 15)) == 0 && ((uintptr_t)w_bf16 & 15) == 0, "stencil3_fwd: operands must be aligned to 4 elements");
  const int blocks = I * (D / TZ) * (H / TY) * (W / TX);
  hipStream_t s = (hipStream_t)stream;
  if (!((groups == 1 && ntiles16 == 1) || (groups == 3 && ntiles16 == 1) || (groups == 1 && ntiles16 == 3))) {
    set_error("stencil3_fwd: unsupported (groups=%d, ntiles16=%d)", groups, ntiles16);
    return SV_ERR_INVALID;
  }
  SV_DISPATCH_ACT(act_dtype,
    StencilArgsT<AT> a{static_cast<const AT*>(x), ldx, cin_load, (const __bf16*)w_bf16, bias, static_cast<AT*>(out), ldc, col_off, cout,
                       static_cast<const AT*>(residual), ldr, stats, I, D, H, W};
    if (groups == 1 && ntiles16 == 1) hipLaunchKernelGGL((stencil3_fwd_kernel<1, 1, AT>), dim3(blocks), dim3(256), 0, s, a);
    else if (groups == 3) hipLaunchKernelGGL((stencil3_fwd_kernel<3, 1, AT>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((stencil3_fwd_kernel<1, 3, AT>), dim3(blocks), dim3(256), 0, s, a););
  return check_launch("sv_stencil3_fwd");
}

extern "C" int sv_stencil3_wgrad(const void* x, int ldx, int cin_load, int groups, const void* dy, int lddy, int cout_load,
                                 float* dw, int cout, int cin, int c_stride, int c_valid, int I, int D, int H, int W, int act_dtype,
                                 void* stream) {
  SV_REQUIRE(x && dy && dw, "stencil3_wgrad: null argument");
  SV_REQUIRE_ACT(act_dtype);
  if (int rc = stencil_check(I, D, H, W)) return rc;
  SV_REQUIRE(cin_load % 4 == 0 && cin_load <= 16 * groups && ldx % 4 == 0 && cout_load % 4 == 0 && cout_load <= 16 && lddy % 4 == 0,
             "stencil3_wgrad: bad channel layout");
  SV_REQUIRE(cout > 0 && cout <= 16 && cin > 0 && c_stride > 0 && c_valid > 0, "stencil3_wgrad: bad channel counts");
  SV_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & (act_dtype == SV_BF16 ? 7 : 15)) == 0, "stencil3_wgrad: operands must be aligned to 4 elements");
  SV_REQUIRE(groups == 1 || groups == 3, "stencil3_wgrad: unsupported groups=%d", groups);
  const int ntiles = I * (D / TZ) * (H / TY) * (W / TX);
  const int blocks = ntiles < 1024 ? ntiles : 1024;
  hipStream_t s = (hipStream_t)stream;
  SV_DISPATCH_ACT(act_dtype,
    StencilWArgsT<AT> a{static_cast<const AT*>(x), ldx, cin_load, static_cast<const AT*>(dy), lddy, cout_load, dw, cout, cin, c_stride, c_valid, I, D, H, W, ntiles};
    if (groups == 1) hipLaunchKernelGGL((stencil3_wgrad_kernel<1, AT>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((stencil3_wgrad_kernel<3, AT>), dim3(blocks), dim3(256), 0, s, a););
  return check_launch("sv_stencil3_wgrad");
}
