// LDS-halo MFMA stencils for the <=16-channel 3x3x3 convolutions of the 32^3 tail (merger, models/merger.py:20-54).
//
// The generic implicit-GEMM engine gathers every (voxel, tap) operand row from L2 (27 x 48 B per voxel) and pads the
// 9 output channels to a 64-wide tile; here a workgroup owns 8x8x8 bricks of voxels, stages a brick + halo ONCE in LDS
// as bf16 [position][16 channels] (32 B rows) and feeds the MFMA straight from it:
//   forward / data-gradient:  out[vox, n] = sum_{tap, c} x[vox + tap, c] * w[n, tap, c]
//       A fragment (voxel rows, 8 consecutive channels of one tap) = ONE 16-byte LDS read; weights [16n][27][16G] in LDS.
//   weight gradient:          dw[co, tap, c] += sum_vox dy[vox, co] * x[vox + tap, c]
//       the contraction runs over VOXELS: both fragments (8 consecutive x-positions per lane) come from the
//       [position][channel] images through ds_read_b64_tr_b16; the four waves split the 27 taps.
// Both kernels are PERSISTENT over bricks: weights / partial sums are set up once per workgroup, and the next brick's
// halo is prefetched into registers while the current one is contracted (global latency hidden behind the MFMA loop).
// The forward accumulators are transposed blocks (weight fragment as the MFMA's first operand): a lane holds four consecutive
// channels of one voxel and stores them straight from its registers (8-byte vectors); per-channel statistics stay in registers.
// bf16 operands, fp32 accumulate (these kernels serve set_math("bf16"); exact-fp32 parity runs use the generic engine).
#include "common.h"
#include <map>
#include <mutex>
#include <type_traits>

namespace sv {

constexpr int TZ = 4, TY = 8, TX = 8;                 // brick of output voxels per tile
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;  // with halo
constexpr int HPOS = HZ * HY * HX;                    // 600 positions
constexpr int NVOX = TZ * TY * TX;                    // 256 voxels = one per thread in the epilogue
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

__device__ __forceinline__ bf16x4 to_bf16x4(float4 v) {
  bf16x4 b;
  b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
  return b;
}
__device__ __forceinline__ bf16x4 to_bf16x4(bf16x4 v) { return v; }

// First brick of a persistent workgroup.  Workgroup ids are dealt round-robin over the 8 XCDs; giving XCD k the k-th
// CONTIGUOUS eighth of every round of gridDim.x bricks keeps neighbouring bricks - which share 2.3x halo data - behind one L2.
__device__ __forceinline__ int xcd_first_tile() {
  const int n = gridDim.x, b = blockIdx.x;
  return (n & 7) == 0 ? (b & 7) * (n >> 3) + (b >> 3) : b;
}

struct TileId { int img, z0, y0, x0; };
template <int TZV = TZ>
__device__ __forceinline__ TileId tile_of(int t, int D, int H, int W) {
  const int tz = D / TZV, ty = H / TY, tx = W / TX;
  TileId r;
  r.x0 = (t % tx) * TX; t /= tx; r.y0 = (t % ty) * TY; t /= ty; r.z0 = (t % tz) * TZV; r.img = t / tz;
  return r;
}

// one thread's share of a halo brick [HPOS][16*G], kept as raw 4-element vectors between the prefetch and the LDS store.
// What does not depend on the brick is worked out ONCE per thread (init): the element offset of every vector from the brick's
// corner and which faces of the halo box it lies on.  A brick then costs, per vector, one AND + compare (does a face it lies on stick
// out of the volume?), one 64-bit add and the load - the div / mod chains and 64-bit multiplies that used to run per vector and brick
// were the bulk of the kernel's instructions (the 9 -> 9 forward: 3,000 instructions around 8 MFMAs).
template <int G, typename AT, int TZV = TZ, int NTHR = 256>   // TZV = brick depth (z-slices = waves of the forward kernel), NTHR = threads sharing the brick
struct HaloRegs {
  static constexpr int HZV = TZV + 2, HPOSV = HZV * HY * HX;
  static constexpr int C = 16 * G, VPP = C / 4, N = (HPOSV * VPP + NTHR - 1) / NTHR;
  typedef typename std::conditional<G == 1, int, long long>::type OffT;   // planes (G > 1) can lie > 2^31 elements apart
  typename V4<AT>::type r[N];
  OffT off[N];      // element offset from the brick's corner position (z0 - 1, y0 - 1, x0 - 1), channel / plane offset included
  int face[N];      // bit 0 / 1: on the z-low / z-high face of the halo box, 2 / 3: y, 4 / 5: x; bit 6: never loaded
  int lds[N];       // element offset of the vector in the LDS image [HZ * HY rows][pitch positions][C] (pitch >= HX, see init)
  // plane == 0: position rows of ldx channels; plane != 0: the channels are stored as planes of ldx channels each
  // (memory channel c at x[(c / ldx) * plane + pos * ldx + c % ldx]) - the dense per-layer buffers of the merger's concat
  __device__ __forceinline__ void init(int ldx, long long plane, int cin_load, int H, int W, int tid, int pitch = HX) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int i = tid + NTHR * k;
      const int h = i / VPP, v = i - h * VPP;
      const int hx = h % HX; const int t2 = h / HX; const int hy = t2 % HY; const int hz = t2 / HY;
      const int cv = v * 4, pl = plane ? cv / ldx : 0;
      face[k] = (hz == 0 ? 1 : 0) | (hz == HZV - 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == HY - 1 ? 8 : 0) | (hx == 0 ? 16 : 0) | (hx == HX - 1 ? 32 : 0) |
                ((i < HPOSV * VPP && cv < cin_load) ? 0 : 64);
      off[k] = (OffT)pl * (OffT)plane + (OffT)(((hz * H + hy) * W + hx) * ldx + (cv - pl * ldx));
      lds[k] = ((hz * HY + hy) * pitch + hx) * C + cv;
    }
  }
  __device__ __forceinline__ void load(const AT* __restrict__ x, int ldx, const TileId& t, int D, int H, int W) {
    // faces of this brick's halo box that stick out of the volume (+ the never-loaded flag)
    const int out = (t.z0 == 0 ? 1 : 0) | (t.z0 + TZV == D ? 2 : 0) | (t.y0 == 0 ? 4 : 0) | (t.y0 + TY == H ? 8 : 0) | (t.x0 == 0 ? 16 : 0) | (t.x0 + TX == W ? 32 : 0) | 64;
    const AT* corner = x + ((((long long)t.img * D + t.z0 - 1) * H + t.y0 - 1) * W + t.x0 - 1) * (long long)ldx;   // may point before the volume: only in-range vectors are read
#pragma unroll
    for (int k = 0; k < N; ++k) {
      typename V4<AT>::type q = V4<AT>::zero();
#ifndef SV_ST_PROBE_NOLOAD
      if ((face[k] & out) == 0) q = V4<AT>::load(corner + off[k]);
#endif
      r[k] = q;
    }
  }
  __device__ __forceinline__ void store(__bf16* Xs, int tid) const {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int i = tid + NTHR * k;
      if (i < HPOSV * VPP) *reinterpret_cast<bf16x4*>(Xs + lds[k]) = to_bf16x4(r[k]);
    }
  }
};

template <typename AT>
struct StencilArgsT {                         // AT = storage element of the activations (x, out, residual)
  const AT* x; int ldx; int cin_load;         // input positions [I*D*H*W][ldx], cin_load (multiple of 4, <= 16*G) elements loaded per position
  long long x_plane, out_plane;               // != 0: input channels / output columns stored as planes of ldx / ldc channels (see HaloRegs::load)
  const __bf16* w;                            // packed weights [NT*16][27][16*G]
  const float* bias; AT* out; int ldc; int col_off; int cout;   // columns written: n < cout (+ zero pads up to a multiple of 4, see header)
  const AT* residual; int ldr;                // optional: out = residual + val (same column window)
  double* stats;                              // optional [SV_BN_SLOTS][2*cout]
  int I, D, H, W, ntiles;
};

// VEC = every output row window is 4-aligned and has room for the zero pads (the merger's buffers): 8 / 16-byte row stores only
// Rows of the LDS halo image hold HXP = 16 positions for 16-channel rows (10 are used): the two y-rows of an M-tile are then 512 bytes apart and
// the four 16-lane groups of an A-fragment ds_read_b128 touch 16 different bank slots each; with the natural pitch of 10 positions (320 B) the
// rows overlap in half the slots - 2-way conflicts on the reads that bound the MFMA loop (one fragment read per MFMA).  Costs 11.5 KB of LDS:
// 3 instead of 4 resident workgroups per CU for G = 1.
template <int G> struct StencilPitch { static constexpr int HXP = G == 1 ? 16 : HX; };
// Brick depth = wave count of the workgroup (in the forward kernel a wave owns one z-slice of 64 voxels).  48-channel rows (G = 3, the 36 -> 9
// layer): with 4 x 8 x 8 bricks the 58 KB halo + 42 KB of weights allowed ONE 4-wave workgroup per CU; 8 x 8 x 8 bricks on 8 waves (96 KB halo)
// give two waves per SIMD and 1.95 instead of 2.34 halo positions loaded per voxel (forward 1.62 -> 1.15 ms, weight gradient 1.59 -> 1.19 ms).
template <int G> struct StencilBrick { static constexpr int TZV = 8, NTHR = TZV * 64; };   // weight-gradient kernel (16-channel rows: 537 -> 495 us against 4 x 8 x 8 on 4 waves)
// The forward kernels take 8 x 8 x 8 bricks for 16-channel rows too (two 8-wave workgroups per CU, 65 KB of LDS each): 1.95 instead of 2.34 halo
// positions per voxel; measured against 4 x 8 x 8 on 4 waves x 3 workgroups: 9 -> 9 forward 434 -> 399 us, data gradient 497 -> 471, 9 -> 36 841 -> 765.
template <int G> struct StencilBrickF { static constexpr int TZV = 8, NTHR = TZV * 64; };
template <int G, int NT, typename AT, bool VEC>
__global__ __launch_bounds__(StencilBrickF<G>::NTHR, 2) void stencil3_fwd_kernel(const StencilArgsT<AT> p) {
  constexpr int C = 16 * G, KTOT = 27 * C, KPAD = (KTOT + 31) / 32 * 32, NSTEP = KPAD / 32, HXP = StencilPitch<G>::HXP;
  constexpr int TZV = StencilBrickF<G>::TZV, NTHR = StencilBrickF<G>::NTHR, NW = NTHR / 64;
  __shared__ __attribute__((aligned(16))) char xc[(TZV + 2) * HY * HXP * C * 2];   // halo brick
  __shared__ __attribute__((aligned(16))) __bf16 Ws[NT * 16 * KPAD];
  __shared__ float red[16 * 16 * 2];
  __bf16* Xs = reinterpret_cast<__bf16*>(xc);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  // weights -> LDS once per workgroup (16-byte vectors; rows padded with zeros to a multiple of 32 k)
  for (int i = tid; i < NT * 16 * KPAD / 8; i += NTHR) {
    const int n = (i * 8) / KPAD, k = (i * 8) - n * KPAD;
    bf16x8 v = VecN<__bf16, 8>::zero();
    if (k < KTOT) v = *reinterpret_cast<const bf16x8*>(p.w + (size_t)n * KTOT + k);
    *reinterpret_cast<bf16x8*>(Ws + n * KPAD + k) = v;
  }
  const int cs4 = (p.cout + 3) & ~3;
  // this lane's output columns: n = nt*16 + lg*4 + j (the accumulators are transposed blocks, see the MFMA loop)
  float bias4[NT][4], st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int n = nt * 16 + lg * 4 + j; bias4[nt][j] = (p.bias && n < p.cout) ? p.bias[n] : 0.f; }

  HaloRegs<G, AT, TZV, NTHR> hr;
  hr.init(p.ldx, p.x_plane, p.cin_load, p.H, p.W, tid, HXP);
  int tile = xcd_first_tile();
  TileId t = tile_of<TZV>(tile < p.ntiles ? tile : 0, p.D, p.H, p.W);
  if (tile < p.ntiles) { hr.load(p.x, p.ldx, t, p.D, p.H, p.W); hr.store(Xs, tid); }
  __syncthreads();
  const int yy = lr >> 3, xx = lr & 7;
  // per lane and MFMA step: the byte offset of the step's (tap, channel chunk) inside the halo image - k = 32 s + 8 lg = tap * C + c
  int toff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int kb = s * 32 + lg * 8;
    int tap = kb / C; const int c = kb - tap * C;
    if (tap > 26) tap = 26;                         // padded k: the weights are zero there, any valid address will do
    const int dz = tap / 9, dy = (tap - dz * 9) / 3, dx = tap - dz * 9 - dy * 3;
    toff[s] = (((dz * HY + dy) * HXP + dx) * C + c) * 2;
  }
  const char* xrow = xc + ((wave * HY + yy) * HXP + xx) * C * 2;                      // this wave's z-slice, M-tile 0
  const char* wrow = reinterpret_cast<const char*>(Ws) + (lr * KPAD + lg * 8) * 2;     // weight row lr, this lane's 8 k of step 0
  // this lane's output columns n0 + j (block nt adds 16 nt), which of them exist, and where its rows start inside a position row
  const int n0 = lg * 4;
  bool colreal[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) colreal[nt][j] = nt * 16 + n0 + j < p.cout;
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    TileId tn = t;
    if (next < p.ntiles) { tn = tile_of<TZV>(next, p.D, p.H, p.W); hr.load(p.x, p.ldx, tn, p.D, p.H, p.W); }   // in flight during the MFMA loop

    // this wave: z-slice `wave`; M-tile mt = rows y = 2mt, 2mt+1; fragment row r = lane&15 -> (yy = r>>3, xx = r&7)
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A fragment of (M-tile mt, step s): halo row (wave, 2 mt + yy, xx) shifted by the lane's tap of step s, 8 channels from c:
    // byte address = abase + mt * (2 HXP C 2) + toff[s], toff per lane and step from the table set up before the brick loop
#ifndef SV_ST_PROBE_NOMMA
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      bf16x8 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(wrow + nt * 16 * KPAD * 2 + s * 64);
      const char* ap = xrow + toff[s];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + mt * (2 * HXP * C * 2));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nt], a, acc[mt][nt], 0, 0, 0);   // B first: transposed block
      }
    }
#endif
    // With the weight fragment as the MFMA's first operand the accumulator block is transposed: lane (lr, lg) holds voxel
    // row lr = (y = 2mt + (lr >> 3), x = lr & 7) of z-slice `wave`, columns nt*16 + lg*4 .. +3 - four consecutive channels,
    // stored straight from the registers as one vector (no LDS staging of the tile, no extra barriers).
    const size_t pos0 = (((size_t)t.img * p.D + t.z0 + wave) * p.H + t.y0 + yy) * p.W + t.x0 + xx;     // M-tile 0; M-tile mt is 2 mt W positions on
    if constexpr (VEC) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int nn = nt * 16 + n0;
        if (nn >= cs4) continue;
        AT* orow; const AT* rrow = nullptr;
        if (p.out_plane) { const int pl = nn / p.ldc; orow = p.out + (size_t)pl * p.out_plane + pos0 * p.ldc + (nn - pl * p.ldc); }
        else orow = p.out + pos0 * p.ldc + p.col_off + nn;
        if (p.residual) rrow = p.residual + pos0 * p.ldr + nn;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = colreal[nt][j] ? acc[mt][nt][j] + bias4[nt][j] : 0.f;
          if (NT == 1 && p.stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { st1[j] += v[j]; st2[j] += v[j] * v[j]; }
          }
          if (p.residual) {
            const float4 rv = ld4f(rrow + (size_t)mt * 2 * p.W * p.ldr);
            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
          }
#ifdef SV_ST_PROBE_NOSTORE
          if (v[0] == 1234.5f)
#endif
          st4f(orow + (size_t)mt * 2 * p.W * p.ldc, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const size_t pos = pos0 + (size_t)mt * 2 * p.W;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int nn = nt * 16 + n0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (nn + j < p.cout) {
              float r = acc[mt][nt][j] + bias4[nt][j];
              if (NT == 1 && p.stats) { st1[j] += r; st2[j] += r * r; }
              if (p.residual) r += ldf(p.residual + pos * p.ldr + nn + j);
              stf(p.out + pos * p.ldc + p.col_off + nn + j, r);
            }
          }
        }
      }
    }
    t = tn;
    __syncthreads();                                 // every wave is done reading the brick: the next one may land in LDS
    if (next < p.ntiles) hr.store(Xs, tid);
    __syncthreads();
  }
  if (NT == 1 && p.stats) {                          // per-channel sum / sum of squares of the stored values (without residual)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      { st1[j] += dpp_move<0xB1>(st1[j]); st1[j] += dpp_move<0x4E>(st1[j]); st1[j] += dpp_move<0x141>(st1[j]); st1[j] += dpp_move<0x140>(st1[j]);
        st2[j] += dpp_move<0xB1>(st2[j]); st2[j] += dpp_move<0x4E>(st2[j]); st2[j] += dpp_move<0x141>(st2[j]); st2[j] += dpp_move<0x140>(st2[j]); }
    }
    if (lr == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(wave * 16 + lg * 4 + j) * 2] = st1[j]; red[(wave * 16 + lg * 4 + j) * 2 + 1] = st2[j]; }
    }
    __syncthreads();
    if (tid < 16 && tid < p.cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += red[(w * 16 + tid) * 2]; b += red[(w * 16 + tid) * 2 + 1]; }
      double* st = p.stats + (size_t)(blockIdx.x % SV_BN_SLOTS) * 2 * p.cout;
      atomicAdd(st + tid, (double)a);
      atomicAdd(st + p.cout + tid, (double)b);
    }
  }
}

template <typename AT>
struct StencilWArgsT {
  const AT* x; int ldx; int cin_load;        // gathered operand (conv input), memory channels = 16*G (zero-padded)
  long long x_plane;                         // != 0: channel planes of ldx channels each (see HaloRegs::load)
  const AT* dy; int lddy; int cout_load;     // anchor operand (output gradient), <= 16 memory channels
  float* dw;                                 // native [cout][cin][27], accumulated with atomics (directly, or through `ws`)
  float* ws;                                 // optional [WG_SLOTS][cout*cin*27] zeroed slot images folded into dw by a second kernel
  float* dbias;                              // optional [cout]: += sum_vox dy[vox][co]
  int cout, cin;                             // real channel counts
  int c_stride, c_valid;                     // memory channel c -> real channel (c / c_stride) * c_valid + c % c_stride, valid if c % c_stride < c_valid
  int I, D, H, W, ntiles;
};

constexpr int WG_SLOTS = 16;   // workgroups spread their partial sums over this many images (1024 workgroups on one image
                               // serialise ~1000 atomics per weight; measured 450 us vs the ~30 us the contraction needs)
__global__ __launch_bounds__(256) void stencil_wgrad_fold_kernel(const float* __restrict__ ws, float* __restrict__ dw, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
#pragma unroll
  for (int sl = 0; sl < WG_SLOTS; ++sl) a += ws[(size_t)sl * n + i];
  dw[i] += a;
}

// Brick depth / wave count as in the forward kernel (StencilBrick): 48-channel rows take 8 x 8 x 8 bricks on 8 waves (one workgroup per CU either
// way: two waves per SIMD instead of one); the NW waves split the 27 taps.
template <int G, typename AT>
__global__ __launch_bounds__(StencilBrick<G>::NTHR, 2) void stencil3_wgrad_kernel(const StencilWArgsT<AT> p) {
  constexpr int C = 16 * G, TZV = StencilBrick<G>::TZV, NTHR = StencilBrick<G>::NTHR, NW = NTHR / 64, NTAPW = (27 + NW - 1) / NW;
  constexpr int HPOSV = (TZV + 2) * HY * HX, NVOXV = TZV * TY * TX;
  __shared__ __attribute__((aligned(16))) __bf16 Xs[HPOSV * C];
  __shared__ __attribute__((aligned(16))) __bf16 Ds[NVOXV * 16];
  __shared__ float bred[NTHR / 16 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4, q = lr >> 2, pp = lr & 3;
  // taps of this wave: t = wave + NW * i, i < ntap
  const int ntap = (27 - wave + NW - 1) / NW;
  f32x4 acc[NTAPW][G];
#pragma unroll
  for (int i = 0; i < NTAPW; ++i)
#pragma unroll
    for (int gg = 0; gg < G; ++gg) acc[i][gg] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;                                   // bias gradient: column tid&15, voxel group tid>>4

  HaloRegs<G, AT, TZV, NTHR> hr;
  hr.init(p.ldx, p.x_plane, p.cin_load, p.H, p.W, tid);
  typename V4<AT>::type dr[4];                        // dy brick: NVOXV voxels x 4 vectors = 4 per thread
  auto load_dy = [&](const TileId& t) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + NTHR * k;
      const int v = i >> 2, c4 = i & 3;
      const int xx = v % TX; const int t2 = v / TX; const int yy = t2 % TY; const int zz = t2 / TY;
      typename V4<AT>::type qv = V4<AT>::zero();
      if (c4 * 4 < p.cout_load)
        qv = V4<AT>::load(p.dy + ((((size_t)t.img * p.D + t.z0 + zz) * p.H + t.y0 + yy) * p.W + t.x0 + xx) * (size_t)p.lddy + c4 * 4);
      dr[k] = qv;
    }
  };
  auto store_dy = [&]() {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x4*>(Ds + (tid + NTHR * k) * 4) = to_bf16x4(dr[k]);   // [voxel][16] == linear i*4
  };

  int tile = xcd_first_tile();
  if (tile < p.ntiles) {
    const TileId t = tile_of<TZV>(tile, p.D, p.H, p.W);
    hr.load(p.x, p.ldx, t, p.D, p.H, p.W); load_dy(t);
    hr.store(Xs, tid); store_dy();
  }
  __syncthreads();
  for (; tile < p.ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < p.ntiles) {
      const TileId t = tile_of<TZV>(next, p.D, p.H, p.W);
      hr.load(p.x, p.ldx, t, p.D, p.H, p.W); load_dy(t);     // in flight during the MFMA loop
    }
    if (p.dbias) {
      const int c = tid & 15, vg = tid >> 4;
#pragma unroll
      for (int k = 0; k < 16; ++k) bsum += (float)Ds[(vg * 16 + k) * 16 + c];
    }
    // 2 TZV chunks of 32 voxels: chunk = (z, half); k = 8*g + j  <->  (y = 4*half + g, x = j)
#pragma unroll 1
    for (int ch = 0; ch < 2 * TZV; ++ch) {
      const int z = ch >> 1, yb = (ch & 1) * 4 + lg;
      const __bf16* asrc = Ds + (((z * TY + yb) * TX) + q) * 16 + pp * 4;       // rows = 4 consecutive x positions
      const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(asrc));
      const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(asrc + 4 * 16));
      const bf16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int i = 0; i < NTAPW; ++i) {
        if (i < ntap) {
          const int tap = wave + NW * i;
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
    __syncthreads();                                   // the brick is consumed
    if (next < p.ntiles) { hr.store(Xs, tid); store_dy(); }
    __syncthreads();
  }
  if (p.dbias) {
    bred[tid] = bsum;                                  // [voxel group][column]
    __syncthreads();
    if (tid < 16 && tid < p.cout) {
      float a = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < NTHR / 16; ++g2) a += bred[g2 * 16 + tid];
      atomicAdd(p.dbias + tid, a);
    }
  }
  // C element: row = lg*4 + j -> co, col = lr -> memory channel 16*gg + lr
  float* dst = p.ws ? p.ws + (size_t)(blockIdx.x % WG_SLOTS) * p.cout * p.cin * 27 : p.dw;
#pragma unroll
  for (int i = 0; i < NTAPW; ++i) {
    if (i < ntap) {
      const int tap = wave + NW * i;
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const int cm = gg * 16 + lr;
        const int grp = cm / p.c_stride, within = cm - grp * p.c_stride;
        const int ci = grp * p.c_valid + within;
        if (within < p.c_valid && ci < p.cin) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = lg * 4 + j;
            if (co < p.cout) atomicAdd(dst + ((size_t)co * p.cin + ci) * 27 + tap, acc[i][gg][j]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// ConvTranspose3d(k = 4, stride 2, pad 1) with <= 8 output channels on an LDS halo brick (the decoder's last up-sampling layer,
// models/decoder.py:37-40: 32 -> 8 channels, 16^3 -> 32^3).  The generic engine gathers 8 taps x 64 B from L2 for every 16-byte output row
// (8.6 GB of gathers for 0.27 GB of output at I = 512: 1.6 ms) and pads 8 output channels to a 16-wide tile on top.
// Here a workgroup owns a 4 x 4 x 8 brick of INPUT positions j (+ 1 halo): output voxel o = 2 j + a (a = parity per axis) receives the two taps
// d = 0, 1 per axis from input i = j - 1 + a + d with kernel index k = a ? 2 - 2 d : 3 - 2 d, i.e. every parity class is a 2 x 2 x 2-tap
// stencil on the same halo brick with its own 8 x 32 weight slices.  MFMA 16x16x32: rows = 16 positions of one class, k = one tap's 32
// channels (one 16-byte LDS read per lane), columns = output channels (weights as the first operand: a lane holds 4 consecutive
// channels of a voxel).  The 8 x 8 x 16 output brick is staged in LDS and leaves as whole 256-byte rows, with the BatchNorm statistics.
struct TConv4Args {
  const __bf16* x; const __bf16* w;   // x [I][D][H][W][32]; w packed forward [cout][64 taps][32] (tap = (kd*4 + kh)*4 + kw)
  const float* bias; __bf16* out; double* stats; int cout; int I, D, H, W, nbricks;
};
constexpr int T4_JZ = 4, T4_JY = 4, T4_JX = 8, T4_HZ = 6, T4_HY = 6, T4_HX = 10, T4_HPOS = T4_HZ * T4_HY * T4_HX;

__global__ __launch_bounds__(256, 2) void tconv4s2_fwd_kernel(const TConv4Args p) {
  __shared__ __attribute__((aligned(16))) __bf16 Xs[T4_HPOS * 32];            // halo brick [pos][32]
  __shared__ __attribute__((aligned(16))) __bf16 Ws[8 * 8 * 256];             // [class][co][tap d*32 + c]
  __shared__ __attribute__((aligned(16))) __bf16 Os[8 * 8 * 16 * 8];          // output brick [oz][oy][ox][8]
  __shared__ float red[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  // class weights -> LDS: Ws[cls][co][t*32 + c] = w[co][tap(cls, t)][c]
  for (int i = tid; i < 8 * 8 * 256 / 8; i += 256) {
    const int c8 = (i & 3) * 8, t = (i >> 2) & 7, co = (i >> 5) & 7, cls = i >> 8;
    const int az = cls >> 2, ay = (cls >> 1) & 1, ax = cls & 1, dz = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
    const int kz = az ? 2 - 2 * dz : 3 - 2 * dz, ky = ay ? 2 - 2 * dy : 3 - 2 * dy, kx = ax ? 2 - 2 * dx : 3 - 2 * dx;
    bf16x8 v = VecN<__bf16, 8>::zero();
    if (co < p.cout) v = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)co * 64 + (kz * 4 + ky) * 4 + kx) * 32 + c8);
    *reinterpret_cast<bf16x8*>(Ws + ((cls * 8 + co) * 256) + t * 32 + c8) = v;
  }
  float bias4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias4[j] = (p.bias && lg < 2 && lg * 4 + j < p.cout) ? p.bias[lg * 4 + j] : 0.f;
  double st1[8], st2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { st1[c] = 0.0; st2[c] = 0.0; }
  const int bz = p.D / T4_JZ, by = p.H / T4_JY, bx = p.W / T4_JX;
  __syncthreads();
  for (int brick = blockIdx.x; brick < p.nbricks; brick += gridDim.x) {
    int t = brick;
    const int x0 = (t % bx) * T4_JX; t /= bx;
    const int y0 = (t % by) * T4_JY; t /= by;
    const int z0 = (t % bz) * T4_JZ; const int img = t / bz;
    // ---- halo brick -> LDS (zero outside the grid): 360 positions x 4 chunks of 8 channels
    for (int i = tid; i < T4_HPOS * 4; i += 256) {
      const int h = i >> 2, c8 = (i & 3) * 8;
      const int hx = h % T4_HX; const int t2 = h / T4_HX; const int hy = t2 % T4_HY; const int hz = t2 / T4_HY;
      const int z = z0 - 1 + hz, y = y0 - 1 + hy, xx = x0 - 1 + hx;
      bf16x8 v = VecN<__bf16, 8>::zero();
      if ((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
        v = *reinterpret_cast<const bf16x8*>(p.x + ((((size_t)img * p.D + z) * p.H + y) * p.W + xx) * 32 + c8);
      *reinterpret_cast<bf16x8*>(Xs + h * 32 + c8) = v;
    }
    __syncthreads();
    // ---- this wave: parity classes 2 wave, 2 wave + 1; 8 groups of 16 positions (jz, 2 jy rows, 8 jx) each
    const int yy = lr >> 3, xx = lr & 7;
#pragma unroll 1
    for (int ci = 0; ci < 2; ++ci) {
      const int cls = 2 * wave + ci, az = cls >> 2, ay = (cls >> 1) & 1, ax = cls & 1;
      bf16x8 b[8];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        b[tt] = VecN<__bf16, 8>::zero();
        if (lr < 8) b[tt] = *reinterpret_cast<const bf16x8*>(Ws + (cls * 8 + lr) * 256 + tt * 32 + lg * 8);
      }
#pragma unroll 2
      for (int g = 0; g < 8; ++g) {
        const int jz = g >> 1, jy = (g & 1) * 2 + yy;
        const int base = ((jz + az) * T4_HY + (jy + ay)) * T4_HX + (xx + ax);
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          const int off = ((tt >> 2) * T4_HY + ((tt >> 1) & 1)) * T4_HX + (tt & 1);
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(Xs + (base + off) * 32 + lg * 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tt], a, acc, 0, 0, 0);   // weights first: lane (lr, lg) holds voxel lr, channels 4 lg .. + 3
        }
        if (lg < 2) {
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (__bf16)(acc[j] + bias4[j]);
          const int oz = 2 * jz + az, oy = 2 * jy + ay, ox = 2 * xx + ax;
          *reinterpret_cast<bf16x4*>(Os + ((oz * 8 + oy) * 16 + ox) * 8 + lg * 4) = o;
        }
      }
    }
    __syncthreads();
    // ---- output brick -> HBM: 1024 voxels x 16 bytes, rows of 16 x-voxels contiguous; statistics of what is stored
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = tid + 256 * k;
      const int ox = v & 15, oy = (v >> 4) & 7, oz = v >> 7;
      const bf16x8 o = *reinterpret_cast<const bf16x8*>(Os + v * 8);
      const size_t pos = (((size_t)img * (2 * p.D) + 2 * z0 + oz) * (2 * p.H) + 2 * y0 + oy) * (2 * p.W) + 2 * x0 + ox;
      *reinterpret_cast<bf16x8*>(p.out + pos * 8) = o;
      if (p.stats) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { const double f = (double)(float)o[c]; st1[c] += f; st2[c] += f * f; }
      }
    }
    __syncthreads();                                   // Xs / Os are rewritten by the next brick
  }
  if (p.stats) {   // per-channel sums: lanes -> waves -> one double atomic per channel and workgroup into one of SV_BN_SLOTS slot images
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = (float)st1[c], b2 = (float)st2[c];
      a = wave_sum(a); b2 = wave_sum(b2);
      if (lane == 0) { red[wave][c] = a; red[wave][8 + c] = b2; }
    }
    __syncthreads();
    if (tid < 16 && (tid & 7) < p.cout) {
      const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      double* st = p.stats + (size_t)(blockIdx.x % SV_BN_SLOTS) * 2 * p.cout;
      atomicAdd(st + (tid >> 3) * p.cout + (tid & 7), (double)v);
    }
  }
}

}  // namespace sv

using namespace sv;

// resident 256-thread workgroups per CU of a kernel (cached per instantiation; 1 if the runtime cannot tell)
static int resident_per_cu(const void* kernel, int threads = 256) {
  static std::mutex mu;
  static std::map<const void*, int> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(kernel);
  if (it != cache.end()) return it->second;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
  cache[kernel] = n;
  return n;
}

static int stencil_check(int I, int D, int H, int W) {
  SV_REQUIRE(I > 0 && D % 8 == 0 && H % TY == 0 && W % TX == 0, "stencil3: grid %dx%dx%d must be a multiple of the 8x8x8 brick", D, H, W);
  return SV_OK;
}

#define SV_STENCIL_LAUNCH(GG, NTT, VV)                                                                                     \
  do {                                                                                                                    \
    constexpr int nthr = StencilBrickF<GG>::NTHR;                                                                         \
    const int nt_ = I * (D / StencilBrickF<GG>::TZV) * (H / TY) * (W / TX);                                               \
    a.ntiles = nt_;                                                                                                       \
    const int resident = 256 * resident_per_cu((const void*)stencil3_fwd_kernel<GG, NTT, AT, VV>, nthr);                  \
    hipLaunchKernelGGL((stencil3_fwd_kernel<GG, NTT, AT, VV>), dim3(nt_ < resident ? nt_ : resident), dim3(nthr), 0, s, a); \
  } while (0)

extern "C" int sv_stencil3_fwd(const void* x, int ldx, int cin_load, int groups, const void* w_bf16, int ntiles16,
                               const float* bias, void* out, int ldc, int col_off, int cout, const void* residual, int ldr,
                               double* stats, int I, int D, int H, int W, long long x_plane_stride, long long out_plane_stride,
                               int act_dtype, void* stream) {
  SV_REQUIRE(x && w_bf16 && out, "stencil3_fwd: null argument");
  SV_REQUIRE(x_plane_stride >= 0 && out_plane_stride >= 0 && x_plane_stride % 4 == 0 && out_plane_stride % 4 == 0, "stencil3_fwd: bad plane strides");
  SV_REQUIRE(!out_plane_stride || (col_off == 0 && !residual && ldc % 4 == 0 && cout % ldc == 0),
             "stencil3_fwd: planar output needs col_off == 0, no residual and whole planes of ldc (multiple of 4) columns");
  SV_REQUIRE_ACT(act_dtype);
  if (int rc = stencil_check(I, D, H, W)) return rc;
    SV_REQUIRE(cin_load % 4 == 0 && cin_load <= 16 * groups && ldx % 4 == 0 && (x_plane_stride ? cin_load % ldx == 0 : ldx >= cin_load),
             "stencil3_fwd: bad input channels (cin_load=%d ldx=%d groups=%d)", cin_load, ldx, groups);
  SV_REQUIRE(cout > 0 && cout <= 16 * ntiles16 && (out_plane_stride || ldc >= col_off + cout), "stencil3_fwd: bad output window");
  SV_REQUIRE(!stats || ntiles16 == 1, "stencil3_fwd: statistics need a single 16-column tile");
  const uintptr_t amask = act_dtype == SV_BF16 ? 7 : 15;
  SV_REQUIRE(((uintptr_t)x & amask) == 0 && ((uintptr_t)w_bf16 & 15) == 0, "stencil3_fwd: operands must be aligned to 4 elements");
  SV_REQUIRE((ldc & 3) != 0 || (((uintptr_t)out | (uintptr_t)residual) & amask) == 0,
             "stencil3_fwd: out/residual must be aligned to 4 elements when ldc is a multiple of 4");
  const int ntiles = I * (D / TZ) * (H / TY) * (W / TX);
  hipStream_t s = (hipStream_t)stream;
  if (!((groups == 1 && ntiles16 == 1) || (groups == 3 && ntiles16 == 1) || (groups == 1 && ntiles16 == 3))) {
    set_error("stencil3_fwd: unsupported (groups=%d, ntiles16=%d)", groups, ntiles16);
    return SV_ERR_INVALID;
  }
  // persistent grid: exactly as many workgroups as stay resident (asked from the runtime per instantiation: registers and
  // LDS decide; a larger grid would queue workgroups behind the resident ones and unbalance the brick loop)
  SV_DISPATCH_ACT(act_dtype,
    const int cs4 = (cout + 3) & ~3;
    const bool vec = out_plane_stride != 0 || ((ldc & 3) == 0 && (col_off & 3) == 0 && col_off + cs4 <= ldc && (!residual || ((ldr & 3) == 0 && cs4 <= ldr)));
    StencilArgsT<AT> a{static_cast<const AT*>(x), ldx, cin_load, x_plane_stride, out_plane_stride, (const __bf16*)w_bf16, bias, static_cast<AT*>(out), ldc, col_off, cout,
                       static_cast<const AT*>(residual), ldr, stats, I, D, H, W, ntiles};
    if (groups == 1 && ntiles16 == 1) { if (vec) SV_STENCIL_LAUNCH(1, 1, true); else SV_STENCIL_LAUNCH(1, 1, false); }
    else if (groups == 3) { if (vec) SV_STENCIL_LAUNCH(3, 1, true); else SV_STENCIL_LAUNCH(3, 1, false); }
    else { if (vec) SV_STENCIL_LAUNCH(1, 3, true); else SV_STENCIL_LAUNCH(1, 3, false); });
  return check_launch("sv_stencil3_fwd");
}

extern "C" size_t sv_stencil3_wgrad_workspace_floats(int cout, int cin) { return (size_t)WG_SLOTS * cout * cin * 27; }

extern "C" int sv_stencil3_wgrad(const void* x, int ldx, int cin_load, int groups, const void* dy, int lddy, int cout_load,
                                 float* dw, float* dbias, float* workspace, int cout, int cin, int c_stride, int c_valid, int I, int D, int H,
                                 int W, long long x_plane_stride, int act_dtype, void* stream) {
  SV_REQUIRE(x && dy && dw, "stencil3_wgrad: null argument");
  SV_REQUIRE(x_plane_stride >= 0 && x_plane_stride % 4 == 0 && (!x_plane_stride || cin_load % ldx == 0), "stencil3_wgrad: bad plane stride");
  SV_REQUIRE_ACT(act_dtype);
  if (int rc = stencil_check(I, D, H, W)) return rc;
  SV_REQUIRE(cin_load % 4 == 0 && cin_load <= 16 * groups && ldx % 4 == 0 && cout_load % 4 == 0 && cout_load <= 16 && lddy % 4 == 0,
             "stencil3_wgrad: bad channel layout");
  SV_REQUIRE(cout > 0 && cout <= 16 && cin > 0 && c_stride > 0 && c_valid > 0, "stencil3_wgrad: bad channel counts");
  SV_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & (act_dtype == SV_BF16 ? 7 : 15)) == 0, "stencil3_wgrad: operands must be aligned to 4 elements");
  SV_REQUIRE(groups == 1 || groups == 3, "stencil3_wgrad: unsupported groups=%d", groups);
    const int ntiles = I * (D / (groups == 3 ? StencilBrick<3>::TZV : StencilBrick<1>::TZV)) * (H / TY) * (W / TX);
  hipStream_t s = (hipStream_t)stream;
  SV_DISPATCH_ACT(act_dtype,
    const int per_cu = groups == 1 ? resident_per_cu((const void*)stencil3_wgrad_kernel<1, AT>, StencilBrick<1>::NTHR)
                                   : resident_per_cu((const void*)stencil3_wgrad_kernel<3, AT>, StencilBrick<3>::NTHR);
    const int resident = 256 * per_cu;
    const int blocks = ntiles < resident ? ntiles : resident;
    StencilWArgsT<AT> a{static_cast<const AT*>(x), ldx, cin_load, x_plane_stride, static_cast<const AT*>(dy), lddy, cout_load, dw, workspace, dbias, cout, cin, c_stride, c_valid,
                        I, D, H, W, ntiles};
    if (groups == 1) hipLaunchKernelGGL((stencil3_wgrad_kernel<1, AT>), dim3(blocks), dim3(StencilBrick<1>::NTHR), 0, s, a);
    else hipLaunchKernelGGL((stencil3_wgrad_kernel<3, AT>), dim3(blocks), dim3(StencilBrick<3>::NTHR), 0, s, a););
  if (workspace) hipLaunchKernelGGL(stencil_wgrad_fold_kernel, dim3(cdiv(cout * cin * 27, 256)), dim3(256), 0, s, workspace, dw, cout * cin * 27);
  return check_launch("sv_stencil3_wgrad");
}

extern "C" int sv_tconv4s2_fwd(const void* x, const void* w_packed, const float* bias, void* out, double* stats, int I, int D, int H, int W,
                               int cin, int cout, void* stream) {
  SV_REQUIRE(x && w_packed && out && I > 0, "tconv4s2_fwd: null argument");
  SV_REQUIRE(cin == 32 && cout > 0 && cout <= 8, "tconv4s2_fwd: built for 32 input and <= 8 output channels (got %d -> %d)", cin, cout);
  SV_REQUIRE(D % T4_JZ == 0 && H % T4_JY == 0 && W % T4_JX == 0, "tconv4s2_fwd: input grid %dx%dx%d must be a multiple of the 4x4x8 brick", D, H, W);
  SV_REQUIRE((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)out) & 15) == 0, "tconv4s2_fwd: operands must be 16-byte aligned");
  TConv4Args a{static_cast<const __bf16*>(x), static_cast<const __bf16*>(w_packed), bias, static_cast<__bf16*>(out), stats, cout, I, D, H, W,
               I * (D / T4_JZ) * (H / T4_JY) * (W / T4_JX)};
  const int blocks = a.nbricks < 512 ? a.nbricks : 512;
  hipLaunchKernelGGL(tconv4s2_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("sv_tconv4s2_fwd");
}
