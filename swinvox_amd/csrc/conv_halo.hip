// Halo-tile convolution for the two widest-grid, short-K convolutions of the ResNet trunk (reference models/encoder.py:22-23, torchvision
// resnet50 behind it):
//   kind 0: 3 x 3 / stride 1 / padding 1, 64 -> 64 channels on the 56 x 56 grid (conv2 of the three layer1 bottlenecks) - forward, and with
//           flip = 1 the data gradient (the same convolution with mirrored taps on the data-gradient weight pack);
//   kind 1: 4 x 4 / stride 1 / pads (2, 1), 16 -> 64 channels on the 112 x 112 space-to-depth image (the 7 x 7 / stride-2 stem).
// The gather engine (igemm_kernel) fetches the activation tile once PER TAP: 9 x 16 KB (+ 9 x 8 KB of weights) from L2 into LDS for 128 x 64
// outputs, and the L2 -> LDS fill (60-80 GB/s per CU) is what bounds it: 338 TFLOP/s on kind 0, 230 on kind 1.  Here
//   * the whole weight tensor (9 x 64 x 64 bf16 = 72 KB / 16 x 64 x 16 = 32 KB) sits in LDS for the life of the workgroup,
//   * a tile of 8 x 32 output positions reads its (8 + KH - 1) x (32 + KW - 1) input patch ONCE (43.5 KB: 1.33 x the tile, instead of 9 x),
//     prefetched into registers while the previous tile is contracted (one buffer: the second would not fit beside the weights),
//   * all taps run from that patch: per tap and 16 channels a wave reads 2 weight fragments and 2 patch fragments (ds_read_b128, XOR-swizzled:
//     conflict-free) for 4 v_mfma_f32_32x32x16_bf16 - weights are the A operand, so a lane ends up with CHANNELS of one position,
//   * the 64 x 64 result of a wave turns through a private 8 KB LDS patch into 16-byte lanes: one store instruction = 8 whole 128-byte rows,
//   * BatchNorm statistics stay in registers across all tiles of the workgroup: one reduction and 128 double atomics per wave per LAUNCH.
// Tiles are drawn at run time from per-XCD counters (a static share stalls the launch when another stream's kernel holds a CU: DESIGN 4b).
// What bounds it then: the output stream (kind 0: 205 MB in + 205 MB out per launch at 512 images).
#include "common.h"
#include "conv_halo.h"
#include <stdlib.h>
#include <atomic>

namespace sv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte slot s of row r of an LDS image holds source chunk s ^ hc_swz(r): the four 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}
// ...: rows r0 + those) then touch 16 different bank slots.  128-byte rows: (r / 2) mod 8 (rows r, r + 1 differ in the 128-byte half);
// 32-byte rows: bit 3 of r (rows r, r + 8 share a slot pair, the XOR sends them to different halves of it)
template <int CI> __device__ __forceinline__ int hc_swz(int r) { return CI == 64 ? ((r >> 1) & 7) : ((r >> 3) & 1); }

template <int CI, int KH, int KW, int PLO>
__global__ __launch_bounds__(256, 1) void conv_halo_kernel(const HaloConvArgs p, int tiles_h, int tiles_w, int ntiles, int* __restrict__ ctr) {
  constexpr int TH = 8, TW = 32, T = KH * KW, PH = TH + KH - 1, PW = TW + KW - 1, NPOS = PH * PW;
  constexpr int CH = CI / 8, ROWB = CI * 2, KS = CI / 16;
  constexpr int NCH = NPOS * CH, NLD = (NCH + 255) / 256;
  constexpr int W_BYTES = T * 64 * ROWB, PATCH_BYTES = (NPOS * ROWB + 127) / 128 * 128, STAGE_BYTES = 4 * 8192;
  constexpr int WCH = T * 64 * CH;
  __shared__ __attribute__((aligned(1024))) char smem[W_BYTES + PATCH_BYTES + STAGE_BYTES + 16];
  const __bf16* __restrict__ X = static_cast<const __bf16*>(p.x);
  const __bf16* __restrict__ Wt = static_cast<const __bf16*>(p.w);
  __bf16* __restrict__ Y = static_cast<__bf16*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kg = lane >> 5;
  const int H = p.H, Wd = p.W;
  char* Wl = smem;
  char* Pl = smem + W_BYTES;
  char* Sl = smem + W_BYTES + PATCH_BYTES + wave * 8192;
  int* mbox = reinterpret_cast<int*>(smem + W_BYTES + PATCH_BYTES + STAGE_BYTES);

  // ---- tile scheduler (gemm_wide_kernel's: 8 contiguous chunks of the tile space, one per XCD; ctr[8] counts finished workgroups)
  const int xcd = blockIdx.x & 7, cq = ntiles >> 3, cr = ntiles & 7;
  const int cbase = xcd * cq + (xcd < cr ? xcd : cr), csize = cq + (xcd < cr ? 1 : 0);
  if (tid == 0) {
    const int v0 = atomicAdd(ctr + xcd, 1), v1 = atomicAdd(ctr + xcd, 1);
    mbox[0] = v0 < csize ? cbase + v0 : -1;
    mbox[1] = v1 < csize ? cbase + v1 : -1;
  }
  // ---- weights -> LDS, once: pack row r = output channel, [tap][CI]; LDS image [tap slot][r][CI] with swizzled 16-byte slots
  for (int id = tid; id < WCH; id += 256) {
    const int c = id % CH, rt = id / CH, t = rt % T, r = rt / T;
    const int ts = p.flip ? T - 1 - t : t;
    *reinterpret_cast<bf16x8*>(Wl + (ts * 64 + r) * ROWB + ((c ^ hc_swz<CI>(r)) << 4)) = *reinterpret_cast<const bf16x8*>(Wt + (size_t)id * 8);
  }
  __syncthreads();
  int cur = __builtin_amdgcn_readfirstlane(mbox[0]), nxt = __builtin_amdgcn_readfirstlane(mbox[1]);
  auto finish = [&]() {
    if (atomicAdd(ctr + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) ctr[i] = 0;
    }
  };
  if (cur < 0) { if (tid == 0) finish(); return; }

  const int tpi = tiles_h * tiles_w;
  auto origin = [&](int t, int& n, int& h0, int& w0) {
    n = t / tpi;
    const int r = t - n * tpi, th = r / tiles_w;
    h0 = th * TH; w0 = (r - th * tiles_w) * TW;
  };
  bf16x8 pre[NLD];
  auto load_patch = [&](int t) {         // chunk c of the patch image = 16-byte slot c % CH of patch position c / CH: lane-linear in LDS
    int n, h0, w0;
    origin(t, n, h0, w0);
    const __bf16* img = X + (size_t)n * H * Wd * CI;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int c = tid + 256 * j, pp = c / CH, s = c % CH, pr = pp / PW, pc = pp - pr * PW;
      const int ih = h0 - PLO + pr, iw = w0 - PLO + pc;
      const bool ok = (NCH % 256 == 0 || c < NCH) && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)Wd;
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (__bf16)0.f;
      if (ok) v = *reinterpret_cast<const bf16x8*>(img + ((size_t)ih * Wd + iw) * CI + ((s ^ hc_swz<CI>(pp)) << 3));
      pre[j] = v;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int c = tid + 256 * j;
      if (NCH % 256 == 0 || c < NCH) *reinterpret_cast<bf16x8*>(Pl + c * 16) = pre[j];
    }
  };
  load_patch(cur);
  store_patch();
  __syncthreads();

  float s1[2][16], s2[2][16];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int j = 0; j < 16; ++j) { s1[mb][j] = 0.f; s2[mb][j] = 0.f; }
  const int swa = hc_swz<CI>(ln);        // rows ln and 32 + ln of a tap's weight image share it (both forms of hc_swz have period <= 32)

  while (true) {
    if (nxt >= 0) load_patch(nxt);
    int drawn = 0;
    if (tid == 0 && nxt >= 0) drawn = atomicAdd(ctr + xcd, 1);

    // ---- contraction: wave w owns tile rows 2 w, 2 w + 1 (two blocks of 32 positions) x 64 channels (two blocks of 32)
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mb][nb][j] = 0.f;
    // fragments of step (tap, 16-channel slice) + 1 are read while the 4 MFMAs of the step run (one wave per SIMD: nobody else hides the LDS)
    bf16x8 fa[2][2], fb[2][2];
    auto frags = [&](int step, bf16x8* a, bf16x8* b) {
      const int t = step / KS, ks = step - t * KS, kh = t / KW, kw = t - kh * KW, ch = 2 * ks + kg;
      a[0] = *reinterpret_cast<const bf16x8*>(Wl + (t * 64 + ln) * ROWB + ((ch ^ swa) << 4));
      a[1] = *reinterpret_cast<const bf16x8*>(Wl + (t * 64 + 32 + ln) * ROWB + ((ch ^ swa) << 4));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int pp = (2 * wave + nb + kh) * PW + ln + kw;
        b[nb] = *reinterpret_cast<const bf16x8*>(Pl + pp * ROWB + ((ch ^ hc_swz<CI>(pp)) << 4));
      }
    };
    frags(0, fa[0], fb[0]);
#pragma unroll
    for (int step = 0; step < T * KS; ++step) {
      const int c = step & 1;
      if (step + 1 < T * KS) frags(step + 1, fa[c ^ 1], fb[c ^ 1]);
      __builtin_amdgcn_sched_barrier(0);               // hipcc otherwise sinks the reads to their first use: the MFMA would wait out the LDS
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                     // every wave is done reading the patch
    if (nxt >= 0) store_patch();
    if (tid == 0) mbox[0] = (nxt >= 0 && drawn < csize) ? cbase + drawn : -1;

    // ---- epilogue of `cur`: lane (ln, kg) holds, per block pair (mb, nb), position 32 nb + ln and channels 32 mb + 8 (j / 4) + 4 kg + j % 4
    int n, h0, w0;
    origin(cur, n, h0, w0);
    const int hr = h0 + 2 * wave;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int q = nb * 32 + ln;
      const float valid = (hr + nb < H && w0 + ln < Wd) ? 1.f : 0.f;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = (__bf16)acc[mb][nb][jg * 4 + i];
            if (p.stats) {               // of what is stored (sv_epilogue.stats), as the gather engine's epilogue counts
              const float v = (float)o[i], m = v * valid;
              s1[mb][jg * 4 + i] += m; s2[mb][jg * 4 + i] += m * v;
            }
          }
          *reinterpret_cast<bf16x4*>(Sl + q * 128 + (((mb * 4 + jg) ^ (q & 7)) << 4) + kg * 8) = o;
        }
      }
    }
    __bf16* yimg = Y + (size_t)n * H * Wd * 64;
#pragma unroll
    for (int itr = 0; itr < 8; ++itr) {  // 8 lanes = one 128-byte row; one instruction = 8 consecutive positions = 1 KB contiguous
      const int q = itr * 8 + (lane >> 3), c8 = lane & 7;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(Sl + q * 128 + ((c8 ^ (q & 7)) << 4));
      const int row = hr + (q >> 5), col = w0 + (q & 31);
      if (row < H && col < Wd) *reinterpret_cast<bf16x8*>(yimg + ((size_t)row * Wd + col) * 64 + c8 * 8) = v;
    }
    __syncthreads();                     // the next patch and the mailbox are complete
    if (nxt < 0) break;
    cur = nxt;
    nxt = __builtin_amdgcn_readfirstlane(mbox[0]);      // thread 0 posts again behind the next barrier, which needs everyone past this read
  }

  if (p.stats) {    // lanes that share kg (32 positions each) -> lanes 0 and 32 -> double atomics into a slot image
    double* st = p.stats + (size_t)((blockIdx.x * 4 + wave) % SV_BN_SLOTS) * 128;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a = s1[mb][j], b = s2[mb][j];
        a = lane_step_add<1>(a); a = lane_step_add<2>(a); a = lane_step_add<4>(a); a = lane_step_add<8>(a); a = lane_step_add<16>(a);
        b = lane_step_add<1>(b); b = lane_step_add<2>(b); b = lane_step_add<4>(b); b = lane_step_add<8>(b); b = lane_step_add<16>(b);
        if (ln == 0) {
          const int co = mb * 32 + 8 * (j >> 2) + 4 * kg + (j & 3);
          atomicAdd(st + co, (double)a);
          atomicAdd(st + 64 + co, (double)b);
        }
      }
    }
  }
  if (tid == 0) finish();
}

// 0: off (every call stays on the gather engine), 1: the shapes above from 256 tiles on, 2: those shapes at any size (tests).  SV_CONV_HALO in
// the environment sets the initial mode, sv_set_conv_halo() changes it (process-wide configuration, like the other A/B switches)
static std::atomic<int>& halo_mode() {
  static std::atomic<int> m{[] { const char* v = getenv("SV_CONV_HALO"); return v ? atoi(v) : 1; }()};
  return m;
}
static std::atomic<long long> halo_launches{0};
bool conv_halo_enabled() { return halo_mode().load(std::memory_order_relaxed) != 0; }

int conv_halo_launch(const HaloConvArgs& a, int kind, hipStream_t stream) {
  if (!conv_halo_enabled()) return 0;
  const int tiles_h = cdiv(a.H, 8), tiles_w = cdiv(a.W, 32);
  const long long nt = (long long)a.N * tiles_h * tiles_w;
  const int mode = halo_mode().load(std::memory_order_relaxed);
  if ((mode == 1 && nt < 256) || nt >= (1ll << 30)) return 0;   // fewer tiles than CUs: 72 KB of weights per workgroup for one tile each
  int* ctr = tile_draw_counters();
  if (!ctr) return 0;
  const int ntiles = (int)nt;
  if (kind == 0)
    hipLaunchKernelGGL((conv_halo_kernel<64, 3, 3, 1>), dim3(256), dim3(256), 0, stream, a, tiles_h, tiles_w, ntiles, ctr);
  else
    hipLaunchKernelGGL((conv_halo_kernel<16, 4, 4, 2>), dim3(256), dim3(256), 0, stream, a, tiles_h, tiles_w, ntiles, ctr);
  halo_launches.fetch_add(1, std::memory_order_relaxed);
  return 1;
}

}  // namespace sv

extern "C" int sv_set_conv_halo(int mode) {
  SV_REQUIRE(mode >= 0 && mode <= 2, "sv_set_conv_halo: mode %d (0 off, 1 auto, 2 always)", mode);
  sv::halo_mode().store(mode, std::memory_order_relaxed);
  return SV_OK;
}
extern "C" long long sv_conv_halo_launches(void) { return sv::halo_launches.load(std::memory_order_relaxed); }
extern "C" int sv_conv_halo_mode(void) { return sv::halo_mode().load(std::memory_order_relaxed); }
