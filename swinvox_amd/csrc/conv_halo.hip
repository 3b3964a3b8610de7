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
//   * the 64 x 64 result of a wave turns, one tile row at a time, through a private 4 KB LDS image into 16-byte lanes: one store instruction =
//     8 whole 128-byte rows,
//   * BatchNorm statistics: per tile a lane sums two channels of one tile row from the staging image (fixed order), then adds into four
//     double registers that live across all tiles of the workgroup: 256 double atomics per wave per LAUNCH.
// Tiles are drawn at run time from per-XCD counters (a static share stalls the launch when another stream's kernel holds a CU: DESIGN 4b).
// (Going on with the next XCD's chunk once the own one is exhausted was built and measured: the XCDs finish up to 14 % apart, yet the launch
// got 10 % SLOWER on the same box, 156 -> 171 us and 359 -> 396 us - the stolen tiles' halo rows are in another L2 and the draw loop sits on
// the path between the two barriers.  Not kept.)
// What bounds it then: the output stream (kind 0: 205 MB in + 205 MB out per launch at 512 images).
#include "common.h"
#include "conv_halo.h"
#include <stdlib.h>
#include <atomic>

namespace sv {

#ifdef SV_HC_PROFILE   // cycles of workgroup 0 / thread 0 per phase: [0] preamble, [1] patch issue, [2] contraction, [3] patch store, [4] epilogue, [5] tiles
__device__ long long hc_prof[8];
__device__ long long hc_wg[512][4];   // per workgroup: wall-clock (100 MHz) at start / loop start / end, tiles
#define HC_T(v) const long long v = clock64()
#define HC_ADD(i, a, b) if (blockIdx.x == 0 && tid == 0) hc_prof[i] += (b) - (a)
#else
#define HC_T(v)
#define HC_ADD(i, a, b)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// weight image: 16-byte slot s of row r holds source chunk s ^ hc_swz(r): the four 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}
// ...: rows r0 + those) then touch 16 different bank slots.  128-byte rows: (r / 2) mod 8 (rows r, r + 1 differ in the 128-byte half);
// 32-byte rows: bit 3 of r (rows r, r + 8 share a slot pair, the XOR sends them to different halves of it)
template <int CI> __device__ __forceinline__ int hc_swz(int r) { return CI == 64 ? ((r >> 1) & 7) : ((r >> 3) & 1); }

template <int CI, int KH, int KW, int PLO, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_halo_kernel(const HaloConvArgs p, int tiles_h, int tiles_w, int ntiles, int* __restrict__ ctr) {
  constexpr int TH = 8, TW = 32, T = KH * KW, PH = TH + KH - 1, PW = TW + KW - 1, NPOS = PH * PW;
  constexpr int CH = CI / 8, ROWB = CI * 2, KS = CI / 16;
  constexpr int NCH = NPOS * CH, NLD = (NCH + 255) / 256;
  // patch rows are PADDED by 16 bytes instead of swizzled (pitch 144 / 48 bytes = 9 / 3 sixteen-byte slots, odd: 16 positions of a read
  // group land on 16 different slots), so a fragment address is lane base + a compile-time offset per (tap, slice): no address arithmetic
  // between the MFMAs.  The weight image cannot afford the padding (82,944 + 48,960 + 32,768 B > 160 KB) and keeps the XOR, whose four
  // variants per lane are loop constants.
  constexpr int PROWB = ROWB + 16;
  constexpr int W_BYTES = T * 64 * ROWB, PATCH_BYTES = (NPOS * PROWB + 127) / 128 * 128, STAGE_BYTES = 4 * 4096;
  constexpr int WCH = T * 64 * CH;
  __shared__ __attribute__((aligned(1024))) char smem[W_BYTES + PATCH_BYTES + STAGE_BYTES + 16];
  const __bf16* __restrict__ X = static_cast<const __bf16*>(p.x);
  const __bf16* __restrict__ Wt = static_cast<const __bf16*>(p.w);
  __bf16* __restrict__ Y = static_cast<__bf16*>(p.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kg = lane >> 5;
  const int H = p.H, Wd = p.W;
  HC_T(t_begin);
#ifdef SV_HC_PROFILE
  const long long wall_begin = wall_clock64();
#endif
  char* Wl = smem;
  char* Pl = smem + W_BYTES;
  char* Sl = smem + W_BYTES + PATCH_BYTES + wave * 4096;
  int* mbox = reinterpret_cast<int*>(smem + W_BYTES + PATCH_BYTES + STAGE_BYTES);

  // ---- tile scheduler (gemm_wide_kernel's: 8 contiguous chunks of the tile space, one per XCD; ctr[8] counts finished workgroups)
  const int xcd = blockIdx.x & 7, cq = ntiles >> 3, cr = ntiles & 7;
  const int cbase = xcd * cq + (xcd < cr ? xcd : cr), csize = cq + (xcd < cr ? 1 : 0);
  if (tid == 0) {
    const int v0 = atomicAdd(ctr + xcd, 1), v1 = atomicAdd(ctr + xcd, 1);
    mbox[0] = v0 < csize ? cbase + v0 : -1;
    mbox[1] = v1 < csize ? cbase + v1 : -1;
  }
  // ---- weights -> LDS, once: pack row r = output channel, [tap][CI]; LDS image [tap slot][r][CI] with swizzled 16-byte slots.  All loads
  // of a thread are in flight together (a load - wait - store loop is 18 global round trips in a row)
  {
    constexpr int NWL = (WCH + 255) / 256;
    u32x4 wr[NWL];
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
      const int id = tid + 256 * j;
      if (WCH % 256 == 0 || id < WCH) wr[j] = *reinterpret_cast<const u32x4*>(Wt + (size_t)id * 8);
    }
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
      const int id = tid + 256 * j;
      const int c = id % CH, rt = id / CH, t = rt % T, r = rt / T;
      const int ts = p.flip ? T - 1 - t : t;
      if (WCH % 256 == 0 || id < WCH) *reinterpret_cast<u32x4*>(Wl + (ts * 64 + r) * ROWB + ((c ^ hc_swz<CI>(r)) << 4)) = wr[j];
    }
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
  // Buffer descriptors over the whole tensors (< 4 GB, checked by the launcher): a load past the end returns zeros - the padding - and a
  // store past the end is dropped, so neither needs a branch.  With exec-masked branches around them hipcc cannot count the memory
  // operations in flight and falls back to s_waitcnt vmcnt(0): at the loop head that waited out the previous tile's stores.
  const unsigned xbytes = (unsigned)p.N * H * Wd * ROWB, ybytes = (unsigned)p.N * H * Wd * 128;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(X), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(Y, 0, ybytes, 0x00020000);
  u32x4 pre[NLD];
  // per-thread constants of its NLD chunks: row / column of the patch position and the byte offset of the chunk relative to the patch origin
  int cpr[NLD], cpc[NLD];
  unsigned crel[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int c = tid + 256 * j, pp = c / CH, s = c % CH;
    cpr[j] = pp / PW - PLO; cpc[j] = pp % PW - PLO;
    crel[j] = (unsigned)((cpr[j] * Wd + cpc[j]) * ROWB + (s << 4));
  }
  auto load_patch = [&](int t) {         // chunk c of the patch = 16-byte slot c % CH of patch position c / CH
    int n, h0, w0;
    origin(t, n, h0, w0);
    const unsigned base = (unsigned)((n * H + h0) * Wd + w0) * ROWB;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const bool ok = (unsigned)(h0 + cpr[j]) < (unsigned)H && (unsigned)(w0 + cpc[j]) < (unsigned)Wd;      // c >= NCH: loaded, never stored
      // bit 31 sends a padding position past the end of the buffer (tensors < 2 GB) - arithmetic, not a select: hipcc turns `ok ? offset :
      // past_the_end` into a branch around the offset arithmetic
#ifdef SV_HC_PROBE_L2      // timing probe: every patch comes from one 1 MB window (L2 hits) - what the HBM latency of the real loads costs
      pre[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, ((base + crel[j]) & 0xffff0u) | ((unsigned)!ok << 31), 0, 0);
#else
      pre[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (base + crel[j]) | ((unsigned)!ok << 31), 0, 0);
#endif
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int c = tid + 256 * j;
      if (NCH % 256 == 0 || c < NCH) *reinterpret_cast<u32x4*>(Pl + (c / CH) * PROWB + (c % CH) * 16) = pre[j];
    }
  };
  load_patch(cur);
  store_patch();
  __syncthreads();
  HC_T(t_loop);
  HC_ADD(0, t_begin, t_loop);
#ifdef SV_HC_PROFILE
  const long long wall_loop = wall_clock64();
  int hc_tiles = 0;
#endif

  // BatchNorm statistics: lane (h = kg, cp = ln) owns channels 2 cp, 2 cp + 1 over the positions of tile row 2 wave + h.  Per tile it sums
  // the STORED bf16 values of its row from the staging image in a fixed order (fp32, <= 32 terms) and adds that to four double
  // accumulators: which workgroup computes which tiles changes from run to run (run-time draw), the sums do not - beyond the order of
  // double additions.  (Per-lane fp32 sums over all tiles of a workgroup, the first form, moved the batch mean by 1e-4 of itself from run to
  // run: harmless beside bf16, but it made two identical training steps differ by 1e-2 in the stem's BatchNorm gradient.)
  double sd[4] = {0.0, 0.0, 0.0, 0.0};   // sum / sum of squares of channel 2 cp, then of channel 2 cp + 1
  const int swa = hc_swz<CI>(ln);        // rows ln and 32 + ln of a tap's weight image share it (both forms of hc_swz have period <= 32)

  while (true) {
    int drawn;                           // thread 0 only, read behind the contraction (no initial value: a merge of two values at the end of
    if (tid == 0) drawn = atomicAdd(ctr + xcd, nxt >= 0 ? 1 : 0);   // this branch would be a copy, i.e. a wait for the return right here)

    HC_T(t0);
    // The next patch goes out before the first MFMA (unconditionally: the last tile re-reads its own).  Spread over the MFMA loop the loads
    // came out worse: hipcc's scheduler parked half of them behind the last MFMA, and their address arithmetic split the loop into blocks.
    load_patch(nxt >= 0 ? nxt : cur);
    __builtin_amdgcn_sched_barrier(0);
    HC_T(t1);
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
    const char* pa = Wl + ln * ROWB;
    const char* pq = Pl + (2 * wave * PW + ln) * PROWB + kg * 16;
    int aoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) aoff[ks] = ((2 * ks + kg) ^ swa) << 4;
    auto frags = [&](int step, bf16x8* a, bf16x8* b) {
      const int t = step / KS, ks = step - t * KS, kh = t / KW, kw = t - kh * KW;
      a[0] = *reinterpret_cast<const bf16x8*>(pa + t * 64 * ROWB + aoff[ks]);
      b[0] = *reinterpret_cast<const bf16x8*>(pq + (kh * PW + kw) * PROWB + ks * 32);
      a[1] = *reinterpret_cast<const bf16x8*>(pa + (t * 64 + 32) * ROWB + aoff[ks]);
      b[1] = *reinterpret_cast<const bf16x8*>(pq + ((kh + 1) * PW + kw) * PROWB + ks * 32);
    };
    frags(0, fa[0], fb[0]);
#pragma unroll
    for (int step = 0; step < T * KS; ++step) {
      const int c = step & 1;
      if (step + 1 < T * KS) frags(step + 1, fa[c ^ 1], fb[c ^ 1]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
      // hipcc otherwise sinks the reads to their first use (the MFMA would wait out the LDS): the four reads of the next step go out around
      // the first MFMA of this one, three MFMAs (96 cycles) before the first of them is needed
      if (step + 1 < T * KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (step + 1 < T * KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __syncthreads();                     // every wave is done reading the patch
    HC_T(t2);
    if (nxt >= 0) store_patch();
    if (tid == 0) mbox[0] = (nxt >= 0 && drawn < csize) ? cbase + drawn : -1;

    HC_T(t3);
    // ---- epilogue of `cur`: lane (ln, kg) holds, per block pair (mb, nb), position 32 nb + ln and channels 32 mb + 8 (j / 4) + 4 kg + j % 4
    int n, h0, w0;
    origin(cur, n, h0, w0);
    const int hr = h0 + 2 * wave;
    const int ncol = min(TW, Wd - w0);
    f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {     // one tile row (32 positions x 64 channels = 4 KB) at a time through the wave's staging image
      const int row = hr + nb;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (__bf16)acc[mb][nb][jg * 4 + i];
          *reinterpret_cast<bf16x4*>(Sl + ln * 128 + (((mb * 4 + jg) ^ (ln & 7)) << 4) + kg * 8) = o;
        }
      }
#pragma unroll
      for (int itr = 0; itr < 4; ++itr) {  // 8 lanes = one 128-byte row; one instruction = 8 consecutive positions = 1 KB contiguous
        const int q = itr * 8 + (lane >> 3), c8 = lane & 7, col = w0 + q;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Sl + q * 128 + ((c8 ^ (q & 7)) << 4));
        const unsigned off = ((unsigned)((n * H + row) * Wd + col) * 128 + c8 * 16) | ((unsigned)!(row < H && col < Wd) << 31);
#ifdef SV_HC_PROBE_NOSTORE  // timing probe: no output stream
        __builtin_amdgcn_raw_buffer_store_b128(v, yr, off | 0x80000000u, 0, 0);
#else
        __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
#endif
      }
      if (p.stats && row < H) {          // of what is stored (sv_epilogue.stats): lane (kg, cp = ln) takes positions 16 kg .. 16 kg + 15 of the row
        const char* srow = Sl + kg * 16 * 128 + (ln & 3) * 4;
        if (ncol == TW) {                // packed fp32 pairs: (channel 2 cp, channel 2 cp + 1) per instruction
#pragma unroll
          for (int i = 0; i < 16; ++i) { // position q = 16 kg + i: its 16-byte slots are XOR-swizzled by q & 7 = i & 7
            const unsigned u = *reinterpret_cast<const unsigned*>(srow + i * 128 + ((((ln >> 2) ^ i) & 7) << 4));
            const f32x2 v = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
            sa += v; sb += v * v;
          }
        } else {                         // a tile cut by the right image edge
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned u = *reinterpret_cast<const unsigned*>(srow + i * 128 + ((((ln >> 2) ^ i) & 7) << 4));
            const float m = 16 * kg + i < ncol ? 1.f : 0.f;
            const f32x2 v = {__uint_as_float(u << 16) * m, __uint_as_float(u & 0xffff0000u) * m};
            sa += v; sb += v * v;
          }
        }
      }
    }
    if (p.stats) { sd[0] += (double)sa[0]; sd[1] += (double)sb[0]; sd[2] += (double)sa[1]; sd[3] += (double)sb[1]; }
    __syncthreads();                     // the next patch and the mailbox are complete
    HC_T(t4);
#ifdef SV_HC_PROFILE
    ++hc_tiles;
#endif
    HC_ADD(2, t1, t2); HC_ADD(3, t2, t3); HC_ADD(4, t3, t4);
    if (nxt < 0) break;
    cur = nxt;
    nxt = __builtin_amdgcn_readfirstlane(mbox[0]);      // thread 0 posts again behind the next barrier, which needs everyone past this read
  }

  HC_T(t_post);
  if (p.stats) {    // 256 double atomics per wave into a slot image (full-wave instructions: 4 per wave)
    double* st = p.stats + (size_t)((blockIdx.x * 8 + wave * 2 + kg) % SV_BN_SLOTS) * 128;
    atomicAdd(st + 2 * ln, sd[0]);
    atomicAdd(st + 64 + 2 * ln, sd[1]);
    atomicAdd(st + 2 * ln + 1, sd[2]);
    atomicAdd(st + 64 + 2 * ln + 1, sd[3]);
  }
#ifdef SV_HC_PROFILE
  if (tid == 0) { hc_wg[blockIdx.x][0] = wall_begin; hc_wg[blockIdx.x][1] = wall_loop; hc_wg[blockIdx.x][2] = wall_clock64(); hc_wg[blockIdx.x][3] = hc_tiles; }
  if (blockIdx.x == 0 && tid == 0) { hc_prof[6] += wall_clock64() - wall_begin; hc_prof[7] += clock64() - t_begin; hc_prof[1] += clock64() - t_post; hc_prof[5] = t_post - t_loop; }   // 100 MHz ticks / shader cycles
#endif
  if (tid == 0) finish();
}

// 0: off (every call stays on the gather engine), 1: the shapes above from 256 tiles on, 2: those shapes at any size (tests).  SV_CONV_HALO in
// the environment sets the initial mode, sv_set_conv_halo() changes it (process-wide configuration, like the other A/B switches)
static std::atomic<int>& halo_mode() {
  static std::atomic<int> m{[] { const char* v = getenv("SV_CONV_HALO"); return v ? atoi(v) : 1; }()};
  return m;
}
// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3 x 3 / stride 1 / padding 1 convolutions with 64 k input and output channels (conv2 of the ResNet bottlenecks):
//   dW[co][tap][ci] = sum over positions of dy[pos][co] * x[pos + tap - 1][ci]
// wgrad_kernel treats it as a GEMM with 9 Ci gathered columns and re-reads the position tiles of x and dy once per 128 x 128 output tile:
// 18 times each at 256 channels (L2 -> LDS fill bound: 320-510 TFLOP/s).  Here a workgroup owns ONE 64 x 64 (co, ci) block for ALL nine taps -
// 9 x 16 accumulator registers per lane, wave w the 32 x 32 sub-block (w / 2, w % 2) - and walks a range of 8 x 32-position tiles: per tile
// the 64-channel slice of the x patch (with halo, 43.5 KB) and of the dy tile (32 KB) go global -> registers -> LDS once, and every tap is a
// shifted view of the same patch.  x and dy are read Co / 64 and Ci / 64 times in total (4 x at 256 channels).  Both MFMA operands need 8
// consecutive POSITIONS per lane from [position][channel] images: ds_read_b64_tr_b16 (two 4 x 16 transposing reads per 32 x 16 fragment);
// the 64-byte halves of a row are swapped on rows with bit 1 set, so that the four rows a 32-lane half reads cover all 64 banks; every
// fragment address is a lane constant + a compile-time offset.  The fp32 block is added once per workgroup to the packed workspace
// [Co][9][Ci] of sv_conv_wgrad (128 contiguous bytes per atomic instruction), which the caller unpacks as before.
// ------------------------------------------------------------------------------------------------
struct HaloWgradArgs { const void* x; const void* dy; float* ws; int N, H, W, Ci, Co; };

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_h;

// TH x TW output positions per tile, rows of 16 k (one K step = 16 positions of a tile row).  Stride 1: 8 x 32 or 16 x 16 (256 positions),
// whichever wastes less of the image - 56^2 -> 8 x 32 (87 % of the tile positions inside the image), 14^2 -> 16 x 16 (77 % against 38 %).
// Stride 2 (S = 2): 8 x 16 outputs from a 17 x 33 input patch (72 KB); a fragment's 8 positions are every second patch row.
// dbias (optional): column sums of dy - a tenth accumulator fed with an all-ones B operand in the workgroups of the first ci block.
template <int TH, int TW, int S, bool BIAS>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_halo_kernel(const HaloWgradArgs p, int Ho, int Wo, int tiles_h, int tiles_w, int ntiles, int nsplit,
                                                                    float* __restrict__ dbias) {
  static_assert(TW % 16 == 0 && (TH * TW) % 32 == 0, "tile rows of 16 k");
  constexpr int NQ = TH * TW, NKS = NQ / 16, PW = S * (TW - 1) + 3, PH = S * (TH - 1) + 3, NPOS = PH * PW;
  constexpr int NXC = NPOS * 8, NXL = (NXC + 255) / 256, NDL = NQ * 8 / 256;
  constexpr int X_BYTES = NPOS * 128, D_BYTES = NQ * 128;
  __shared__ __attribute__((aligned(1024))) char smem[X_BYTES + D_BYTES];
  char* Xl = smem;
  char* Dl = smem + X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, Wd = p.W, Ci = p.Ci, Co = p.Co;
  // workgroup -> (split, output block): the workgroups of one XCD (blockIdx.x % 8) take consecutive splits and all output blocks of a split -
  // they read the same position tiles, once from HBM into that XCD's L2
  const int nci = Ci >> 6, nob = nci * (Co >> 6);
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, spx = nsplit >> 3;
  const int split = xcd * spx + li / nob, ob = li % nob;
  const int cob = ob / nci, cib = ob - cob * nci;
  const int t0 = (int)((long long)ntiles * split / nsplit), t1 = (int)((long long)ntiles * (split + 1) / nsplit);
  if (t0 >= t1) return;
  const bool bias = BIAS && cib == 0;

  const unsigned xbytes = (unsigned)p.N * H * Wd * Ci * 2, dbytes = (unsigned)p.N * Ho * Wo * Co * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, dbytes, 0x00020000);
  const int tpi = tiles_h * tiles_w;
  // per-thread constants of its chunks (16 bytes = 8 channels of one position)
  int xpr[NXL], xpc[NXL];
  unsigned xrel[NXL];
#pragma unroll
  for (int j = 0; j < NXL; ++j) {
    const int c = tid + 256 * j, pp = c >> 3, sc = c & 7;
    xpr[j] = pp / PW - 1; xpc[j] = pp % PW - 1;
    xrel[j] = (unsigned)(((xpr[j] * Wd + xpc[j]) * Ci + cib * 64 + sc * 8) * 2);
  }
  u32x4 xpre[NXL], dpre[NDL];
  auto load_tile = [&](int t) {
    const int n = t / tpi, r = t - n * tpi, th = r / tiles_w, h0 = th * TH, w0 = (r - th * tiles_w) * TW;     // output tile origin
    const unsigned xpos0 = (unsigned)((n * H + S * h0) * Wd + S * w0), dpos0 = (unsigned)((n * Ho + h0) * Wo + w0);
#pragma unroll
    for (int j = 0; j < NXL; ++j) {
      const bool ok = (unsigned)(S * h0 + xpr[j]) < (unsigned)H && (unsigned)(S * w0 + xpc[j]) < (unsigned)Wd;
      xpre[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (xpos0 * Ci * 2 + xrel[j]) | ((unsigned)!ok << 31), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NDL; ++j) {
      const int c = tid + 256 * j, q = c >> 3, sc = c & 7, rr = q / TW, cc = q % TW;
      const bool ok = h0 + rr < Ho && w0 + cc < Wo;         // positions of the tile outside the image contribute nothing
      dpre[j] = __builtin_amdgcn_raw_buffer_load_b128(dr, ((dpos0 + rr * Wo + cc) * Co * 2 + (cob * 64 + sc * 8) * 2) | ((unsigned)!ok << 31), 0, 0);
    }
  };
  auto store_tile = [&]() {               // 16-byte slot s of row r -> s ^ 4 on rows with bit 1 set (the 64-byte halves swapped)
#pragma unroll
    for (int j = 0; j < NXL; ++j) {
      const int c = tid + 256 * j, pp = c >> 3, sc = c & 7;
      if (NXC % 256 == 0 || c < NXC) *reinterpret_cast<u32x4*>(Xl + pp * 128 + ((sc ^ ((pp & 2) << 1)) << 4)) = xpre[j];
    }
#pragma unroll
    for (int j = 0; j < NDL; ++j) {
      const int c = tid + 256 * j, q = c >> 3, sc = c & 7;
      *reinterpret_cast<u32x4*>(Dl + q * 128 + ((sc ^ ((q & 2) << 1)) << 4)) = dpre[j];
    }
  };

  // fragment addressing (32x32x16: lane = column l & 31 of the block, k = 8 (l >> 5) ..; per 16-lane group a 4 x 16 transposing read).
  // Row of the image a lane addresses = (lane part) + (compile-time part); which 64-byte half of the row holds its columns depends on bit 1 of
  // that row: bit 1 of the lane part's low bits XOR bit 1 of the compile-time part - two bases per kw, picked at compile time.
  const int mb = wave >> 1, nb = wave & 1, g = lane >> 4, lr = lane & 15, q4 = lr >> 2, pq = lr & 3;
  const int arow = 8 * (g >> 1) + q4;                                   // dy image: + 16 s + 4 hi
  const char* abase = Dl + arow * 128 + ((mb ^ (q4 >> 1)) << 6) + 32 * (g & 1) + 8 * pq;
  // x patch: fragment position kk = 8 (g >> 1) + q4 + 4 hi of the K step -> patch row (S r + kh) PW + S (c0 + kk) + kw
  const char* bbase[3][2];                                               // [kw][bit 1 of the compile-time part]
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int lanebit = S == 1 ? (((kw + q4) >> 1) & 1) : (q4 & 1);      // S = 2: row = C + 2 q4 + ...: bit 1 = bit1(C) ^ (q4 & 1)
    const int half = nb ^ lanebit;
#pragma unroll
    for (int par = 0; par < 2; ++par) bbase[kw][par] = Xl + (S * arow) * 128 + ((half ^ par) << 6) + 32 * (g & 1) + 8 * pq;
  }
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.f;

  f32x16 acc[9], accb;
#pragma unroll
  for (int j = 0; j < 16; ++j) accb[j] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  load_tile(t0);
  store_tile();
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    if (t + 1 < t1) load_tile(t + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {     // 16 positions per step: tile row s / (TW / 16), columns 16 (s % (TW / 16)) ..
      const int r = s_ / (TW / 16), c0 = 16 * (s_ % (TW / 16));
      const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)(abase + (16 * s_) * 128));
      const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)(abase + (16 * s_ + 4) * 128));
      const bf16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
      if constexpr (BIAS) { if (bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, accb, 0, 0, 0); }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          // compile-time part of the patch row: C = (S r + kh) PW + S c0 + kw; its bit 1 (S = 1: of the row term alone, the kw part is in lanebit)
          constexpr int dummy = 0; (void)dummy;
          const int crow = (S * r + kh) * PW + S * c0 + kw;
          const int par = S == 1 ? ((((S * r + kh) * PW + c0) >> 1) & 1) : ((crow >> 1) & 1);
          const char* bp = bbase[kw][par] + crow * 128;
          const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)(bp));
          const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_h*)(bp + S * 4 * 128));
          const bf16x8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[kh * 3 + kw], 0, 0, 0);
        }
    }
    __syncthreads();                       // every wave is done with the images
    if (t + 1 < t1) store_tile();
    __syncthreads();
  }
  // dW block: lane = ci column l & 31, register j = co row (j & 3) + 8 (j >> 2) + 4 (l >> 5) of the 32 x 32 sub-block
  float* wsb = p.ws + ((size_t)(cob * 64 + mb * 32) * 9) * Ci + cib * 64 + nb * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int co = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
      atomicAdd(wsb + ((size_t)co * 9 + t) * Ci, acc[t][j]);
    }
  if (BIAS && bias && nb == 0 && (lane & 31) == 0) {   // every column of the all-ones product holds the sums: one lane per row half writes them
#pragma unroll
    for (int j = 0; j < 16; ++j) atomicAdd(dbias + cob * 64 + mb * 32 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), accb[j]);
  }
}

static std::atomic<int>& halo_wgrad_mode() {
  static std::atomic<int> m{[] { const char* v = getenv("SV_CONV_HALO_WGRAD"); return v ? atoi(v) : 1; }()};
  return m;
}

// ------------------------------------------------------------------------------------------------
// 3 x 3 / stride 1 / padding 1 with MORE than 64 channels (conv2 of the layer2 / layer3 bottlenecks: 128 and 256 channels), forward and data
// gradient.  The weights no longer fit in LDS (9 x 256 x 256 bf16 = 1.1 MB), so a work item is (tile of 256 positions, block of 64 output
// channels) and its K loop runs over blocks of 64 INPUT channels: per block the [9][64][64] weight slice (72 KB) and the 64-channel slice of the
// patch (43-47 KB) go global -> registers (prefetched under the previous block's 144 MFMAs per wave) -> LDS, and the contraction of
// conv_halo_kernel runs on them; the accumulators live across the blocks.  Per MFLOP that is 6.2 KB of L2 -> LDS fill against 15 KB for the gather
// engine's 128 x 128 tile (which stages the activation tile once per tap).  Items are drawn at run time, the output blocks of a tile
// consecutively (they share the patch in L2).  Epilogue and statistics as in conv_halo_kernel, on the item's 128-byte piece of each output row.
// TH x TW = 8 x 32 or 16 x 16 as in the weight gradient (14 x 14 maps fill 77 % of a 16 x 16 tile, 38 % of an 8 x 32 one).
// ------------------------------------------------------------------------------------------------
struct HaloBlockedArgs {
  const void* x; const void* w; void* y; double* stats;
  int N, H, W, Ci, Co, flip;
};

template <int TH, int TW>
__global__ __launch_bounds__(256, 1) void conv3x3_halo_blocked_kernel(const HaloBlockedArgs p, int tiles_h, int tiles_w, int nitems, int* __restrict__ ctr) {
  static_assert(TH * TW == 256 && (TW == 32 || TW == 16), "256 positions per tile");
  constexpr int PW = TW + 2, NPOS = (TH + 2) * PW, PROWB = 144, NCH = NPOS * 8, NLD = (NCH + 255) / 256, NWL = 9 * 64 * 8 / 256;
  constexpr int W_BYTES = 9 * 64 * 128, PATCH_BYTES = (NPOS * PROWB + 127) / 128 * 128, STAGE_BYTES = 4 * 4096;
  constexpr int RPB = 32 / TW;                 // tile rows per block of 32 positions (1 or 2)
  __shared__ __attribute__((aligned(1024))) char smem[W_BYTES + PATCH_BYTES + STAGE_BYTES + 16];
  const __bf16* __restrict__ Wt = static_cast<const __bf16*>(p.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kg = lane >> 5;
  const int H = p.H, Wd = p.W, Ci = p.Ci, Co = p.Co, nci = Ci >> 6, nco = Co >> 6;
  char* Wl = smem;
  char* Pl = smem + W_BYTES;
  char* Sl = smem + W_BYTES + PATCH_BYTES + wave * 4096;
  int* mbox = reinterpret_cast<int*>(smem + W_BYTES + PATCH_BYTES + STAGE_BYTES);

  // ---- item scheduler (conv_halo_kernel's): 8 contiguous chunks of the item space, one per XCD; ctr[8] counts finished workgroups
  const int xcd = blockIdx.x & 7, cq = nitems >> 3, cr = nitems & 7;
  const int cbase = xcd * cq + (xcd < cr ? xcd : cr), csize = cq + (xcd < cr ? 1 : 0);
  if (tid == 0) {
    const int v0 = atomicAdd(ctr + xcd, 1), v1 = atomicAdd(ctr + xcd, 1);
    mbox[0] = v0 < csize ? cbase + v0 : -1;
    mbox[1] = v1 < csize ? cbase + v1 : -1;
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

  const unsigned xbytes = (unsigned)p.N * H * Wd * Ci * 2, ybytes = (unsigned)p.N * H * Wd * Co * 2, wbytes = (unsigned)Co * 9 * Ci * 2;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wt), 0, wbytes, 0x00020000);
  const int tpi = tiles_h * tiles_w;
  auto origin = [&](int item, int& n, int& h0, int& w0, int& cob) {
    const int t = item / nco;
    cob = item - t * nco;
    n = t / tpi;
    const int r = t - n * tpi, th = r / tiles_w;
    h0 = th * TH; w0 = (r - th * tiles_w) * TW;
  };
  // per-thread constants: patch chunks (position row / column, byte offset relative to the tile origin) and weight chunks (row r of the
  // slice = output channel, tap, 16-byte chunk c -> LDS slot of the [tap slot][r][64] image, source offset relative to the slice origin)
  int cpr[NLD], cpc[NLD];
  unsigned crel[NLD], wrel[NWL];
  int wdst[NWL];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int c = tid + 256 * j, pp = c >> 3, sc = c & 7;
    cpr[j] = pp / PW - 1; cpc[j] = pp % PW - 1;
    crel[j] = (unsigned)(((cpr[j] * Wd + cpc[j]) * Ci + sc * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < NWL; ++j) {
    const int id = tid + 256 * j, c = id & 7, rt = id >> 3, t = rt % 9, r = rt / 9;
    const int ts = p.flip ? 8 - t : t;
    wrel[j] = (unsigned)(((r * 9 + t) * Ci + c * 8) * 2);
    wdst[j] = (ts * 64 + r) * 128 + ((c ^ hc_swz<64>(r)) << 4);
  }
  u32x4 pre[NLD], wpre[NWL];
  auto load_block = [&](int item, int cib) {
    int n, h0, w0, cob;
    origin(item, n, h0, w0, cob);
    const unsigned base = (unsigned)((n * H + h0) * Wd + w0) * Ci * 2 + cib * 128;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const bool ok = (unsigned)(h0 + cpr[j]) < (unsigned)H && (unsigned)(w0 + cpc[j]) < (unsigned)Wd;
      pre[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (base + crel[j]) | ((unsigned)!ok << 31), 0, 0);
    }
    const unsigned wbase = (unsigned)(cob * 64) * 9 * Ci * 2 + cib * 128;
#pragma unroll
    for (int j = 0; j < NWL; ++j) wpre[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, wbase + wrel[j], 0, 0);
  };
  auto store_block = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int c = tid + 256 * j;
      if (NCH % 256 == 0 || c < NCH) *reinterpret_cast<u32x4*>(Pl + (c >> 3) * PROWB + (c & 7) * 16) = pre[j];
    }
#pragma unroll
    for (int j = 0; j < NWL; ++j) *reinterpret_cast<u32x4*>(Wl + wdst[j]) = wpre[j];
  };

  double sd[4] = {0.0, 0.0, 0.0, 0.0};
  const int swa = hc_swz<64>(ln);
  const char* pa = Wl + ln * 128;
  // block nb of the wave = 32 positions: tile rows (4 w + RPB nb) .. , position ln -> row ln / TW, column ln % TW
  const char* pq[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) pq[nb] = Pl + (((2 * wave + nb) * RPB + ln / TW) * PW + ln % TW) * PROWB + kg * 16;
  int aoff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) aoff[ks] = ((2 * ks + kg) ^ swa) << 4;

  load_block(cur, 0);
  while (true) {
    int drawn;
    if (tid == 0) drawn = atomicAdd(ctr + xcd, nxt >= 0 ? 1 : 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mb][nb][j] = 0.f;
    for (int cib = 0; cib < nci; ++cib) {
      __syncthreads();                   // everyone is done with the previous slices
      store_block();
      __syncthreads();
      // next slices: the next input-channel block of this item, or the first one of the next item (the last item re-reads its own)
      if (cib + 1 < nci) load_block(cur, cib + 1);
      else load_block(nxt >= 0 ? nxt : cur, 0);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 fa[2][2], fb[2][2];
      auto frags = [&](int step, bf16x8* a, bf16x8* b) {
        const int t = step >> 2, ks = step & 3, kh = t / 3, kw = t - kh * 3;
        a[0] = *reinterpret_cast<const bf16x8*>(pa + t * 64 * 128 + aoff[ks]);
        b[0] = *reinterpret_cast<const bf16x8*>(pq[0] + (kh * PW + kw) * PROWB + ks * 32);
        a[1] = *reinterpret_cast<const bf16x8*>(pa + (t * 64 + 32) * 128 + aoff[ks]);
        b[1] = *reinterpret_cast<const bf16x8*>(pq[1] + (kh * PW + kw) * PROWB + ks * 32);
      };
      frags(0, fa[0], fb[0]);
#pragma unroll
      for (int step = 0; step < 36; ++step) {
        const int c = step & 1;
        if (step + 1 < 36) frags(step + 1, fa[c ^ 1], fb[c ^ 1]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
        if (step + 1 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (step + 1 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
    }
    if (tid == 0) mbox[0] = (nxt >= 0 && drawn < csize) ? cbase + drawn : -1;

    // ---- epilogue of `cur` (conv_halo_kernel's, on the 64 output channels of the item)
    int n, h0, w0, cob;
    origin(cur, n, h0, w0, cob);
    const int ncol = min(TW, Wd - w0);
    f32x2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int row0 = h0 + (2 * wave + nb) * RPB;             // first tile row of the block
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
          bf16x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (__bf16)acc[mb][nb][jg * 4 + i];
          *reinterpret_cast<bf16x4*>(Sl + ln * 128 + (((mb * 4 + jg) ^ (ln & 7)) << 4) + kg * 8) = o;
        }
      }
#pragma unroll
      for (int itr = 0; itr < 4; ++itr) {
        const int q = itr * 8 + (lane >> 3), c8 = lane & 7, row = row0 + q / TW, col = w0 + q % TW;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Sl + q * 128 + ((c8 ^ (q & 7)) << 4));
        const unsigned off = ((unsigned)((n * H + row) * Wd + col) * Co * 2 + cob * 128 + c8 * 16) | ((unsigned)!(row < H && col < Wd) << 31);
        __builtin_amdgcn_raw_buffer_store_b128(v, yr, off, 0, 0);
      }
      if (p.stats) {                     // lane (kg, cp = ln): channels 2 cp, 2 cp + 1 over positions 16 kg .. 16 kg + 15 of the block
        const int prow = row0 + (TW == 16 ? kg : 0), pc0 = TW == 16 ? 0 : 16 * kg;
        if (prow < H) {
          const char* srow = Sl + kg * 16 * 128 + (ln & 3) * 4;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned u = *reinterpret_cast<const unsigned*>(srow + i * 128 + ((((ln >> 2) ^ i) & 7) << 4));
            const float m = pc0 + i < ncol ? 1.f : 0.f;
            const f32x2 v = {__uint_as_float(u << 16) * m, __uint_as_float(u & 0xffff0000u) * m};
            sa += v; sb += v * v;
          }
        }
      }
    }
    if (p.stats) {
      sd[0] += (double)sa[0]; sd[1] += (double)sb[0]; sd[2] += (double)sa[1]; sd[3] += (double)sb[1];
      // one item = one block of output channels: the sums go out per item (the next item of the workgroup may be another block)
      double* st = p.stats + (size_t)((blockIdx.x * 8 + wave * 2 + kg) % SV_BN_SLOTS) * 2 * Co + cob * 64;
      atomicAdd(st + 2 * ln, sd[0]);
      atomicAdd(st + Co + 2 * ln, sd[1]);
      atomicAdd(st + 2 * ln + 1, sd[2]);
      atomicAdd(st + Co + 2 * ln + 1, sd[3]);
      sd[0] = sd[1] = sd[2] = sd[3] = 0.0;
    }
    __syncthreads();                     // the mailbox is posted, every wave is past its reads of it
    if (nxt < 0) break;
    cur = nxt;
    nxt = __builtin_amdgcn_readfirstlane(mbox[0]);
  }
  if (tid == 0) finish();
}

static std::atomic<long long> halo_launches{0};
bool conv_halo_enabled() { return halo_mode().load(std::memory_order_relaxed) != 0; }

int conv_halo_launch(const HaloConvArgs& a, int kind, hipStream_t stream) {
  if (!conv_halo_enabled()) return 0;
  const int tiles_h = cdiv(a.H, 8), tiles_w = cdiv(a.W, 32);
  const long long nt = (long long)a.N * tiles_h * tiles_w;
  const int mode = halo_mode().load(std::memory_order_relaxed);
  if ((long long)a.N * a.H * a.W * 128 >= (1ll << 31)) return 0;          // buffer descriptors: 32-bit byte offsets, bit 31 = "outside"
  if ((mode == 1 && nt < 256) || nt >= (1ll << 30)) return 0;   // fewer tiles than CUs: 72 KB of weights per workgroup for one tile each
  int* ctr = tile_draw_counters();
  if (!ctr) return 0;
  const int ntiles = (int)nt;
  if (kind == 0)
    hipLaunchKernelGGL((conv_halo_kernel<64, 3, 3, 1, 1>), dim3(256), dim3(256), 0, stream, a, tiles_h, tiles_w, ntiles, ctr);
  else   // 32 KB of weights + 18 KB patch + 16 KB staging: two workgroups per CU - one's epilogue runs under the other's MFMAs
    hipLaunchKernelGGL((conv_halo_kernel<16, 4, 4, 2, 2>), dim3(512), dim3(256), 0, stream, a, tiles_h, tiles_w, ntiles, ctr);
  halo_launches.fetch_add(1, std::memory_order_relaxed);
  return 1;
}

// 3 x 3 / stride 1 / padding 1 with Ci, Co multiples of 64 (not both 64: that is conv_halo_launch's resident-weights kernel); 1 = taken
int conv_halo_blocked_launch(const void* x, const void* w, void* y, double* stats, int N, int H, int W, int Ci, int Co, int flip, hipStream_t stream) {
  const int mode = halo_mode().load(std::memory_order_relaxed);
  static const int blocked_on = [] { const char* v = getenv("SV_CONV_HALO_BLOCKED"); return v ? atoi(v) : 1; }();
  if (mode == 0 || !blocked_on || (Ci & 63) || (Co & 63) || Ci > 512 || Co > 512) return 0;
  if ((long long)N * H * W * (Ci > Co ? Ci : Co) * 2 >= (1ll << 31)) return 0;
  const double u0 = (double)H * W / ((double)cdiv(H, 8) * 8 * cdiv(W, 32) * 32), u1 = (double)H * W / ((double)cdiv(H, 16) * 16 * cdiv(W, 16) * 16);
  const bool sq = u1 > u0;
  if (mode == 1 && (sq ? u1 : u0) < 0.5) return 0;
  const int tiles_h = cdiv(H, sq ? 16 : 8), tiles_w = cdiv(W, sq ? 16 : 32);
  const long long ni = (long long)N * tiles_h * tiles_w * (Co >> 6);
  if ((mode == 1 && ni < 1024) || ni >= (1ll << 30)) return 0;
  int* ctr = tile_draw_counters();
  if (!ctr) return 0;
  HaloBlockedArgs a{x, w, y, stats, N, H, W, Ci, Co, flip};
  if (sq) hipLaunchKernelGGL((conv3x3_halo_blocked_kernel<16, 16>), dim3(256), dim3(256), 0, stream, a, tiles_h, tiles_w, (int)ni, ctr);
  else hipLaunchKernelGGL((conv3x3_halo_blocked_kernel<8, 32>), dim3(256), dim3(256), 0, stream, a, tiles_h, tiles_w, (int)ni, ctr);
  halo_launches.fetch_add(1, std::memory_order_relaxed);
  return 1;
}

// 3 x 3 / padding 1 / stride 1 or 2 weight gradient into the packed workspace ws [Co][9][Ci] (zero on entry), dbias += column sums of dy; 1 = taken
int conv_halo_wgrad_launch(const void* x, const void* dy, float* ws, float* dbias, int N, int H, int W, int Ho, int Wo, int stride, int Ci, int Co,
                           hipStream_t stream) {
  const int mode = halo_wgrad_mode().load(std::memory_order_relaxed);
  if (mode == 0 || (Ci & 63) || (Co & 63) || (stride != 1 && stride != 2)) return 0;
  if (Ho != (H + 2 - 3) / stride + 1 || Wo != (W + 2 - 3) / stride + 1) return 0;
  if ((long long)N * H * W * Ci * 2 >= (1ll << 31) || (long long)N * Ho * Wo * Co * 2 >= (1ll << 31)) return 0;
  // tile shape by the share of tile positions that lie inside the (output) image
  const double u0 = (double)Ho * Wo / ((double)cdiv(Ho, 8) * 8 * cdiv(Wo, 32) * 32), u1 = (double)Ho * Wo / ((double)cdiv(Ho, 16) * 16 * cdiv(Wo, 16) * 16);
  const double u2 = (double)Ho * Wo / ((double)cdiv(Ho, 8) * 8 * cdiv(Wo, 16) * 16);
  const bool sq = u1 > u0;
  const double util = stride == 2 ? u2 : (sq ? u1 : u0);
  if (mode == 1 && util < 0.5) return 0;                    // small maps (7 x 7): most of every tile would be padding
  if (mode == 1 && stride == 2 && Ho * Wo < 28 * 28 / 2) return 0;     // stride 2 below 28 x 28 outputs: wgrad_kernel is as fast (14^2 -> 7^2: 106 vs 129 us)
  const int th = stride == 2 ? 8 : (sq ? 16 : 8), tw = stride == 2 ? 16 : (sq ? 16 : 32);
  const int tiles_h = cdiv(Ho, th), tiles_w = cdiv(Wo, tw);
  const long long nt = (long long)N * tiles_h * tiles_w;
  const int nob = (Ci >> 6) * (Co >> 6);
  // splits: a multiple of 8 (one group of consecutive splits per XCD), about one workgroup per CU, at least 4 tiles per split
  int nsplit = 256 / nob; nsplit = nsplit / 8 * 8;
  if (nsplit < 8) nsplit = 8;
  if (mode == 1 && nt < 4ll * nsplit) return 0;
  if (nt >= (1ll << 30)) return 0;
  HaloWgradArgs a{x, dy, ws, N, H, W, Ci, Co};
  const dim3 grid(nsplit * nob);
#define SV_HALO_WG(TH_, TW_, S_)                                                                                                                           \
  do {                                                                                                                                                   \
    if (dbias) hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<TH_, TW_, S_, true>), grid, dim3(256), 0, stream, a, Ho, Wo, tiles_h, tiles_w, (int)nt, nsplit, dbias); \
    else hipLaunchKernelGGL((conv3x3_wgrad_halo_kernel<TH_, TW_, S_, false>), grid, dim3(256), 0, stream, a, Ho, Wo, tiles_h, tiles_w, (int)nt, nsplit, dbias);      \
  } while (0)
  if (stride == 2) SV_HALO_WG(8, 16, 2);
  else if (sq) SV_HALO_WG(16, 16, 1);
  else SV_HALO_WG(8, 32, 1);
#undef SV_HALO_WG
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
#ifdef SV_HC_PROFILE
extern "C" int sv_conv_halo_prof_wg(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sv::hc_wg), sizeof(long long) * 2048) == hipSuccess ? 0 : -1;
}
extern "C" int sv_conv_halo_prof(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sv::hc_prof), sizeof(long long) * 8) != hipSuccess) return -1;
  if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sv::hc_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
extern "C" int sv_set_conv_halo_wgrad(int mode) {
  SV_REQUIRE(mode >= 0 && mode <= 2, "sv_set_conv_halo_wgrad: mode %d (0 off, 1 auto, 2 always)", mode);
  sv::halo_wgrad_mode().store(mode, std::memory_order_relaxed);
  return SV_OK;
}
