// L2/HBM -> LDS fill-rate probe for the operand stream of a dense GEMM (no MFMA, no stores): persistent 256-thread workgroups, two per CU,
// walk (row tile, column tile) pairs of C = A[M][K] x W[N][K]^T exactly as gemm_wide_kernel does and DMA the K slices of both operands
// into a 3-slot LDS ring.  ROWB = bytes of one row fetched per slice (64: 16 rows x 64 B per wave instruction; 128: 8 rows x 128 B).
//   hipcc --offload-arch=gfx950 -O3 -o dma_fill_probe dma_fill_probe.hip && ./dma_fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// BM x BN tile, slice = ROWB bytes of every row; ring of NS slots
template <int ROWB, int BM, int BN, int NS, int WPS, int NW>
__global__ __launch_bounds__(NW * 64, WPS) void fill_kernel(const char* A, const char* W, int M, int N, int Kbytes, int ntiles, int* sink) {
  constexpr int SLOT = (BM + BN) * ROWB;
  __shared__ __attribute__((aligned(1024))) char smem[NS * SLOT];
  constexpr int RPP = 1024 / ROWB;                 // rows per DMA piece
  constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;   // pieces per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (N + BN - 1) / BN, nk = Kbytes / ROWB;
  const int lrow = lane / (ROWB / 16), lcol = (lane % (ROWB / 16)) * 16;
  for (int it = 0;; ++it) {
    const int w0 = it * (int)gridDim.x;
    int left = ntiles - w0; if (left > (int)gridDim.x) left = gridDim.x;
    if (left <= 0 || (int)blockIdx.x >= left) break;
    const int tlin = w0 + xcd_remap(blockIdx.x, left);
    const int row0 = (tlin / tiles_n) * BM, col0 = (tlin % tiles_n) * BN;
    const char* pa[PA]; const char* pb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) { int m = row0 + (wave * PA + i) * RPP + lrow; if (m >= M) m = M - 1; pa[i] = A + (size_t)m * Kbytes + lcol; }
#pragma unroll
    for (int i = 0; i < PB; ++i) { int n = col0 + (wave * PB + i) * RPP + lrow; if (n >= N) n = N - 1; pb[i] = W + (size_t)n * Kbytes + lcol; }
    auto issue = [&](int kt, int st) {
      char* base = smem + st * SLOT;
#pragma unroll
      for (int i = 0; i < PA; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + kt * ROWB), (lptr_t)(base + (wave * PA + i) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + kt * ROWB), (lptr_t)(base + BM * ROWB + (wave * PB + i) * 1024), 16, 0, 0);
    };
    int st = 0;
    for (int s = 0; s < NS - 1 && s < nk; ++s) issue(s, s);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + NS - 1 < nk) {
        if constexpr (NS == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(PA + PB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        issue(kt + NS - 1, (st + NS - 1) % NS);
      } else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      st = (st + 1) % NS;
    }
    asm volatile("s_barrier" ::: "memory");
  }
  if (sink && smem[threadIdx.x] == 77 && smem[threadIdx.x + 4096] == 78) sink[0] = 1;
}

template <int ROWB, int BM, int BN, int NS, int WPS, int NW = 4>
static void run(const char* tag, const char* A, const char* W, int M, int N, int K, int* sink) {
  const int ntiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int nb = ntiles < 256 * WPS ? ntiles : 256 * WPS;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((fill_kernel<ROWB, BM, BN, NS, WPS, NW>), dim3(nb), dim3(NW * 64), 0, 0, A, W, M, N, K * 2, ntiles, sink);
  hipEventRecord(e0);
  const int iters = 5;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fill_kernel<ROWB, BM, BN, NS, WPS, NW>), dim3(nb), dim3(NW * 64), 0, 0, A, W, M, N, K * 2, ntiles, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double bytes = (double)ntiles * (BM + BN) * K * 2;
  printf("%-34s M=%7d K=%5d N=%5d  %8.1f us  %6.2f TB/s filled  (%5.1f GB/s per CU)  unique %.0f MB\n", tag, M, K, N, ms * 1e3, bytes / ms / 1e9,
         bytes / ms / 1e6 / 256, ((double)M * K + (double)N * K) * 2 / 1e6);
}

int main() {
  const int shapes[][3] = {{100352, 1536, 384}, {100352, 384, 1536}, {100352, 384, 384}, {25088, 3072, 768}, {25088, 768, 3072}, {100352, 1024, 256}, {100352, 1600, 384}};
  char *A, *W; int* sink;
  hipMalloc(&A, (size_t)100352 * 1600 * 2 + (1 << 20)); hipMalloc(&W, (size_t)3072 * 3072 * 2); hipMalloc(&sink, 4);
  hipMemset(A, 1, (size_t)100352 * 1600 * 2); hipMemset(W, 1, (size_t)3072 * 3072 * 2);
  for (auto& s : shapes) {
    run<64, 256, 128, 3, 2>("64B rows  256x128 ring3 2wg/cu", A, W, s[0], s[2], s[1], sink);
    run<128, 128, 64, 3, 2>("128B rows 128x64  ring3 2wg/cu", A, W, s[0], s[2], s[1], sink);
    run<128, 128, 128, 2, 2>("128B rows 128x128 ring2 2wg/cu", A, W, s[0], s[2], s[1], sink);
    run<128, 256, 128, 2, 1>("128B rows 256x128 ring2 1wg/cu", A, W, s[0], s[2], s[1], sink);
    run<128, 256, 256, 2, 1>("128B rows 256x256 ring2 1wg/cu", A, W, s[0], s[2], s[1], sink);
    run<64, 256, 256, 3, 1>("64B rows  256x256 ring3 1wg/cu", A, W, s[0], s[2], s[1], sink);
    run<128, 256, 128, 3, 1, 8>("128B rows 256x128 ring3 1wg 8w", A, W, s[0], s[2], s[1], sink);
    run<128, 256, 256, 2, 1, 8>("128B rows 256x256 ring2 1wg 8w", A, W, s[0], s[2], s[1], sink);
    run<256, 128, 128, 2, 1, 8>("256B rows 128x128 ring2 1wg 8w", A, W, s[0], s[2], s[1], sink);
    run<256, 128, 64, 3, 1, 8>("256B rows 128x64  ring3 1wg 8w", A, W, s[0], s[2], s[1], sink);
  }
  return 0;
}
