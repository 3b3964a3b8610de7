// What v_permlane16_swap / v_permlane32_swap leave in their two registers (gfx950): prints, per lane, the source lane of both results.
//   hipcc --offload-arch=gfx950 -O2 permlane_swap_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r16[0]; o[64 + threadIdx.x] = r16[1]; o[128 + threadIdx.x] = r32[0]; o[192 + threadIdx.x] = r32[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" l%d<-%u", l, h[r * 64 + l]); printf("\n"); }
  return 0;
}
