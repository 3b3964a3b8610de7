// Halo-tile convolution (conv_halo.hip): the interface igemm.hip dispatches through.  Internal header - not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace sv {

struct HaloConvArgs {
  const void* x;      // [N][H][W][CI] bf16, rows of exactly CI elements
  const void* w;      // packed weights [64][taps][CI] bf16 (forward pack, or the data-gradient pack with flip = 1)
  void* y;            // [N][H][W][64] bf16, rows of exactly 64 elements
  double* stats;      // optional [SV_BN_SLOTS][2 * 64]: per-channel sum / sum of squares of the stored (bf16-rounded) results
  int N, H, W;
  int flip;           // 1: tap t of the pack is applied at the mirrored offset (data gradient of a stride-1 convolution)
};

// 1 when the call was taken (launched on `stream`), 0 when the shape is not one of the kernel's (the caller goes on with the engine).
// kind: 0 = 3 x 3, padding 1, CI = 64;  1 = 4 x 4, padding (2, 1), CI = 16 (the ResNet stem on the space-to-depth image)
int conv_halo_launch(const HaloConvArgs& a, int kind, hipStream_t stream);
bool conv_halo_enabled();
// the same convolution (3 x 3 / stride 1 / padding 1, plain store + optional statistics) with Ci, Co multiples of 64 up to 512: work items of 256
// positions x 64 output channels with a K loop over blocks of 64 input channels; w = the [Co][9][Ci] pack (data-gradient pack with flip = 1)
int conv_halo_blocked_launch(const void* x, const void* w, void* y, double* stats, int N, int H, int W, int Ci, int Co, int flip, hipStream_t stream);

// 3 x 3 / padding 1 / stride 1 or 2 weight gradient (Ci, Co multiples of 64, rows of exactly Ci / Co elements) into the packed workspace
// ws [Co][9][Ci] fp32 (zero on entry), dbias (optional) += column sums of dy; 1 = taken, 0 = not this kernel's shape or switched off
int conv_halo_wgrad_launch(const void* x, const void* dy, float* ws, float* dbias, int N, int H, int W, int Ho, int Wo, int stride, int Ci, int Co,
                           hipStream_t stream);

// launch slot of the run-time tile schedulers (igemm.hip): 16 ints, zero on entry, zeroed again by the launch's last workgroup
int* tile_draw_counters();

}  // namespace sv
