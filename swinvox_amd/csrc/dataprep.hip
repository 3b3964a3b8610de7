// Input preparation on the device: the reference does this per sample in DataLoader workers with numpy/cv2
// (utils/data_loaders.py:52-88, utils/data_transforms.py, utils/binvox_rw.py:118-149); at thousands of views per second that
// CPU path is the bottleneck, so the raw bytes (RLE voxel runs, 8-bit renderings) are shipped and expanded here.
//
//   sv_binvox_decode  run-length (value, count) byte pairs -> dense float occupancy, xzy -> xyz transposition
//   sv_augment_views  crop -> bilinear resize -> background composite -> colour jitter -> PCA noise -> normalise ->
//                     flip -> channel permutation -> CHW float, two launches (grey means for the contrast step, then the image)
#include "common.h"

namespace sv {

// ---- binvox: one workgroup per volume; runs are placed with a running prefix sum over the counts -------------------------
__global__ void __launch_bounds__(256) binvox_decode_kernel(const unsigned char* __restrict__ rle, const long long* __restrict__ pair_off, int d0, int d1,
                                                            int d2, int fix_coords, float* __restrict__ out, int* __restrict__ decoded) {
  __shared__ int wsum[4];
  __shared__ int running_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long p0 = pair_off[b], p1 = pair_off[b + 1];
  const int total = d0 * d1 * d2;
  float* o = out + (size_t)b * total;
  if (tid == 0) running_s = 0;
  __syncthreads();
  for (long long base = p0; base < p1; base += 256) {
    const long long p = base + tid;
    int val = 0, cnt = 0;
    if (p < p1) {
      val = rle[2 * p];
      cnt = rle[2 * p + 1];
    }
    int inc = cnt;                               // inclusive scan inside the wave, then across the four waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int start = running_s + inc - cnt;
    for (int i = 0; i < w; ++i) start += wsum[i];
    const float v = val ? 1.f : 0.f;             // .astype(bool)
    for (int t = 0; t < cnt; ++t) {
      const int f = start + t;
      if (f >= total) break;                     // malformed stream: the host sees decoded != total
      if (fix_coords) {                          // file order is [x][z][y]; np.transpose(data, (0, 2, 1)) -> [x][y][z]
        const int k = f % d2, j = (f / d2) % d1, i = f / (d1 * d2);
        o[((size_t)i * d2 + k) * d1 + j] = v;
      } else {
        o[f] = v;
      }
    }
    __syncthreads();
    if (tid == 0) running_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) decoded[b] = running_s;
}

// ---- view augmentation -----------------------------------------------------------------------------------------------------
struct AugGeom {
  int I, V, Hs, Ws, C, x0, y0, cw, ch, Ho, Wo;
};

// cv2.resize(..., INTER_LINEAR) source coordinate of a destination index (float arithmetic as in OpenCV's resize.cpp)
__device__ __forceinline__ void lin_coord(int d, double scale, int ssize, int& s, float& f) {
  float fx = (float)((d + 0.5) * scale - 0.5);
  s = (int)floorf(fx);
  fx -= (float)s;
  if (s < 0) { fx = 0.f; s = 0; }
  if (s >= ssize - 1) { fx = 0.f; s = ssize - 1; }
  f = fx;
}

// resized crop at (oy, ox): horizontal pass on the two source rows, then the vertical pass; products and sums are rounded
// separately (no fused multiply-add) so that the CPU restatement reproduces them bit for bit
__device__ __forceinline__ void sample_pixel(const unsigned char* __restrict__ img, const AugGeom& g, int oy, int ox, float px[4]) {
  int sx, sy;
  float fx, fy;
  lin_coord(ox, (double)g.cw / g.Wo, g.cw, sx, fx);
  lin_coord(oy, (double)g.ch / g.Ho, g.ch, sy, fy);
  const int sx1 = min(sx + 1, g.cw - 1), sy1 = min(sy + 1, g.ch - 1);
  const unsigned char* r0 = img + ((size_t)(g.y0 + sy) * g.Ws + g.x0) * g.C;
  const unsigned char* r1 = img + ((size_t)(g.y0 + sy1) * g.Ws + g.x0) * g.C;
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  for (int c = 0; c < g.C; ++c) {
    const float t00 = (float)r0[sx * g.C + c] / 255.f, t01 = (float)r0[sx1 * g.C + c] / 255.f;
    const float t10 = (float)r1[sx * g.C + c] / 255.f, t11 = (float)r1[sx1 * g.C + c] / 255.f;
    const float h0 = __fadd_rn(__fmul_rn(t00, a0), __fmul_rn(t01, a1));
    const float h1 = __fadd_rn(__fmul_rn(t10, a0), __fmul_rn(t11, a1));
    px[c] = __fadd_rn(__fmul_rn(h0, b0), __fmul_rn(h1, b1));
  }
}

// RandomBackground: pixels whose (resized) alpha is exactly zero take the background colour
__device__ __forceinline__ void composite(float px[4], int C, const float bg[3]) {
  if (C == 4 && px[3] == 0.f) {
    px[0] = bg[0];
    px[1] = bg[1];
    px[2] = bg[2];
  }
}
__device__ __forceinline__ float grey_of(const float px[4]) { return 0.114f * px[0] + 0.587f * px[1] + 0.299f * px[2]; }

__global__ void __launch_bounds__(256) augment_grey_mean_kernel(const unsigned char* __restrict__ src, AugGeom g, const sv_aug_sample* __restrict__ prm,
                                                                double* __restrict__ grey_sum) {
  __shared__ float red[4];
  const int img = blockIdx.y;
  const sv_aug_sample& P = prm[img / g.V];
  const unsigned char* s = src + (size_t)img * g.Hs * g.Ws * g.C;
  const int npix = g.Ho * g.Wo;
  float acc = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    float px[4];
    sample_pixel(s, g, p / g.Wo, p % g.Wo, px);
    composite(px, g.C, P.bg);
    acc += grey_of(px);
  }
  const float r = block_sum<4>(acc, red);
  if (threadIdx.x == 0) atomicAdd(&grey_sum[img], (double)r);
}

__global__ void __launch_bounds__(256) augment_apply_kernel(const unsigned char* __restrict__ src, AugGeom g, const sv_aug_sample* __restrict__ prm,
                                                            const unsigned char* __restrict__ flip, const double* __restrict__ grey_sum,
                                                            float* __restrict__ out) {
  const int img = blockIdx.y;
  const sv_aug_sample P = prm[img / g.V];
  const unsigned char* s = src + (size_t)img * g.Hs * g.Ws * g.C;
  const int npix = g.Ho * g.Wo;
  const bool fl = flip != nullptr && flip[img] != 0;
  float mean_grey = (float)(grey_sum[img] / (double)npix);
  // the contrast step blends with the mean grey of the image as it is at that point of the jitter order: brightness scales it,
  // saturation and contrast itself leave it unchanged
  float contrast_mean = mean_grey;
  for (int k = 0; k < 3; ++k) {
    if (P.jitter_order[k] == 1) break;
    if (P.jitter_order[k] == 0) contrast_mean *= P.jitter_value[0];
  }
  float* o = out + (size_t)img * 3 * npix;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
    const int oy = p / g.Wo, ox = p % g.Wo;
    float px[4];
    sample_pixel(s, g, oy, fl ? g.Wo - 1 - ox : ox, px);           // np.fliplr of the finished image = sampling the mirrored column
    composite(px, g.C, P.bg);
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                  // ColorJitter: 0 brightness, 1 contrast, 2 saturation
      const int kind = P.jitter_order[k];
      const float a = P.jitter_value[kind];
      const float other = kind == 0 ? 0.f : (kind == 1 ? contrast_mean : grey_of(px));
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = a * px[c] + (1.f - a) * other;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = (px[c] + P.noise[c] - P.mean[c]) / P.std[c];   // RandomNoise (already in channel order), Normalize
#pragma unroll
    for (int c = 0; c < 3; ++c) o[(size_t)c * npix + p] = px[P.perm[c]];                // RandomPermuteRGB, ToTensor (HWC -> CHW)
  }
}

}  // namespace sv

using namespace sv;
#define STREAM static_cast<hipStream_t>(stream)

extern "C" int sv_binvox_decode(const unsigned char* rle, const long long* pair_offsets, int B, int d0, int d1, int d2, int fix_coords,
                                float* out, int* decoded, void* stream) {
  SV_REQUIRE(rle && pair_offsets && out && decoded && B > 0 && d0 > 0 && d1 > 0 && d2 > 0, "binvox_decode: bad arguments");
  SV_REQUIRE((long long)d0 * d1 * d2 < (1ll << 30), "binvox_decode: volume too large");
  hipLaunchKernelGGL(binvox_decode_kernel, dim3(B), dim3(256), 0, STREAM, rle, pair_offsets, d0, d1, d2, fix_coords, out, decoded);
  return check_launch("sv_binvox_decode");
}

extern "C" int sv_augment_views(const unsigned char* src, int I, int V, int Hs, int Ws, int C, int crop_h, int crop_w, int out_h, int out_w,
                                const sv_aug_sample* params_dev, const unsigned char* flip_dev, double* grey_sum_ws, float* out, void* stream) {
  SV_REQUIRE(src && params_dev && grey_sum_ws && out, "augment_views: bad arguments");
  SV_REQUIRE(I > 0 && V > 0 && I % V == 0 && (C == 3 || C == 4) && Hs > 0 && Ws > 0 && out_h > 0 && out_w > 0, "augment_views: bad shapes");
  AugGeom g;
  g.I = I; g.V = V; g.Hs = Hs; g.Ws = Ws; g.C = C; g.Ho = out_h; g.Wo = out_w;
  if (Hs > crop_h && Ws > crop_w && crop_h > 0 && crop_w > 0) {      // centre crop (data_transforms.py:135-139 / :222-226, no bounding box)
    g.x0 = (Ws - crop_w) / 2; g.y0 = (Hs - crop_h) / 2; g.cw = crop_w; g.ch = crop_h;
  } else {
    g.x0 = 0; g.y0 = 0; g.cw = Ws; g.ch = Hs;
  }
  hipError_t e = hipMemsetAsync(grey_sum_ws, 0, sizeof(double) * (size_t)I, STREAM);
  if (e != hipSuccess) { set_error("augment_views: memset failed: %s", hipGetErrorString(e)); return SV_ERR_LAUNCH; }
  const int bx = cdiv((long long)out_h * out_w, 256 * 4);
  hipLaunchKernelGGL(augment_grey_mean_kernel, dim3(bx, I), dim3(256), 0, STREAM, src, g, params_dev, grey_sum_ws);
  hipLaunchKernelGGL(augment_apply_kernel, dim3(bx, I), dim3(256), 0, STREAM, src, g, params_dev, flip_dev, grey_sum_ws, out);
  return check_launch("sv_augment_views");
}
