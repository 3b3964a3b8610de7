// HBM-bound glue kernels of the SwinVox path: layout transposes, poolings, the decoder seed / head, the
// cross-view up-sampling, the merger's view softmax, dropout masks, the BCE loss and the IoU counters.
// All tensors channels-last fp32 unless said otherwise; 16-byte accesses wherever the layout allows.
#include "common.h"

namespace sv {

static inline unsigned grid_for(long long n, int per_block = 256, int cap = 8192) {
  long long b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// dst[b][c][r] = src[b][r][c]; 32x32 LDS tiles (+1 pad), coalesced on both sides
template <typename AT>
__global__ __launch_bounds__(256) void transpose_kernel(const AT* __restrict__ src, AT* __restrict__ dst, int R, int C,
                                                        int lds_, int ldd, long long sb, long long db) {
  __shared__ AT tile[32][sizeof(AT) == 4 ? 33 : 34];
  const AT* s = src + (size_t)blockIdx.z * sb;
  AT* d = dst + (size_t)blockIdx.z * db;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) tile[ty + 8 * i][tx] = s[(size_t)r * lds_ + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) d[(size_t)c * ldd + r] = tile[tx][ty + 8 * i];
  }
}

// out[r, col_off + c] = a + b (+ c + d)
template <typename AT>
__global__ __launch_bounds__(256) void add_n_kernel(const AT* __restrict__ a, const AT* __restrict__ b, const AT* __restrict__ c,
                                                    const AT* __restrict__ d, AT* __restrict__ out, long long M, int C, int ldo) {
  const int cv = C >> 2;
  const long long total = M * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cv; const int col = (int)(i - r * cv) * 4;
    const size_t o = (size_t)r * C + col;
    float4 v = ld4f(a + o);
    const float4 w = ld4f(b + o);
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    if (c) { const float4 u = ld4f(c + o); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    if (d) { const float4 u = ld4f(d + o); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    st4f(out + (size_t)r * ldo + col, v);
  }
}

template <typename AT>
__global__ __launch_bounds__(256) void axpby_kernel(const AT* __restrict__ a, const AT* __restrict__ b, AT* __restrict__ out,
                                                    float alpha, float beta, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    stf(out + i, alpha * ldf(a + i) + (b ? beta * ldf(b + i) : 0.f));
}
template <typename AT>
__global__ __launch_bounds__(256) void axpby_vec_kernel(const AT* __restrict__ a, const AT* __restrict__ b, AT* __restrict__ out,
                                                        float alpha, float beta, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = ld4f(a + i * 4);
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    if (b) { const float4 w = ld4f(b + i * 4); v.x += beta * w.x; v.y += beta * w.y; v.z += beta * w.z; v.w += beta * w.w; }
    st4f(out + i * 4, v);
  }
}

// out = dy * (y > 0)   (ReLU backward where no contraction epilogue can absorb it)
template <typename AT>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const AT* __restrict__ dy, const AT* __restrict__ y, AT* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) stf(out + i, ldf(y + i) > 0.f ? ldf(dy + i) : 0.f);
}

// ---- ResNet stem max-pool 3x3 s2 p1 (NHWC); the arg-max tap is kept (first maximum in scan order, as torch).
// A thread owns 4 adjacent channels (8 / 16-byte accesses, one uchar4 of tap indices); C % 4 == 0.
template <typename AT>
__global__ __launch_bounds__(256) void maxpool2d_fwd_kernel(const AT* __restrict__ x, AT* __restrict__ y, uint8_t* __restrict__ idx,
                                                            int N, int H, int W, int C, int Ho, int Wo) {
  const int cv = C >> 2;
  const long long total = (long long)N * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4; long long t = i / cv;
    const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); const int n = (int)(t / Ho);
    float best[4] = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const float4 q = ld4f(x + (((size_t)n * H + ih) * W + iw) * C + c);
          const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) if (v[j] > best[j]) { best[j] = v[j]; bi[j] = kh * 3 + kw; }
        }
      }
    const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c;
    st4f(y + o, make_float4(best[0], best[1], best[2], best[3]));
    *reinterpret_cast<uchar4*>(idx + o) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
  }
}
// gather form (no atomics, every input position written once): an input (ih, iw) belongs to the windows oh = (ih+1-kh)/2 with
// kh of the parity of ih+1 (same along w) -> at most 2 x 2 candidate outputs, each contributing when its arg-max tap is this position
template <typename AT>
__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const AT* __restrict__ dy, const uint8_t* __restrict__ idx, AT* __restrict__ dx,
                                                            int N, int H, int W, int C, int Ho, int Wo) {
  const int cv = C >> 2;
  const long long total = (long long)N * H * W * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4; long long t = i / cv;
    const int iw = (int)(t % W); t /= W; const int ih = (int)(t % H); const int n = (int)(t / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kh = (ih + 1) & 1; kh < 3; kh += 2) {
      const int oh = (ih + 1 - kh) >> 1;
      if (ih + 1 - kh < 0 || oh >= Ho) continue;
      for (int kw = (iw + 1) & 1; kw < 3; kw += 2) {
        const int ow = (iw + 1 - kw) >> 1;
        if (iw + 1 - kw < 0 || ow >= Wo) continue;
        const size_t o = (((size_t)n * Ho + oh) * Wo + ow) * C + c;
        const uchar4 id = *reinterpret_cast<const uchar4*>(idx + o);
        const float4 g = ld4f(dy + o);
        const int tap = kh * 3 + kw;
        if (id.x == tap) acc[0] += g.x;
        if (id.y == tap) acc[1] += g.y;
        if (id.z == tap) acc[2] += g.z;
        if (id.w == tap) acc[3] += g.w;
      }
    }
    st4f(dx + (((size_t)n * H + ih) * W + iw) * C + c, make_float4(acc[0], acc[1], acc[2], acc[3]));
  }
}

// ---- 2x2 average pool (NHWC), output may be a column slice of a wider buffer
template <typename AT>
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const AT* __restrict__ x, AT* __restrict__ y, int N, int H, int W, int C,
                                                           int ldy, int col_off) {
  const int Ho = H / 2, Wo = W / 2, cv = C / 4;
  const long long total = (long long)N * Ho * Wo * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4; long long t = i / cv;
    const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); const int n = (int)(t / Ho);
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = ld4f(x + (((size_t)n * H + oh * 2 + (k >> 1)) * W + ow * 2 + (k & 1)) * C + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= 0.25f; s.y *= 0.25f; s.z *= 0.25f; s.w *= 0.25f;
    st4f(y + (((size_t)n * Ho + oh) * Wo + ow) * ldy + col_off + c, s);
  }
}
template <typename AT>
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const AT* __restrict__ dy, AT* __restrict__ dx, int N, int H, int W, int C,
                                                           int ldy, int col_off) {
  const int Ho = H / 2, Wo = W / 2, cv = C / 4;
  const long long total = (long long)N * H * W * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * 4; long long t = i / cv;
    const int w = (int)(t % W); t /= W; const int h = (int)(t % H); const int n = (int)(t / H);
    float4 v = make_float4(0, 0, 0, 0);
    if (h / 2 < Ho && w / 2 < Wo) {
      v = ld4f(dy + (((size_t)n * Ho + h / 2) * Wo + w / 2) * ldy + col_off + c);
      v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f;
    }
    st4f(dx + (((size_t)n * H + h) * W + w) * C + c, v);
  }
}

// ---- decoder seed: AdaptiveAvgPool2d 7->2 (bins [0:4],[3:7]) + depth replication -> [I,2,2,2,C]   (decoder.py:17,59-67)
template <typename AT>
__global__ __launch_bounds__(256) void decoder_seed_fwd_kernel(const AT* __restrict__ f, AT* __restrict__ out, int I, int C) {
  const long long total = (long long)I * 4 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int bx = (int)(t % 2); t /= 2; const int by = (int)(t % 2); const int n = (int)(t / 2);
    float s = 0.f;
    for (int r = by * 3; r < by * 3 + 4; ++r)
      for (int q = bx * 3; q < bx * 3 + 4; ++q) s += ldf(f + (((size_t)n * 7 + r) * 7 + q) * C + c);
    s *= (1.f / 16.f);
    stf(out + ((((size_t)n * 2 + 0) * 2 + by) * 2 + bx) * C + c, s);
    stf(out + ((((size_t)n * 2 + 1) * 2 + by) * 2 + bx) * C + c, s);
  }
}
template <typename AT>
__global__ __launch_bounds__(256) void decoder_seed_bwd_kernel(const AT* __restrict__ dout, AT* __restrict__ df, int I, int C) {
  const long long total = (long long)I * 49 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int q = (int)(t % 7); t /= 7; const int r = (int)(t % 7); const int n = (int)(t / 7);
    float s = 0.f;
    for (int by = 0; by < 2; ++by)
      for (int bx = 0; bx < 2; ++bx)
        if (r >= by * 3 && r < by * 3 + 4 && q >= bx * 3 && q < bx * 3 + 4)
          s += ldf(dout + ((((size_t)n * 2 + 0) * 2 + by) * 2 + bx) * C + c) + ldf(dout + ((((size_t)n * 2 + 1) * 2 + by) * 2 + bx) * C + c);
    stf(df + i, s * (1.f / 16.f));
  }
}

// ---- MaxPool3d(2) with floor (33->16, 17->8, 9->4) on [N,D,H,W,C]
template <typename AT>
__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(const AT* __restrict__ x, AT* __restrict__ y, uint8_t* __restrict__ idx,
                                                            int N, int D, int H, int W, int C) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long total = (long long)N * Do * Ho * Wo * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); t /= Ho; const int od = (int)(t % Do); const int n = (int)(t / Do);
    float best = -3.4e38f; int bi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = ldf(x + ((((size_t)n * D + od * 2 + (k >> 2)) * H + oh * 2 + ((k >> 1) & 1)) * W + ow * 2 + (k & 1)) * C + c);
      if (v > best) { best = v; bi = k; }
    }
    stf(y + i, best); idx[i] = (uint8_t)bi;
  }
}
template <typename AT>
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(const AT* __restrict__ dy, const uint8_t* __restrict__ idx, AT* __restrict__ dx,
                                                            int N, int D, int H, int W, int C) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long long total = (long long)N * D * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int w = (int)(t % W); t /= W; const int h = (int)(t % H); t /= H; const int d = (int)(t % D); const int n = (int)(t / D);
    float v = 0.f;
    const int od = d / 2, oh = h / 2, ow = w / 2;
    if (od < Do && oh < Ho && ow < Wo) {
      const size_t o = ((((size_t)n * Do + od) * Ho + oh) * Wo + ow) * C + c;
      const int k = ((d & 1) << 2) | ((h & 1) << 1) | (w & 1);
      if (idx[o] == k) v = ldf(dy + o);
    }
    stf(dx + i, v);
  }
}

// ---- dropout (same kernel forward and backward: y = x * mask / keep, mask from (seed, element index))
template <typename AT>
__global__ __launch_bounds__(256) void dropout_kernel(const AT* __restrict__ x, AT* __restrict__ y, long long n, float p, uint32_t seed,
                                                      const uint32_t* __restrict__ epoch) {
  seed = eff_seed(seed, epoch);
  const float keep_inv = 1.f / (1.f - p);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    stf(y + i, uniform01(seed, (uint64_t)i) < p ? 0.f : ldf(x + i) * keep_inv);
}
// per-image drop-path factors: scale[i] = 0 or 1/keep
__global__ void droppath_scale_kernel(float* __restrict__ scale, int I, float p, uint32_t seed, const uint32_t* __restrict__ epoch) {
  seed = eff_seed(seed, epoch);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < I) scale[i] = uniform01(seed, (uint64_t)i) < p ? 0.f : 1.f / (1.f - p);
}
// y[row, :] = x[row, :] * scale[row / rows_per_scale]
template <typename AT>
__global__ __launch_bounds__(256) void rowscale_kernel(const AT* __restrict__ x, const float* __restrict__ scale, AT* __restrict__ y,
                                                       long long rows, int C, int rows_per_scale) {
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    stf(y + i, ldf(x + i) * scale[(i / C) / rows_per_scale]);
}
template <typename AT>
__global__ __launch_bounds__(256) void rowscale_vec_kernel(const AT* __restrict__ x, const float* __restrict__ scale, AT* __restrict__ y,
                                                           long long rows, int C, int rows_per_scale) {
  const int cv = C >> 2;
  const long long total = rows * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float sc = scale[(i / cv) / rows_per_scale];
    float4 v = ld4f(x + i * 4);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    st4f(y + i * 4, v);
  }
}

// ---- cross-view attention spatial path (cross_view_attention.py:26-32,68 and :110-120)
// depthwise 2x2 stride-2 conv 7->3 (native weight [C,1,2,2])
template <typename AT>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const AT* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                         AT* __restrict__ y, int I, int C) {
  const long long total = (long long)I * 9 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int ox = (int)(t % 3); t /= 3; const int oy = (int)(t % 3); const int n = (int)(t / 3);
    float s = b ? b[c] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += ldf(x + (((size_t)n * 7 + oy * 2 + (k >> 1)) * 7 + ox * 2 + (k & 1)) * C + c) * w[c * 4 + k];
    stf(y + i, s);
  }
}
// dx (all 49 positions; row/col 6 receive 0)
template <typename AT>
__global__ __launch_bounds__(256) void dwconv_bwd_dx_kernel(const AT* __restrict__ dy, const float* __restrict__ w, AT* __restrict__ dx,
                                                            int I, int C) {
  const long long total = (long long)I * 49 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int xq = (int)(t % 7); t /= 7; const int yq = (int)(t % 7); const int n = (int)(t / 7);
    float v = 0.f;
    if (yq < 6 && xq < 6) v = ldf(dy + (((size_t)n * 3 + yq / 2) * 3 + xq / 2) * C + c) * w[c * 4 + (yq & 1) * 2 + (xq & 1)];
    stf(dx + i, v);
  }
}
// dw[c][k], db[c]: one thread per (c,k) walks a slice of the images (grid.y slices, one atomic per slice)
template <typename AT>
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const AT* __restrict__ dy, const AT* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ db, int I, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * 4) return;
  const int c = i >> 2, k = i & 3;
  float s = 0.f, sb = 0.f;
  const int per = (I + gridDim.y - 1) / gridDim.y;
  const int n0 = blockIdx.y * per, n1 = n0 + per < I ? n0 + per : I;
  for (int n = n0; n < n1; ++n)
    for (int o = 0; o < 9; ++o) {
      const int oy = o / 3, ox = o % 3;
      const float g = ldf(dy + ((size_t)n * 9 + o) * C + c);
      s += g * ldf(x + (((size_t)n * 7 + oy * 2 + (k >> 1)) * 7 + ox * 2 + (k & 1)) * C + c);
      sb += g;
    }
  atomicAdd(dw + i, s);
  if (k == 0 && db) atomicAdd(db + c, sb);
}
// bilinear 3->7 (align_corners=False) + residual: y = up(small) + x.  Separable taps per output index:
// src = (dst+0.5)*3/7-0.5 clamped at 0 -> i0 = floor, frac; taps {1,0,0},{6/7,1/7,0},{3/7,4/7,0},{0,1,0},{0,4/7,3/7},{0,1/7,6/7},{0,0,1}
__device__ __forceinline__ void up_taps(int o, int& i0, int& i1, float& w0, float& w1) {
  float src = (o + 0.5f) * (3.0f / 7.0f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src; i1 = i0 < 2 ? i0 + 1 : 2;
  w1 = src - (float)i0; w0 = 1.f - w1;
}
template <typename AT>
__global__ __launch_bounds__(256) void upsample_add_fwd_kernel(const AT* __restrict__ small, const AT* __restrict__ x, int ldx,
                                                               AT* __restrict__ y, int I, int C) {
  const long long total = (long long)I * 49 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int xq = (int)(t % 7); t /= 7; const int yq = (int)(t % 7); const int n = (int)(t / 7);
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    up_taps(yq, y0, y1, wy0, wy1); up_taps(xq, x0, x1, wx0, wx1);
    const AT* s = small + (size_t)n * 9 * C + c;
    const float v = wy0 * (wx0 * ldf(s + (y0 * 3 + x0) * C) + wx1 * ldf(s + (y0 * 3 + x1) * C)) +
                    wy1 * (wx0 * ldf(s + (y1 * 3 + x0) * C) + wx1 * ldf(s + (y1 * 3 + x1) * C));
    stf(y + i, v + ldf(x + (((size_t)n * 7 + yq) * 7 + xq) * ldx + c));
  }
}
template <typename AT>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const AT* __restrict__ dy, AT* __restrict__ dsmall, int I, int C) {
  const long long total = (long long)I * 9 * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C); long long t = i / C;
    const int sx = (int)(t % 3); t /= 3; const int sy = (int)(t % 3); const int n = (int)(t / 3);
    float acc = 0.f;
    for (int yq = 0; yq < 7; ++yq) {
      int y0, y1; float wy0, wy1; up_taps(yq, y0, y1, wy0, wy1);
      const float wy = (y0 == sy ? wy0 : 0.f) + (y1 == sy ? wy1 : 0.f);
      if (wy == 0.f) continue;
      for (int xq = 0; xq < 7; ++xq) {
        int x0, x1; float wx0, wx1; up_taps(xq, x0, x1, wx0, wx1);
        const float wx = (x0 == sx ? wx0 : 0.f) + (x1 == sx ? wx1 : 0.f);
        if (wx != 0.f) acc += wy * wx * ldf(dy + (((size_t)n * 7 + yq) * 7 + xq) * C + c);
      }
    }
    stf(dsmall + i, acc);
  }
}

// ---- decoder head: ConvTranspose3d(8,1,k=1) + cat -> channels-last [M,12] raw (9 used) and [M] logits (decoder.py:45,83-94)
template <typename AT>
__global__ __launch_bounds__(256) void decoder_head_fwd_kernel(const AT* __restrict__ x8, const float* __restrict__ w, const float* __restrict__ bias,
                                                               AT* __restrict__ raw, AT* __restrict__ vol, long long M) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long long)gridDim.x * 256) {
    const float4 a = ld4f(x8 + i * 8), b = ld4f(x8 + i * 8 + 4);
    float v = a.x * w[0] + a.y * w[1] + a.z * w[2] + a.w * w[3] + b.x * w[4] + b.y * w[5] + b.z * w[6] + b.w * w[7];
    if (bias) v += bias[0];
    st4f(raw + i * 12, a);
    st4f(raw + i * 12 + 4, b);
    st4f(raw + i * 12 + 8, make_float4(v, 0.f, 0.f, 0.f));
    stf(vol + i, v);
  }
}
// dx8[c] = draw[c] + w[c]*(draw[8] + dvol); dw[c] += sum x8[c]*(draw[8]+dvol); dbias += sum(draw[8]+dvol)
template <typename AT>
__global__ __launch_bounds__(256) void decoder_head_bwd_kernel(const AT* __restrict__ draw, const AT* __restrict__ dvol, const AT* __restrict__ x8,
                                                               const float* __restrict__ w, AT* __restrict__ dx8, float* __restrict__ dw,
                                                               float* __restrict__ dbias, long long M) {
  __shared__ float sc[4];
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long long)gridDim.x * 256) {
    const float4 ga = ld4f(draw + i * 12), gb = ld4f(draw + i * 12 + 4);
    const float g = ldf(draw + i * 12 + 8) + (dvol ? ldf(dvol + i) : 0.f);
    const float4 a = ld4f(x8 + i * 8), b = ld4f(x8 + i * 8 + 4);
    st4f(dx8 + i * 8, make_float4(ga.x + w[0] * g, ga.y + w[1] * g, ga.z + w[2] * g, ga.w + w[3] * g));
    st4f(dx8 + i * 8 + 4, make_float4(gb.x + w[4] * g, gb.y + w[5] * g, gb.z + w[6] * g, gb.w + w[7] * g));
    acc[0] += a.x * g; acc[1] += a.y * g; acc[2] += a.z * g; acc[3] += a.w * g;
    acc[4] += b.x * g; acc[5] += b.y * g; acc[6] += b.z * g; acc[7] += b.w * g; acc[8] += g;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float s = block_sum<4>(acc[k], sc);
    if (threadIdx.x == 0) { if (k < 8) atomicAdd(dw + k, s); else if (dbias) atomicAdd(dbias, s); }
  }
}

// ---- merger tail: softmax over the V views per voxel, weighted sum of the coarse volumes (merger.py:91-104)
template <typename AT>
__global__ __launch_bounds__(256) void merge_views_fwd_kernel(const AT* __restrict__ wl, const AT* __restrict__ vol, AT* __restrict__ out,
                                                              int B, int V, int S) {
  const long long total = (long long)B * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int s = (int)(i % S); const long long b = i / S;
    const AT* wp = wl + (size_t)b * V * S + s; const AT* vp = vol + (size_t)b * V * S + s;
    float mx = -3.4e38f;
    for (int v = 0; v < V; ++v) mx = fmaxf(mx, ldf(wp + (size_t)v * S));
    float den = 0.f, num = 0.f;
    for (int v = 0; v < V; ++v) { const float e = expf(ldf(wp + (size_t)v * S) - mx); den += e; num += e * ldf(vp + (size_t)v * S); }
    stf(out + i, num / den);
  }
}
template <typename AT>
__global__ __launch_bounds__(256) void merge_views_bwd_kernel(const AT* __restrict__ wl, const AT* __restrict__ vol, const AT* __restrict__ out,
                                                              const AT* __restrict__ dout, AT* __restrict__ dwl, AT* __restrict__ dvol,
                                                              int B, int V, int S) {
  const long long total = (long long)B * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int s = (int)(i % S); const long long b = i / S;
    const size_t base = (size_t)b * V * S + s;
    float mx = -3.4e38f;
    for (int v = 0; v < V; ++v) mx = fmaxf(mx, ldf(wl + base + (size_t)v * S));
    float den = 0.f;
    for (int v = 0; v < V; ++v) den += expf(ldf(wl + base + (size_t)v * S) - mx);
    const float g = ldf(dout + i), o = ldf(out + i), inv = 1.f / den;
    for (int v = 0; v < V; ++v) {
      const float pv = expf(ldf(wl + base + (size_t)v * S) - mx) * inv;
      stf(dvol + base + (size_t)v * S, g * pv);
      stf(dwl + base + (size_t)v * S, pv * g * (ldf(vol + base + (size_t)v * S) - o));
    }
  }
}
// mean over views (USE_MERGER off): out = mean_v vol
__global__ __launch_bounds__(256) void mean_views_kernel(const float* __restrict__ vol, float* __restrict__ out, int B, int V, int S) {
  const long long total = (long long)B * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int s = (int)(i % S); const long long b = i / S;
    float a = 0.f;
    for (int v = 0; v < V; ++v) a += vol[((size_t)b * V + v) * S + s];
    out[i] = a / V;
  }
}

// ---- storage conversion at the module boundaries (nn.Module inputs/outputs are fp32 torch tensors)
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long n) {
  const long long n4 = n >> 2;     // 4 elements per thread, scalar tail
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) st4f(dst + i * 4, ld4f(src + i * 4));
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = (TD)(float)src[i];
}

// ---- BCE-with-logits (mean) forward + gradient in one pass (core/train.py:165,249,255)
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n, float* __restrict__ loss,
                                                  float* __restrict__ dx, const float* __restrict__ gscale) {
  __shared__ float sc[4];
  const float g = (gscale ? gscale[0] : 1.f) / (float)n;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float xv = x[i], tv = t[i];
    const float e = expf(-fabsf(xv));
    acc += fmaxf(xv, 0.f) - xv * tv + log1pf(e);
    if (dx) {
      const float sig = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      dx[i] = (sig - tv) * g;
    }
  }
  acc = block_sum<4>(acc, sc);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, acc / (float)n);
}

// ---- thresholded-occupancy counters for IoU and F-score (core/test.py:141-163), every threshold from one pass:
//      counts[b][th] = {intersection (= true positives), union, false positives, false negatives}
constexpr int IOU_MAX_TH = 8;
__global__ __launch_bounds__(256) void iou_counts_kernel(const float* __restrict__ logits, const float* __restrict__ gt, const float* __restrict__ ths,
                                                         int nth, int S, float* __restrict__ counts) {
  __shared__ float sc[4];
  const int b = blockIdx.x;
  float th[IOU_MAX_TH], acc[IOU_MAX_TH][4];
#pragma unroll
  for (int k = 0; k < IOU_MAX_TH; ++k) {
    th[k] = k < nth ? ths[k] : 2.f;
    acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
  }
  for (int s = threadIdx.x; s < S; s += 256) {
    const float xv = logits[(size_t)b * S + s], gv = gt[(size_t)b * S + s];
    const float pr = 1.f / (1.f + expf(-xv));
#pragma unroll
    for (int k = 0; k < IOU_MAX_TH; ++k) {
      const float v = pr >= th[k] ? 1.f : 0.f;
      acc[k][0] += v * gv;
      acc[k][1] += (v + gv >= 1.f) ? 1.f : 0.f;
      acc[k][2] += v * (1.f - gv);
      acc[k][3] += (1.f - v) * gv;
    }
  }
#pragma unroll
  for (int k = 0; k < IOU_MAX_TH; ++k) {
    if (k >= nth) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float r = block_sum<4>(acc[k][j], sc);
      if (threadIdx.x == 0) counts[((size_t)b * nth + k) * 4 + j] = r;
    }
  }
}

}  // namespace sv

using namespace sv;
#define STREAM ((hipStream_t)stream)
#define CA(p) static_cast<const AT*>(p)
#define MA(p) static_cast<AT*>(p)

// ---- layout kernels of the ResNet stem in its 4x4 / stride-1 formulation on the space-to-depth image (models/encoder.py) and of the
//      merger's 16-channel stencil weights: the re-indexing that used to be torch index ops on the path
namespace sv {
// ---- refiner head: Conv3d(1 -> Co, k = 4, p = 2) on a D^3 grid (reference models/refiner.py:21-26) as a (4, 1, 1)-tap convolution over 16
// channels - the trick of the ResNet stem: xc[n][z][Y][X][(cy, cx)] = x[n][z][Y + cy - 2][X + cx - 2] (zero outside), Y, X in [0, D + 1), so
// that  y[n][oz][oy][ox][co] = sum_{kz, cy, cx} xc[n][oz + kz - 2][oy][ox][(cy, cx)] w[co][kz][cy][cx]:  the one-channel layer leaves the
// engine's scalar gather path (Ci = 1: 375 / 778 / 410 us for forward / data gradient / weight gradient at 64 samples, the data gradient on
// a 16-wide tile with ONE real column) for 32-byte channel vectors and a 16-column data gradient.  One thread per (position, cy): 4 outputs.
template <typename AT>
__global__ __launch_bounds__(256) void head_pack_x_kernel(const AT* __restrict__ x, AT* __restrict__ xc, int D, long long n) {
  const int P = D + 1;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const int cy = (int)(t & 3); long long r = t >> 2;
    const int X = (int)(r % P); r /= P;
    const int Y = (int)(r % P); r /= P;           // r = n * D + z
    const int yy = Y + cy - 2;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)yy < (unsigned)D) {
      const AT* src = x + (r * D + yy) * D;
#pragma unroll
      for (int cx = 0; cx < 4; ++cx) { const int xx = X + cx - 2; if ((unsigned)xx < (unsigned)D) v[cx] = ldf(src + xx); }
    }
    st4f(xc + ((r * P + Y) * P + X) * 16 + cy * 4, make_float4(v[0], v[1], v[2], v[3]));
  }
}
// the data gradient back: dx[n][z][y][x] = sum_{cy, cx} dxc[n][z][y - cy + 2][x - cx + 2][(cy, cx)] (every (Y, X) that read this voxel)
template <typename AT>
__global__ __launch_bounds__(256) void head_unpack_dx_kernel(const AT* __restrict__ dxc, AT* __restrict__ dx, int D, long long n) {
  const int P = D + 1;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const int xq = (int)(t % D); long long r = t / D;
    const int yq = (int)(r % D); r /= D;          // r = n * D + z
    float a = 0.f;
#pragma unroll
    for (int cy = 0; cy < 4; ++cy) {
      const int Y = yq - cy + 2;                  // -1 ... D + 1
      if ((unsigned)Y >= (unsigned)P) continue;
#pragma unroll
      for (int cx = 0; cx < 4; ++cx) {
        const int X = xq - cx + 2;
        if ((unsigned)X < (unsigned)P) a += ldf(dxc + ((r * P + Y) * P + X) * 16 + cy * 4 + cx);
      }
    }
    stf(dx + t, a);
  }
}

// x16[i][oy][ox][(sy, sx, c)] = img[i][c][2 oy + sy][2 ox + sx] (c < 3), 0 for the pad channel c = 3; one thread per (i, oy, ox, sy): 8 outputs
template <typename AT>
__global__ __launch_bounds__(256) void stem_s2d_kernel(const AT* __restrict__ img, AT* __restrict__ x16, long long n) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const int sy = (int)(t & 1); long long r = t >> 1;
    const int ox = (int)(r % 112); r /= 112;
    const int oy = (int)(r % 112); const long long i = r / 112;
    const AT* src = img + (i * 3) * 224 * 224 + (size_t)(2 * oy + sy) * 224 + 2 * ox;
    AT* dst = x16 + ((i * 112 + oy) * 112 + ox) * 16 + sy * 8;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[sx * 4 + c] = src[(size_t)c * 224 * 224 + sx];
      stf(dst + sx * 4 + 3, 0.f);
    }
  }
}
// ---- encoder input in one pass: the NCHW renderings (fp32 as the module receives them, or the storage type) -> both backbones' first operands
//   x16[i][oy][ox][(sy, sx, c)] = img[i][c][2 oy + sy][2 ox + sx]  (c < 3, 0 for c = 3)   the stem's space-to-depth image (stem_s2d_kernel)
//   xp [i][py][px][(ky, kx, c)] = img[i][c][4 py + ky][4 px + kx]                           the Swin patch rows: PatchEmbed's Conv2d(3, C, 4, 4)
//                                                                                            (timm, behind models/swin_transformer.py:78) is a Linear(48, C) on them
// Both are rearrangements of the same 4 x 4 x 3 pixel block: one thread per block reads it once (12 row pieces of 4 pixels) and writes the 48
// patch values and the four 16-value space-to-depth rows.  Replaces cast (fp32 -> storage) + NCHW -> NHWC transpose + space-to-depth, and the
// 3-channel gathers of the patch-embedding convolution (6-byte taps on the engine's scalar path: 47 TFLOP/s forward, 53 weight gradient).
template <typename IT, typename AT>
__global__ __launch_bounds__(256) void encoder_prep_kernel(const IT* __restrict__ img, AT* __restrict__ x16, AT* __restrict__ xp, int S, long long n) {
  const int P = S >> 2, Q = S >> 1;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const int px = (int)(t % P); long long r = t / P;
    const int py = (int)(r % P); const long long i = r / P;
    const IT* src = img + (i * 3) * S * S + (size_t)(4 * py) * S + 4 * px;
    float v[3][4][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        const float4 q = ld4f(src + ((size_t)c * S + ky) * S);
        v[c][ky][0] = q.x; v[c][ky][1] = q.y; v[c][ky][2] = q.z; v[c][ky][3] = q.w;
      }
    AT* dp = xp + ((i * P + py) * P + px) * 48;
#pragma unroll
    for (int g = 0; g < 12; ++g) {                     // 4 consecutive outputs (ky, kx, c) = 4 g .. 4 g + 3
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int e = 4 * g + j, ky = e / 12, kx = (e / 3) % 4, c = e % 3; o[j] = v[c][ky][kx]; }
      st4f(dp + 4 * g, make_float4(o[0], o[1], o[2], o[3]));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        AT* ds = x16 + ((i * Q + 2 * py + a) * Q + 2 * px + b) * 16;
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
#pragma unroll
          for (int sx = 0; sx < 2; ++sx)
            st4f(ds + (sy * 2 + sx) * 4, make_float4(v[0][2 * a + sy][2 * b + sx], v[1][2 * a + sy][2 * b + sx], v[2][2 * a + sy][2 * b + sx], 0.f));
      }
  }
}
// wp[co][(ty, tx)][(sy, sx, c)] = w[co][c][2 ty + sy - 1][2 tx + sx - 1] inside the 7x7 kernel and c < 3, else 0
template <typename WT>
__global__ void stem_pack_kernel(const float* __restrict__ w, WT* __restrict__ wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * 256) return;
  const int co = idx >> 8, tap = (idx >> 4) & 15, s = idx & 15;
  const int ty = tap >> 2, tx = tap & 3, sy = s >> 3, sx = (s >> 2) & 1, c = s & 3;
  const int ky = 2 * ty + sy - 1, kx = 2 * tx + sx - 1;
  const bool ok = c < 3 && ky >= 0 && ky < 7 && kx >= 0 && kx < 7;
  wp[idx] = (WT)(ok ? w[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f);
}
// dw[co][c][ky][kx] += dw16[co][(sy, sx, c)][ty][tx]  (native weight-gradient layout of the 4x4 formulation)
__global__ void stem_unpack_grad_kernel(const float* __restrict__ dw16, float* __restrict__ dw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * 3 * 49) return;
  const int kx = idx % 7, ky = (idx / 7) % 7, c = (idx / 49) % 3, co = idx / 147;
  const int ty = (ky + 1) >> 1, sy = (ky + 1) & 1, tx = (kx + 1) >> 1, sx = (kx + 1) & 1;
  dw[idx] += dw16[((co * 16 + (sy * 8 + sx * 4 + c)) * 4 + ty) * 4 + tx];
}
// merger.py:20-54 weights [cout][cin][27] -> bf16 stencil packs: forward [16][27][ncols] with wp[co][tap][col(ci)] = w[co][ci][tap];
// data-gradient [ncols][27][16] with wp[col(ci)][26 - tap][co] = w[co][ci][tap]; col(ci) = 12 (ci / 9) + ci % 9 for the 36-channel concat
__global__ void merger_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int cout, int cin, int ncols, int dgrad, int cat) {
  const int total = 16 * 27 * ncols;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int co, tap, col;
    if (!dgrad) { col = idx % ncols; tap = (idx / ncols) % 27; co = idx / (27 * ncols); }
    else { co = idx % 16; tap = 26 - (idx / 16) % 27; col = idx / (16 * 27); }
    const int ci = cat ? ((col % 12) < 9 ? (col / 12) * 9 + col % 12 : -1) : col;
    const bool ok = co < cout && ci >= 0 && ci < cin;
    wp[idx] = (__bf16)(ok ? w[((size_t)co * cin + ci) * 27 + tap] : 0.f);
  }
}
}  // namespace sv
using namespace sv;

extern "C" int sv_stem_space_to_depth(const void* images, void* x16, int I, int act_dtype, void* stream) {
  SV_REQUIRE(images && x16 && I > 0, "stem_space_to_depth: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  const long long n = (long long)I * 112 * 112 * 2;
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(stem_s2d_kernel<AT>, dim3(grid_for(n)), dim3(256), 0, STREAM, CA(images), MA(x16), n););
  return check_launch("sv_stem_space_to_depth");
}
extern "C" int sv_encoder_prep(const void* images, int images_f32, void* x16, void* xp, int I, int S, int act_dtype, void* stream) {
  SV_REQUIRE(images && x16 && xp && I > 0 && S >= 4 && S % 4 == 0, "encoder_prep: bad arguments (I=%d S=%d)", I, S);
  SV_REQUIRE_ACT(act_dtype);
  SV_REQUIRE((((uintptr_t)images | (uintptr_t)x16 | (uintptr_t)xp) & 15) == 0, "encoder_prep: buffers must be 16-byte aligned");
  const long long n = (long long)I * (S / 4) * (S / 4);
  if (images_f32) {
    SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL((encoder_prep_kernel<float, AT>), dim3(grid_for(n)), dim3(256), 0, STREAM, static_cast<const float*>(images), MA(x16), MA(xp), S, n););
  } else {
    SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL((encoder_prep_kernel<AT, AT>), dim3(grid_for(n)), dim3(256), 0, STREAM, CA(images), MA(x16), MA(xp), S, n););
  }
  return check_launch("sv_encoder_prep");
}
extern "C" int sv_head_pack_x(const void* x, void* xc, int N, int D, int act_dtype, void* stream) {
  SV_REQUIRE(x && xc && N > 0 && D > 0, "head_pack_x: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  const long long n = (long long)N * D * (D + 1) * (D + 1) * 4;
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(head_pack_x_kernel<AT>, dim3(grid_for(n)), dim3(256), 0, STREAM, CA(x), MA(xc), D, n););
  return check_launch("sv_head_pack_x");
}
extern "C" int sv_head_unpack_dx(const void* dxc, void* dx, int N, int D, int act_dtype, void* stream) {
  SV_REQUIRE(dxc && dx && N > 0 && D > 0, "head_unpack_dx: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  const long long n = (long long)N * D * D * D;
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(head_unpack_dx_kernel<AT>, dim3(grid_for(n)), dim3(256), 0, STREAM, CA(dxc), MA(dx), D, n););
  return check_launch("sv_head_unpack_dx");
}
extern "C" int sv_stem_pack(const float* w, void* wp, int out_dtype, void* stream) {
  SV_REQUIRE(w && wp, "stem_pack: bad arguments");
  SV_REQUIRE_ACT(out_dtype);
  if (out_dtype == SV_BF16) hipLaunchKernelGGL(stem_pack_kernel<__bf16>, dim3(64), dim3(256), 0, STREAM, w, static_cast<__bf16*>(wp));
  else hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(64), dim3(256), 0, STREAM, w, static_cast<float*>(wp));
  return check_launch("sv_stem_pack");
}
extern "C" int sv_stem_unpack_grad(const float* dw16, float* dw, void* stream) {
  SV_REQUIRE(dw16 && dw, "stem_unpack_grad: bad arguments");
  hipLaunchKernelGGL(stem_unpack_grad_kernel, dim3(cdiv(64 * 147, 256)), dim3(256), 0, STREAM, dw16, dw);
  return check_launch("sv_stem_unpack_grad");
}
extern "C" int sv_merger_pack(const float* w, void* wp_bf16, int cout, int cin, int dgrad, int concat, void* stream) {
  SV_REQUIRE(w && wp_bf16 && cout > 0 && cout <= 16 && cin > 0 && (concat ? cin == 36 : cin <= 16), "merger_pack: bad arguments (cout=%d cin=%d)", cout, cin);
  const int ncols = concat ? 48 : 16;
  hipLaunchKernelGGL(merger_pack_kernel, dim3(cdiv(16 * 27 * ncols, 256)), dim3(256), 0, STREAM, w, static_cast<__bf16*>(wp_bf16), cout, cin, ncols, dgrad ? 1 : 0, concat ? 1 : 0);
  return check_launch("sv_merger_pack");
}

extern "C" int sv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream) {
  SV_REQUIRE(src && dst && n > 0, "cast: bad arguments");
  SV_REQUIRE_ACT(src_dtype); SV_REQUIRE_ACT(dst_dtype);
  SV_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "cast: buffers must be 16-byte aligned");
  const dim3 g(grid_for((n + 3) / 4));
  if (src_dtype == SV_F32 && dst_dtype == SV_BF16) hipLaunchKernelGGL((cast_kernel<float, __bf16>), g, dim3(256), 0, STREAM, (const float*)src, (__bf16*)dst, n);
  else if (src_dtype == SV_BF16 && dst_dtype == SV_F32) hipLaunchKernelGGL((cast_kernel<__bf16, float>), g, dim3(256), 0, STREAM, (const __bf16*)src, (float*)dst, n);
  else if (src_dtype == SV_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, dim3(256), 0, STREAM, (const float*)src, (float*)dst, n);
  else hipLaunchKernelGGL((cast_kernel<__bf16, __bf16>), g, dim3(256), 0, STREAM, (const __bf16*)src, (__bf16*)dst, n);
  return check_launch("sv_cast");
}
extern "C" int sv_transpose(const void* src, void* dst, int batch, int R, int C, int lds, int ldd, long long src_bstride,
                            long long dst_bstride, int act_dtype, void* stream) {
  SV_REQUIRE(src && dst && batch > 0 && R > 0 && C > 0 && lds >= C && ldd >= R, "transpose: bad arguments");
  SV_REQUIRE(batch <= 65535, "transpose: batch %d too large", batch);
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(transpose_kernel<AT>, dim3(cdiv(C, 32), cdiv(R, 32), batch), dim3(256), 0, STREAM, CA(src), MA(dst), R, C, lds, ldd,
                                                src_bstride, dst_bstride););
  return check_launch("sv_transpose");
}
extern "C" int sv_add_n(const void* a, const void* b, const void* c, const void* d, void* out, long long M, int C, int ldo, int act_dtype, void* stream) {
  SV_REQUIRE(a && b && out && M > 0 && C % 4 == 0 && ldo % 4 == 0 && ldo >= C, "add_n: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(add_n_kernel<AT>, dim3(grid_for(M * (C / 4))), dim3(256), 0, STREAM, CA(a), CA(b), CA(c), CA(d), MA(out), M, C, ldo););
  return check_launch("sv_add_n");
}
extern "C" int sv_axpby(const void* a, const void* b, void* out, float alpha, float beta, long long n, int act_dtype, void* stream) {
  SV_REQUIRE(a && out && n > 0, "axpby: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  const bool vec = n % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
  SV_DISPATCH_ACT(act_dtype,
    if (vec) hipLaunchKernelGGL(axpby_vec_kernel<AT>, dim3(grid_for(n / 4)), dim3(256), 0, STREAM, CA(a), CA(b), MA(out), alpha, beta, n / 4);
    else hipLaunchKernelGGL(axpby_kernel<AT>, dim3(grid_for(n)), dim3(256), 0, STREAM, CA(a), CA(b), MA(out), alpha, beta, n););
  return check_launch("sv_axpby");
}
extern "C" int sv_relu_bwd(const void* dy, const void* y, void* out, long long n, int act_dtype, void* stream) {
  SV_REQUIRE(dy && y && out && n > 0, "relu_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(relu_bwd_kernel<AT>, dim3(grid_for(n)), dim3(256), 0, STREAM, CA(dy), CA(y), MA(out), n););
  return check_launch("sv_relu_bwd");
}
extern "C" int sv_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int act_dtype, void* stream) {
  SV_REQUIRE(x && y && idx && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "maxpool2d_fwd: bad arguments (C must be a multiple of 4)");
  SV_REQUIRE_ACT(act_dtype);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(maxpool2d_fwd_kernel<AT>, dim3(grid_for((long long)N * Ho * Wo * (C / 4))), dim3(256), 0, STREAM, CA(x), MA(y), idx, N, H, W, C, Ho, Wo););
  return check_launch("sv_maxpool2d_fwd");
}
extern "C" int sv_maxpool2d_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int act_dtype, void* stream) {
  SV_REQUIRE(dy && idx && dx && N > 0 && C % 4 == 0, "maxpool2d_bwd: bad arguments (C must be a multiple of 4)");
  SV_REQUIRE_ACT(act_dtype);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(maxpool2d_bwd_kernel<AT>, dim3(grid_for((long long)N * H * W * (C / 4))), dim3(256), 0, STREAM, CA(dy), idx, MA(dx), N, H, W, C, Ho, Wo););
  return check_launch("sv_maxpool2d_bwd");
}
extern "C" int sv_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int ldy, int col_off, int act_dtype, void* stream) {
  SV_REQUIRE(x && y && N > 0 && C % 4 == 0 && ldy % 4 == 0 && col_off % 4 == 0 && ldy >= col_off + C, "avgpool2_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(avgpool2_fwd_kernel<AT>, dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, STREAM, CA(x), MA(y), N, H, W, C, ldy, col_off););
  return check_launch("sv_avgpool2_fwd");
}
extern "C" int sv_avgpool2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int ldy, int col_off, int act_dtype, void* stream) {
  SV_REQUIRE(dy && dx && N > 0 && C % 4 == 0 && ldy % 4 == 0 && col_off % 4 == 0, "avgpool2_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(avgpool2_bwd_kernel<AT>, dim3(grid_for((long long)N * H * W * (C / 4))), dim3(256), 0, STREAM, CA(dy), MA(dx), N, H, W, C, ldy, col_off););
  return check_launch("sv_avgpool2_bwd");
}
extern "C" int sv_decoder_seed_fwd(const void* feat, void* out, int I, int C, int act_dtype, void* stream) {
  SV_REQUIRE(feat && out && I > 0 && C > 0, "decoder_seed_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(decoder_seed_fwd_kernel<AT>, dim3(grid_for((long long)I * 4 * C)), dim3(256), 0, STREAM, CA(feat), MA(out), I, C););
  return check_launch("sv_decoder_seed_fwd");
}
extern "C" int sv_decoder_seed_bwd(const void* dout, void* dfeat, int I, int C, int act_dtype, void* stream) {
  SV_REQUIRE(dout && dfeat && I > 0 && C > 0, "decoder_seed_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(decoder_seed_bwd_kernel<AT>, dim3(grid_for((long long)I * 49 * C)), dim3(256), 0, STREAM, CA(dout), MA(dfeat), I, C););
  return check_launch("sv_decoder_seed_bwd");
}
extern "C" int sv_maxpool3d_fwd(const void* x, void* y, uint8_t* idx, int N, int D, int H, int W, int C, int act_dtype, void* stream) {
  SV_REQUIRE(x && y && idx && N > 0 && D > 1 && H > 1 && W > 1 && C > 0, "maxpool3d_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(maxpool3d_fwd_kernel<AT>, dim3(grid_for((long long)N * (D / 2) * (H / 2) * (W / 2) * C)), dim3(256), 0, STREAM, CA(x), MA(y), idx, N, D, H, W, C););
  return check_launch("sv_maxpool3d_fwd");
}
extern "C" int sv_maxpool3d_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int D, int H, int W, int C, int act_dtype, void* stream) {
  SV_REQUIRE(dy && idx && dx && N > 0, "maxpool3d_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(maxpool3d_bwd_kernel<AT>, dim3(grid_for((long long)N * D * H * W * C)), dim3(256), 0, STREAM, CA(dy), idx, MA(dx), N, D, H, W, C););
  return check_launch("sv_maxpool3d_bwd");
}
extern "C" int sv_dropout(const void* x, void* y, long long n, float p, uint32_t seed, const uint32_t* seed_epoch, int act_dtype, void* stream) {
  SV_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(dropout_kernel<AT>, dim3(grid_for(n)), dim3(256), 0, STREAM, CA(x), MA(y), n, p, seed, seed_epoch););
  return check_launch("sv_dropout");
}
extern "C" int sv_droppath_scale(float* scale, int I, float p, uint32_t seed, const uint32_t* seed_epoch, void* stream) {
  SV_REQUIRE(scale && I > 0 && p >= 0.f && p < 1.f, "droppath_scale: bad arguments");
  hipLaunchKernelGGL(droppath_scale_kernel, dim3(cdiv(I, 64)), dim3(64), 0, STREAM, scale, I, p, seed, seed_epoch);
  return check_launch("sv_droppath_scale");
}
extern "C" int sv_rowscale(const void* x, const float* scale, void* y, long long rows, int C, int rows_per_scale, int act_dtype, void* stream) {
  SV_REQUIRE(x && scale && y && rows > 0 && C > 0 && rows_per_scale > 0, "rowscale: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  const bool vec = C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  SV_DISPATCH_ACT(act_dtype,
    if (vec) hipLaunchKernelGGL(rowscale_vec_kernel<AT>, dim3(grid_for(rows * (C / 4))), dim3(256), 0, STREAM, CA(x), scale, MA(y), rows, C, rows_per_scale);
    else hipLaunchKernelGGL(rowscale_kernel<AT>, dim3(grid_for(rows * C)), dim3(256), 0, STREAM, CA(x), scale, MA(y), rows, C, rows_per_scale););
  return check_launch("sv_rowscale");
}
extern "C" int sv_dwconv2x2_fwd(const void* x, const float* w, const float* b, void* y, int I, int C, int act_dtype, void* stream) {
  SV_REQUIRE(x && w && y && I > 0 && C > 0, "dwconv2x2_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(dwconv_fwd_kernel<AT>, dim3(grid_for((long long)I * 9 * C)), dim3(256), 0, STREAM, CA(x), w, b, MA(y), I, C););
  return check_launch("sv_dwconv2x2_fwd");
}
extern "C" int sv_dwconv2x2_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, int I, int C, int act_dtype, void* stream) {
  SV_REQUIRE(dy && x && w && dx && dw && I > 0 && C > 0, "dwconv2x2_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype,
    hipLaunchKernelGGL(dwconv_bwd_dx_kernel<AT>, dim3(grid_for((long long)I * 49 * C)), dim3(256), 0, STREAM, CA(dy), w, MA(dx), I, C);
    hipLaunchKernelGGL(dwconv_bwd_w_kernel<AT>, dim3(cdiv(C * 4, 256), I < 32 ? I : 32), dim3(256), 0, STREAM, CA(dy), CA(x), dw, db, I, C););
  return check_launch("sv_dwconv2x2_bwd");
}
extern "C" int sv_upsample3to7_add_fwd(const void* small, const void* x, int ldx, void* y, int I, int C, int act_dtype, void* stream) {
  SV_REQUIRE(small && x && y && I > 0 && C > 0 && ldx >= C, "upsample3to7_add_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(upsample_add_fwd_kernel<AT>, dim3(grid_for((long long)I * 49 * C)), dim3(256), 0, STREAM, CA(small), CA(x), ldx, MA(y), I, C););
  return check_launch("sv_upsample3to7_add_fwd");
}
extern "C" int sv_upsample3to7_bwd(const void* dy, void* dsmall, int I, int C, int act_dtype, void* stream) {
  SV_REQUIRE(dy && dsmall && I > 0 && C > 0, "upsample3to7_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(upsample_bwd_kernel<AT>, dim3(grid_for((long long)I * 9 * C)), dim3(256), 0, STREAM, CA(dy), MA(dsmall), I, C););
  return check_launch("sv_upsample3to7_bwd");
}
extern "C" int sv_decoder_head_fwd(const void* x8, const float* w, const float* bias, void* raw12, void* vol, long long M, int act_dtype, void* stream) {
  SV_REQUIRE(x8 && w && raw12 && vol && M > 0, "decoder_head_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(decoder_head_fwd_kernel<AT>, dim3(grid_for(M)), dim3(256), 0, STREAM, CA(x8), w, bias, MA(raw12), MA(vol), M););
  return check_launch("sv_decoder_head_fwd");
}
extern "C" int sv_decoder_head_bwd(const void* draw12, const void* dvol, const void* x8, const float* w, void* dx8, float* dw, float* dbias,
                                   long long M, int act_dtype, void* stream) {
  SV_REQUIRE(draw12 && x8 && w && dx8 && dw && M > 0, "decoder_head_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(decoder_head_bwd_kernel<AT>, dim3(grid_for(M, 256, 1024)), dim3(256), 0, STREAM, CA(draw12), CA(dvol), CA(x8), w, MA(dx8), dw, dbias, M););
  return check_launch("sv_decoder_head_bwd");
}
extern "C" int sv_merge_views_fwd(const void* wlogit, const void* vol, void* out, int B, int V, int S, int act_dtype, void* stream) {
  SV_REQUIRE(wlogit && vol && out && B > 0 && V > 0 && S > 0, "merge_views_fwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(merge_views_fwd_kernel<AT>, dim3(grid_for((long long)B * S)), dim3(256), 0, STREAM, CA(wlogit), CA(vol), MA(out), B, V, S););
  return check_launch("sv_merge_views_fwd");
}
extern "C" int sv_merge_views_bwd(const void* wlogit, const void* vol, const void* out, const void* dout, void* dwlogit, void* dvol,
                                  int B, int V, int S, int act_dtype, void* stream) {
  SV_REQUIRE(wlogit && vol && out && dout && dwlogit && dvol && B > 0 && V > 0 && S > 0, "merge_views_bwd: bad arguments");
  SV_REQUIRE_ACT(act_dtype);
  SV_DISPATCH_ACT(act_dtype, hipLaunchKernelGGL(merge_views_bwd_kernel<AT>, dim3(grid_for((long long)B * S)), dim3(256), 0, STREAM, CA(wlogit), CA(vol), CA(out), CA(dout),
                                                MA(dwlogit), MA(dvol), B, V, S););
  return check_launch("sv_merge_views_bwd");
}
extern "C" int sv_mean_views(const float* vol, float* out, int B, int V, int S, void* stream) {
  SV_REQUIRE(vol && out && B > 0 && V > 0 && S > 0, "mean_views: bad arguments");
  hipLaunchKernelGGL(mean_views_kernel, dim3(grid_for((long long)B * S)), dim3(256), 0, STREAM, vol, out, B, V, S);
  return check_launch("sv_mean_views");
}
extern "C" int sv_bce_logits(const float* x, const float* t, long long n, float* loss_accum, float* dx, const float* gscale_dev, void* stream) {
  SV_REQUIRE(x && t && n > 0 && (loss_accum || dx), "bce_logits: bad arguments");
  hipLaunchKernelGGL(bce_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, STREAM, x, t, n, loss_accum, dx, gscale_dev);
  return check_launch("sv_bce_logits");
}
extern "C" int sv_iou_counts(const float* logits, const float* gt, const float* thresholds_dev, int nth, int B, int S, float* counts, void* stream) {
  SV_REQUIRE(logits && gt && thresholds_dev && counts && nth > 0 && nth <= IOU_MAX_TH && B > 0 && S > 0, "iou_counts: bad arguments (1..8 thresholds)");
  hipLaunchKernelGGL(iou_counts_kernel, dim3(B), dim3(256), 0, STREAM, logits, gt, thresholds_dev, nth, S, counts);
  return check_launch("sv_iou_counts");
}
