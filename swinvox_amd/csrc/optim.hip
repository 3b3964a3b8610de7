// Optimiser step on ONE flat fp32 buffer per module (reference core/train.py:98-131 solvers, :279-292 clip + step).
//
// The reference clips each module's gradient to an L2 norm of 1.0 and then runs torch.optim.Adam / SGD on ~340 separate
// parameter tensors.  Here a module's parameters, gradients and optimiser moments are contiguous buffers with one shared
// layout, so the whole update is two streaming launches: a sum of squares into 16 double slots and one fused
// clip + weight-decay + moment + parameter update that reads the slots on the device (no host synchronisation).
// HBM traffic per step: 4 B (norm) + 28 B (Adam: p, g, m, v in; p, m, v out) per parameter.
#include "common.h"

namespace sv {

constexpr int OPT_SLOTS = 16;

__global__ void __launch_bounds__(256) grad_sumsq_kernel(const float* __restrict__ g, long long n, float gscale, double* __restrict__ slots) {
  __shared__ float red[4];
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc = fmaf(v.x, v.x, acc);
    acc = fmaf(v.y, v.y, acc);
    acc = fmaf(v.z, v.z, acc);
    acc = fmaf(v.w, v.w, acc);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    acc = fmaf(v, v, acc);
  }
  const float s = block_sum<4>(acc, red);
  if (threadIdx.x == 0) atomicAdd(&slots[blockIdx.x & (OPT_SLOTS - 1)], (double)s * (double)gscale * (double)gscale);
}

// gradient multiplier of the step: gscale (1/world of the data-parallel mean) x the clip_grad_norm_ coefficient
// min(1, max_norm / (||g|| + 1e-6)) (torch.nn.utils.clip_grad_norm_, reference core/train.py:279-282)
// `finite` = the squared norm is finite, i.e. no gradient element is inf / NaN.  A non-finite gradient SKIPS the step
// (parameters, moments and the effective step count stay untouched) - what torch.amp.GradScaler.step() does in the
// reference after unscale_ found an inf (core/train.py:276-293) - instead of poisoning p / m / v with NaN for good.
__device__ __forceinline__ float grad_multiplier(const double* slots, float gscale, float max_norm, bool& finite) {
  finite = true;
  if (slots == nullptr) return gscale;
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < OPT_SLOTS; ++i) s += slots[i];
  finite = isfinite(s);
  if (max_norm <= 0.f) return gscale;
  const float coef = max_norm / ((float)sqrt(s) + 1e-6f);
  return gscale * fminf(coef, 1.f);
}

// steps that really happened = host step count - device count of skipped steps (read before block 0 may bump it: every
// block of a skipped step returns without reading it, every block of a taken step only reads it)
__device__ __forceinline__ long long effective_step(long long step, long long* skipped, bool finite) {
  if (!finite) {
    if (skipped != nullptr && blockIdx.x == 0 && threadIdx.x == 0) skipped[0] += 1;
    return 0;
  }
  return step - (skipped != nullptr ? skipped[0] : 0);
}

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, gscale, max_norm;
  double dbeta1, dbeta2, dlr;
  long long step;
  float step_size, bc2_sqrt;   // filled on the device from the effective step
};

// torch.optim.Adam (coupled L2 weight decay, no amsgrad), the operation order of torch/optim/adam.py::_single_tensor_adam
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float gm) {
  g = g * gm;
  g = fmaf(a.weight_decay, p, g);
  const float w = 1.f - a.beta1;                      // exp_avg.lerp_(grad, 1 - beta1), both branches of ATen's lerp
  m = w < 0.5f ? m + w * (g - m) : g - (g - m) * (1.f - w);
  v = fmaf(g * g, 1.f - a.beta2, v * a.beta2);        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, AdamArgs a, const double* __restrict__ slots,
                                                        long long* __restrict__ skipped) {
  bool finite;
  const float gm = grad_multiplier(slots, a.gscale, a.max_norm, finite);
  const long long t = effective_step(a.step, skipped, finite);
  if (!finite) return;
  a.step_size = (float)(a.dlr / (1.0 - pow(a.dbeta1, (double)t)));        // lr / bias_correction1
  a.bc2_sqrt = (float)sqrt(1.0 - pow(a.dbeta2, (double)t));
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, a, gm);
    adam_one(pp.y, gg.y, mm.y, vv.y, a, gm);
    adam_one(pp.z, gg.z, mm.z, vv.z, a, gm);
    adam_one(pp.w, gg.w, mm.w, vv.w, a, gm);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    adam_one(p[i], g[i], m[i], v[i], a, gm);
  }
}

struct SgdArgs {
  float lr, momentum, weight_decay, gscale, max_norm;
  long long step;
  int first;                   // filled on the device: the effective step is the first one
};

// torch.optim.SGD (momentum, dampening 0, no nesterov): buf = g (first step) or momentum * buf + g; p -= lr * buf
__device__ __forceinline__ void sgd_one(float& p, float g, float& b, const SgdArgs& a, float gm) {
  g = g * gm;
  g = fmaf(a.weight_decay, p, g);
  b = a.first ? g : fmaf(a.momentum, b, g);
  p = p - a.lr * b;
}

__global__ void __launch_bounds__(256) sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                                                       SgdArgs a, const double* __restrict__ slots, long long* __restrict__ skipped) {
  bool finite;
  const float gm = grad_multiplier(slots, a.gscale, a.max_norm, finite);
  const long long t = effective_step(a.step, skipped, finite);
  if (!finite) return;
  a.first = t == 1 ? 1 : 0;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 bb = reinterpret_cast<float4*>(buf)[i];
    sgd_one(pp.x, gg.x, bb.x, a, gm);
    sgd_one(pp.y, gg.y, bb.y, a, gm);
    sgd_one(pp.z, gg.z, bb.z, a, gm);
    sgd_one(pp.w, gg.w, bb.w, a, gm);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(buf)[i] = bb;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    sgd_one(p[i], g[i], buf[i], a, gm);
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int stream_grid(long long n) {
  const long long b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace sv

using namespace sv;
#define STREAM static_cast<hipStream_t>(stream)

extern "C" int sv_grad_sumsq(const float* g, long long n, float gscale, double* slots16, void* stream) {
  SV_REQUIRE(g && slots16 && n > 0, "grad_sumsq: bad arguments");
  SV_REQUIRE(aligned16(g), "grad_sumsq: the gradient buffer must be 16-byte aligned");
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(stream_grid(n)), dim3(256), 0, STREAM, g, n, gscale, slots16);
  return check_launch("sv_grad_sumsq");
}

extern "C" int sv_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                            double weight_decay, long long step, float gscale, const double* slots16, float max_norm,
                            long long* skipped_steps, void* stream) {
  SV_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  SV_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "adam_step: buffers must be 16-byte aligned");
  SV_REQUIRE(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "adam_step: bad hyper-parameters");
  AdamArgs a;
  a.lr = (float)lr;
  a.dlr = lr;
  a.dbeta1 = beta1;
  a.dbeta2 = beta2;
  a.step = step;
  a.beta1 = (float)beta1;
  a.beta2 = (float)beta2;
  a.eps = (float)eps;
  a.weight_decay = (float)weight_decay;
  a.gscale = gscale;
  a.max_norm = max_norm;
  a.step_size = 0.f;
  a.bc2_sqrt = 1.f;
  hipLaunchKernelGGL(adam_step_kernel, dim3(stream_grid(n)), dim3(256), 0, STREAM, p, g, m, v, n, a, slots16, skipped_steps);
  return check_launch("sv_adam_step");
}

extern "C" int sv_sgd_step(float* p, const float* g, float* buf, long long n, double lr, double momentum, double weight_decay, long long step,
                           float gscale, const double* slots16, float max_norm, long long* skipped_steps, void* stream) {
  SV_REQUIRE(p && g && buf && n > 0 && step >= 1, "sgd_step: bad arguments");
  SV_REQUIRE(aligned16(p) && aligned16(g) && aligned16(buf), "sgd_step: buffers must be 16-byte aligned");
  SgdArgs a;
  a.lr = (float)lr;
  a.momentum = (float)momentum;
  a.weight_decay = (float)weight_decay;
  a.gscale = gscale;
  a.max_norm = max_norm;
  a.step = step;
  a.first = 0;
  hipLaunchKernelGGL(sgd_step_kernel, dim3(stream_grid(n)), dim3(256), 0, STREAM, p, g, buf, n, a, slots16, skipped_steps);
  return check_launch("sv_sgd_step");
}
