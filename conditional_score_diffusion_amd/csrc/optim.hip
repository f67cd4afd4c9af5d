// optim.hip - the parameter update of one training step on flat fp32 buffers:
// gradient clipping by global norm + Adam + exponential moving average in ONE pass over (param, grad, m, v, ema)
// - an HBM-bound stream of 5 reads + 4 writes per element.  Mirrors losses.py:26-53 (torch.optim.Adam with the
// reference's warm-up / clip_grad_norm_) and models/ema.py:61-90.
#include <algorithm>

#include "common.h"

using namespace csd;

namespace {

struct AdamArgs {
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, max_norm, one_minus_decay;
};

__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ ema, const float* __restrict__ grad_norm,
                                                       AdamArgs a, size_t n) {
  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
  float coef = 1.f;
  if (grad_norm && a.max_norm >= 0.f) coef = fminf(a.max_norm / (grad_norm[0] + 1e-6f), 1.f);
  const float step = a.lr / a.bc1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float pv = p[i];
    float gv = g[i] * coef;
    if (a.weight_decay != 0.f) gv = fmaf(a.weight_decay, pv, gv);
    const float mv = a.beta1 * m[i] + (1.f - a.beta1) * gv;              // exp_avg.lerp_(grad, 1 - beta1)
    const float vv = a.beta2 * v[i] + (1.f - a.beta2) * gv * gv;         // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / a.bc2_sqrt + a.eps;
    pv = pv - step * (mv / denom);                                       // param.addcdiv_(exp_avg, denom, value=-step)
    p[i] = pv;
    if (ema) {
      const float s = ema[i];
      ema[i] = s - (s - pv) * a.one_minus_decay;                         // models/ema.py:85-89
    }
  }
}

__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, float one_minus_decay, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float s = ema[i];
    ema[i] = s - (s - p[i]) * one_minus_decay;
  }
}

// ||a||_2 of one long vector: 1024 workgroups leave fp64 partials, the last stage adds them in index order (deterministic)
#define GN_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ a, double* __restrict__ partial, size_t n) {
  __shared__ double red[4];
  double s = 0;
  const size_t n4 = n / 4;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)GN_BLOCKS * 256) {
    const float4 v = a4[i];
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = a[n4 * 4 + threadIdx.x]; s += (double)v * v; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const double* __restrict__ partial, float* __restrict__ out) {
  double s = 0;
  for (int i = threadIdx.x; i < GN_BLOCKS; i += 64) s += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (threadIdx.x == 0) out[0] = (float)sqrt(s);
}

}  // namespace

extern "C" size_t csd_global_norm_scratch_bytes(void) { return GN_BLOCKS * sizeof(double); }

extern "C" int csd_global_norm(const float* a, float* out, int64_t n, void* scratch, void* stream) {
  CSD_REQUIRE(a && out && scratch && n > 0, "global_norm: bad arguments");
  CSD_REQUIRE(((uintptr_t)a & 15) == 0, "global_norm: the vector must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(GN_BLOCKS), dim3(256), 0, (hipStream_t)stream, a, static_cast<double*>(scratch),
                     (size_t)n);
  CSD_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<const double*>(scratch), out);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                             const float* grad_norm, int64_t n, int step, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float max_norm, float ema_decay, void* stream) {
  CSD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.max_norm = max_norm;
  a.one_minus_decay = 1.f - ema_decay;
  const unsigned grid = (unsigned)std::min<size_t>(((size_t)n + 255) / 256, 16384);
  hipLaunchKernelGGL(adam_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, ema,
                     grad_norm, a, (size_t)n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_ema_update(float* ema, const float* param, int64_t n, float decay, void* stream) {
  CSD_REQUIRE(ema && param && n > 0, "ema_update: bad arguments");
  const unsigned grid = (unsigned)std::min<size_t>(((size_t)n + 255) / 256, 16384);
  hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ema, param, 1.f - decay, (size_t)n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}
