// sampler.hip - the predictor / corrector "noise-add" steps of the reverse-SDE sampler and the
// on-device noise source.  HBM-bound elementwise work on the 3-channel state (NCHW fp32).
//
//   langevin   (sampling/correctors.py:58-78, 88-108; VE => alpha = 1):
//       score = net/std; gbar = mean_b ||score_b||_2 ; nbar = mean_b ||z_b||_2
//       step = (snr*nbar/gbar)^2 * 2 ; x_mean = x + step*score ; x = x_mean + sqrt(2*step)*z
//     The batch means couple the samples: a first kernel writes per-(sample,chunk)
//     fp64 partial sums, the update kernel folds them (deterministic order) in its prologue.
//   reverse diffusion (sampling/predictors.py:84-89,97-102 with sde_lib.py:135-140,410-418):
//       x_mean = x + G^2*score ; x = x_mean + G*z
// fp contraction is disabled so that mul/add round exactly like the reference's separate torch ops.
#include <algorithm>

#include "common.h"

#pragma clang fp contract(off)

namespace csd {

#define SQ_THREADS 256

int sumsq_nchunk(int64_t per) {
  int64_t n = per / (SQ_THREADS * 4 * 8);   // >= 8 float4 per thread
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return (int)n;
}

// partial[(b*nchunk + chunk)*2 + {0,1}] = sum a^2 , sum b^2 over the chunk (fp64)
__global__ __launch_bounds__(SQ_THREADS) void sumsq_rows_kernel(const float* __restrict__ a, int64_t a_stride,
                                                                const float* __restrict__ bb,
                                                                double* __restrict__ partial, int64_t per,
                                                                int nchunk) {
  __shared__ double red[2][SQ_THREADS / 64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int64_t len = (per + nchunk - 1) / nchunk;
  const int64_t i0 = chunk * len, i1 = min(per, i0 + len);
  const float* pa = a + (size_t)b * a_stride;
  const float* pb = bb + (size_t)b * per;
  double sa = 0, sb = 0;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += SQ_THREADS) {
    const float va = pa[i], vb = pb[i];
    sa += (double)va * va;
    sb += (double)vb * vb;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sa += __shfl_xor(sa, off);
    sb += __shfl_xor(sb, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = sa;
    red[1][threadIdx.x >> 6] = sb;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0;
    for (int w = 0; w < SQ_THREADS / 64; ++w) { ta += red[0][w]; tb += red[1][w]; }
    partial[((size_t)b * nchunk + chunk) * 2 + 0] = ta;
    partial[((size_t)b * nchunk + chunk) * 2 + 1] = tb;
  }
}

__global__ __launch_bounds__(256) void langevin_update_kernel(float* __restrict__ x, float* __restrict__ x_mean,
                                                              const float* __restrict__ net,
                                                              int64_t net_stride, const float* __restrict__ z,
                                                              const double* __restrict__ partial, int nchunk,
                                                              float std, float snr, float alpha, int B, int64_t per,
                                                              size_t total, int* __restrict__ nonfinite) {
  __shared__ float s_step;
  if (threadIdx.x < 64) {
    // fold partials: lanes stride over samples; each lane folds its samples' chunks in order
    double g = 0, n = 0;
    for (int b = threadIdx.x; b < B; b += 64) {
      double sa = 0, sb = 0;
      for (int c = 0; c < nchunk; ++c) {
        sa += partial[((size_t)b * nchunk + c) * 2 + 0];
        sb += partial[((size_t)b * nchunk + c) * 2 + 1];
      }
      g += sqrt(sa) / (double)std;   // ||net_b/std|| = ||net_b||/std
      n += sqrt(sb);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      g += __shfl_xor(g, off);
      n += __shfl_xor(n, off);
    }
    if (threadIdx.x == 0) {
      const float gbar = (float)(g / B), nbar = (float)(n / B);
      const float r = snr * nbar / gbar;
      s_step = r * r * 2.f * alpha;  // (snr*nbar/gbar)**2 * 2 * alpha; alpha = 1 for the VE SDEs, sde.alphas[timestep] for VP / subVP
      // the finiteness contract of the fused loop: the norms see every element of the score and of the noise, so a NaN / Inf
      // anywhere (an fp16-operand mode past its range) shows here at no cost; csd_pc_sample reports the flag at its final sync
      if (blockIdx.x == 0 && nonfinite && !(fabsf(s_step) <= 3.0e38f)) *nonfinite = 1;
    }
  }
  __syncthreads();
  const float step = s_step;
  const float nz = sqrtf(step * 2.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / (size_t)per;
    const float score = net[b * net_stride + (i - b * per)] / std;
    const float xm = x[i] + step * score;
    x_mean[i] = xm;
    x[i] = xm + nz * z[i];
  }
}

// global-norm exactness mode: sums[0] = sum_b ||net_b||/std, sums[1] = sum_b ||z_b|| over THIS rank's samples,
// fp32 like the reference's torch.norm(...).mean() operands; the caller all-reduces the two floats over the ranks
__global__ __launch_bounds__(64) void norm_sums_kernel(const double* __restrict__ partial, int nchunk, float std, int B,
                                                       float* __restrict__ sums) {
  double g = 0, n = 0;
  for (int b = threadIdx.x; b < B; b += 64) {
    double sa = 0, sb = 0;
    for (int c = 0; c < nchunk; ++c) {
      sa += partial[((size_t)b * nchunk + c) * 2 + 0];
      sb += partial[((size_t)b * nchunk + c) * 2 + 1];
    }
    g += sqrt(sa) / (double)std;
    n += sqrt(sb);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    g += __shfl_xor(g, off);
    n += __shfl_xor(n, off);
  }
  if (threadIdx.x == 0) { sums[0] = (float)g; sums[1] = (float)n; }
}

// the Langevin update with the step size of the GLOBAL batch: gbar = sums[0] / Bg, nbar = sums[1] / Bg
__global__ __launch_bounds__(256) void langevin_update_global_kernel(float* __restrict__ x, float* __restrict__ x_mean,
                                                                     const float* __restrict__ net, int64_t net_stride,
                                                                     const float* __restrict__ z,
                                                                     const float* __restrict__ sums, int Bg, float std,
                                                                     float snr, float alpha, int64_t per, size_t total,
                                                                     int* __restrict__ nonfinite) {
  const float gbar = sums[0] / (float)Bg, nbar = sums[1] / (float)Bg;
  const float r = snr * nbar / gbar;
  const float step = r * r * 2.f * alpha;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nonfinite && !(fabsf(step) <= 3.0e38f)) *nonfinite = 1;
  const float nz = sqrtf(step * 2.f);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / (size_t)per;
    const float score = net[b * net_stride + (i - b * per)] / std;
    const float xm = x[i] + step * score;
    x_mean[i] = xm;
    x[i] = xm + nz * z[i];
  }
}

__global__ __launch_bounds__(256) void reverse_diffusion_update_kernel(float* __restrict__ x,
                                                                       float* __restrict__ x_mean,
                                                                       const float* __restrict__ net,
                                                                       int64_t net_stride,
                                                                       const float* __restrict__ z, float std,
                                                                       float G, int64_t per, size_t total) {
  const float G2 = G * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / (size_t)per;
    const float score = net[b * net_stride + (i - b * per)] / std;
    const float rev_f = 0.f - G2 * score;   // f = 0 for VE
    const float xm = x[i] - rev_f;
    x_mean[i] = xm;
    x[i] = xm + G * z[i];
  }
}

// out[b] = ||a_b||_2 (fp64 sum of squares, one workgroup per sample): the per-sample norms of the Langevin step size
// (sampling/correctors.py:102-103) for callers that combine them across ranks before the update
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ a, float* __restrict__ out, int64_t per) {
  __shared__ double red[4];
  const float* p = a + (size_t)blockIdx.x * per;
  double s = 0;
  for (int64_t i = threadIdx.x; i < per; i += 256) { const float v = p[i]; s += (double)v * v; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
}

// general one-step update shared by the Euler-Maruyama / ancestral-sampling predictors and the annealed
// Langevin corrector (sampling/predictors.py:52-76,105-179, correctors.py:111-142 of the reference): with
// per-call scalars p, a, c     x_mean = p*x + a*score,   x = x_mean + c*z
__global__ __launch_bounds__(256) void affine_noise_update_kernel(float* __restrict__ x, float* __restrict__ x_mean,
                                                                  const float* __restrict__ score,
                                                                  const float* __restrict__ z, float p, float a,
                                                                  float c, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const float xm = p * x[i] + a * score[i];
    x_mean[i] = xm;
    x[i] = xm + c * z[i];
  }
}

// ---- Philox4x32-10 + Box-Muller ------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__global__ void randn_kernel(float* __restrict__ out, size_t n, uint64_t seed, uint64_t stream_id) {
  const size_t n4 = (n + 3) / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    uint32_t c0 = (uint32_t)i, c1 = (uint32_t)(i >> 32), c2 = (uint32_t)stream_id, c3 = (uint32_t)(stream_id >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    // uniforms in (0,1]: (u32 + 1) * 2^-32 computed via the top 24 bits to stay exact in fp32
    const float u0 = ((float)(c0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c2 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, cs0, s1, cs1;
    sincosf(6.283185307179586f * u1, &s0, &cs0);
    sincosf(6.283185307179586f * u3, &s1, &cs1);
    const float v[4] = {r0 * cs0, r0 * s0, r1 * cs1, r1 * s1};
    const size_t base = i * 4;
    if (base + 3 < n) {
      *reinterpret_cast<float4*>(out + base) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int j = 0; j < 4 && base + j < n; ++j) out[base + j] = v[j];
    }
  }
}

__global__ void scale_rows_kernel(float* __restrict__ out, const float* __restrict__ in,
                                  const float* __restrict__ scale, int divide, size_t per, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const float sc = scale[i / per];
    out[i] = divide ? in[i] / sc : in[i] * sc;
  }
}

// the finiteness contract for loops without a Langevin corrector, and for the state the loop returns: one pass over x at the END
__global__ __launch_bounds__(256) void finite_check_kernel(const float* __restrict__ x, size_t total, int* __restrict__ nonfinite) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) bad |= !(fabsf(x[i]) <= 3.0e38f);
  if (bad) *nonfinite = 1;
}

static int ew_grid(size_t total) { return (int)std::min<size_t>(cdiv64(total, 256), 4096); }

int finite_check_launch(const float* x, size_t total, int* nonfinite, hipStream_t s) {
  hipLaunchKernelGGL(finite_check_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, total, nonfinite);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int sumsq_rows_launch(const float* net, int64_t net_stride, const float* z, double* partial, int B, int64_t per,
                      int nchunk, hipStream_t s) {
  hipLaunchKernelGGL(sumsq_rows_kernel, dim3(nchunk, B), dim3(SQ_THREADS), 0, s, net, net_stride, z, partial, per,
                     nchunk);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int langevin_update_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z,
                           const double* partial, int nchunk, float std, float snr, float alpha, int B, int64_t per,
                           hipStream_t s, int* nonfinite) {
  const size_t total = (size_t)B * per;
  hipLaunchKernelGGL(langevin_update_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, x_mean, net, net_stride, z,
                     partial, nchunk, std, snr, alpha, B, per, total, nonfinite);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int norm_sums_launch(const double* partial, int nchunk, float std, int B, float* sums, hipStream_t s) {
  hipLaunchKernelGGL(norm_sums_kernel, dim3(1), dim3(64), 0, s, partial, nchunk, std, B, sums);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int langevin_update_global_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z,
                                  const float* sums, int Bg, float std, float snr, float alpha, int B, int64_t per,
                                  hipStream_t s, int* nonfinite) {
  const size_t total = (size_t)B * per;
  hipLaunchKernelGGL(langevin_update_global_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, x_mean, net, net_stride, z,
                     sums, Bg, std, snr, alpha, per, total, nonfinite);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int reverse_diffusion_update_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z,
                                    float std, float G, int B, int64_t per, hipStream_t s) {
  const size_t total = (size_t)B * per;
  hipLaunchKernelGGL(reverse_diffusion_update_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, x_mean, net,
                     net_stride, z, std, G, per, total);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// x_mean = p*x + (a/std)*net, x = x_mean + c*z on the network's (row-strided) output: the affine update rules inside the fused loop
__global__ __launch_bounds__(256) void affine_net_update_kernel(float* __restrict__ x, float* __restrict__ x_mean,
                                                                const float* __restrict__ net, int64_t net_stride,
                                                                const float* __restrict__ z, float std, float p, float a, float c,
                                                                int64_t per, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / (size_t)per;
    const float score = net[b * net_stride + (i - b * per)] / std;
    const float xm = p * x[i] + a * score;
    x_mean[i] = xm;
    x[i] = xm + c * z[i];
  }
}

int affine_net_update_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z, float std, float p,
                             float a, float c, int B, int64_t per, hipStream_t s) {
  const size_t total = (size_t)B * per;
  hipLaunchKernelGGL(affine_net_update_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, x_mean, net, net_stride, z, std, p, a, c,
                     per, total);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int affine_noise_update_launch(float* x, float* x_mean, const float* score, const float* z, float p, float a, float c,
                               size_t total, hipStream_t s) {
  hipLaunchKernelGGL(affine_noise_update_kernel, dim3(ew_grid(total)), dim3(256), 0, s, x, x_mean, score, z, p, a, c,
                     total);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int randn_launch(float* out, int64_t n, uint64_t seed, uint64_t stream_id, hipStream_t s) {
  if (n <= 0) return CSD_OK;
  CSD_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "randn: output must be 16-byte aligned");
  hipLaunchKernelGGL(randn_kernel, dim3(ew_grid((size_t)(n + 3) / 4)), dim3(256), 0, s, out, (size_t)n, seed,
                     stream_id);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int scale_rows_launch(float* out, const float* in, const float* scale, int divide, int B, int64_t per,
                      hipStream_t s) {
  const size_t total = (size_t)B * per;
  if (total == 0) return CSD_OK;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(ew_grid(total)), dim3(256), 0, s, out, in, scale, divide,
                     (size_t)per, total);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// use_path bridge (sde_lib.py:323-339): y_t = w0 * y0 + w1 * y_{t+tau} + std * z, in place on the state; prev == nullptr: w1 term absent
__global__ void bridge_update_kernel(const float* __restrict__ y0, float* __restrict__ state, const float* __restrict__ z, float w0,
                                     float w1, float sd, int has_prev, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float m = has_prev ? y0[i] * w0 + state[i] * w1 : y0[i] * w0;      // (the host path: axpby(axpby(y, y_prev, w0, w1), z, 1, std))
    state[i] = m * 1.0f + z[i] * sd;
  }
}

int bridge_update_launch(const float* y0, float* state, const float* z, float w0, float w1, float sd, int has_prev, size_t n,
                         hipStream_t s) {
  const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(bridge_update_kernel, dim3(grid), dim3(256), 0, s, y0, state, z, w0, w1, sd, has_prev, n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

}  // namespace csd

using namespace csd;

extern "C" size_t csd_update_scratch_bytes(int B) { return (size_t)B * 64 * 2 * sizeof(double); }

extern "C" int csd_langevin_step(float* x, float* x_mean, const float* net, const float* z, float std, float snr, float alpha,
                                 int B, int64_t per_sample, void* scratch, void* stream) {
  CSD_REQUIRE(B > 0 && per_sample > 0 && scratch && alpha > 0.f, "langevin_step: bad arguments");
  const int nchunk = sumsq_nchunk(per_sample);
  int rc = sumsq_rows_launch(net, per_sample, z, (double*)scratch, B, per_sample, nchunk, (hipStream_t)stream);
  if (rc) return rc;
  return langevin_update_launch(x, x_mean, net, per_sample, z, (const double*)scratch, nchunk, std, snr, alpha, B, per_sample,
                                (hipStream_t)stream);
}

extern "C" int csd_reverse_diffusion_step(float* x, float* x_mean, const float* net, const float* z, float std,
                                          float G, int B, int64_t per_sample, void* stream) {
  CSD_REQUIRE(B > 0 && per_sample > 0, "reverse_diffusion_step: bad arguments");
  return reverse_diffusion_update_launch(x, x_mean, net, per_sample, z, std, G, B, per_sample, (hipStream_t)stream);
}

extern "C" int csd_row_norms(const float* a, float* out, int B, int64_t per_sample, void* stream) {
  CSD_REQUIRE(a && out && B > 0 && per_sample > 0, "row_norms: bad arguments");
  hipLaunchKernelGGL(row_norms_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, out, per_sample);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_affine_noise_step(float* x, float* x_mean, const float* score, const float* z, float p, float a,
                                     float c, int64_t n, void* stream) {
  CSD_REQUIRE(x && x_mean && score && z && n > 0, "affine_noise_step: bad arguments");
  return affine_noise_update_launch(x, x_mean, score, z, p, a, c, (size_t)n, (hipStream_t)stream);
}

extern "C" int csd_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  return randn_launch(out, n, seed, stream_id, (hipStream_t)stream);
}

extern "C" int csd_scale_rows(float* out, const float* in, const float* scale, int divide, int B,
                              int64_t per_sample, void* stream) {
  return scale_rows_launch(out, in, scale, divide, B, per_sample, (hipStream_t)stream);
}
