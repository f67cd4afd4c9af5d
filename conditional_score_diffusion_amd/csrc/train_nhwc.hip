// train_nhwc.hip - NHWC entry points of the training operators: the differentiable DDPM-family graph keeps its activations
// in the library's own layout ([B, H, W, C] fp32), so no layer pays a layout change (csd_conv2d_ex / csd_conv2d_wgrad_ex take
// layout flags; the kernels are the same).  GroupNorm(+act) forward reuses the streaming statistics kernels of norm.hip and keeps
// the per-(sample, channel) statistics for a three-kernel backward; attention works on a packed qkv tensor.
#include <algorithm>

#include "common.h"

using namespace csd;

namespace {

inline size_t al64(size_t v) { return (v + 63) / 64 * 64; }

__device__ __forceinline__ float t_act(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: return v / (1.0f + expf(-v));
    case CSD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case CSD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
__device__ __forceinline__ float t_dact(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: { const float s = 1.0f / (1.0f + expf(-v)); return s * (1.0f + v * (1.0f - s)); }
    case CSD_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? 1.f : 0.2f;
    case CSD_ACT_ELU: return v > 0.f ? 1.f : expf(v);
    default: return 1.f;
  }
}

#define TN_THREADS 256
// pass 1 of the GroupNorm backward: per (sample, pixel chunk) and channel, sum du and sum du*xhat over the chunk's pixels.
// thread = (pixel row, 4 channels); rows are folded through LDS in row order (deterministic).  partial [B][nchunk][C][2]
__global__ __launch_bounds__(TN_THREADS) void gn_bwd_stats_nhwc_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ rs, const float* __restrict__ ms, double* __restrict__ partial, int HW, int C, int act, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) double sred[];   // [rows][C][2] folded in place
  const int C4 = C >> 2;
  const int rows = TN_THREADS / C4;
  const int tid = threadIdx.x;
  const int row = tid / C4, c = (tid - row * C4) * 4;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const bool active = row < rows;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (active) {
    const float4 r4 = *reinterpret_cast<const float4*>(rs + (size_t)b * C + c);
    const float4 m4 = *reinterpret_cast<const float4*>(ms + (size_t)b * C + c);
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b4 = *reinterpret_cast<const float4*>(beta + c);
    const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const size_t base = (size_t)b * HW * C + c;
    for (int p = p0 + row; p < p1; p += rows) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
      const float4 dv = *reinterpret_cast<const float4*>(dy + base + (size_t)p * C);
      const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = xa[j] * rr[j] + mm[j];
        const float du = da[j] * t_dact(xh * gg[j] + bb[j], act);
        s1[j] += du;
        s2[j] += (double)du * xh;
      }
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sred[((size_t)row * C + c + j) * 2] = s1[j];
      sred[((size_t)row * C + c + j) * 2 + 1] = s2[j];
    }
  }
  __syncthreads();
  for (int i = tid; i < C * 2; i += TN_THREADS) {
    double a = 0;
    for (int r = 0; r < rows; ++r) a += sred[(size_t)r * C * 2 + i];
    partial[((size_t)b * nchunk + chunk) * C * 2 + i] = a;
  }
}

// pass 2: fold the chunks; dgamma / dbeta rows; per-(sample, channel) coefficients of pass 3:
//   dx = du*ca - cb - xhat*cc,  ca = rstd*gamma, cb = rstd*mean_g(du*gamma), cc = rstd*mean_g(du*gamma*xhat)
// (the chunk fold runs on L lanes per channel, lane l taking chunks l, l + L, ... and the lanes folded in lane order: one thread per
// channel walking 128 chunks of a small-batch step was 17 us of pure latency per GroupNorm)
__global__ void gn_bwd_final_nhwc_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                         const float* __restrict__ rs, float* __restrict__ dgamma_rows,
                                         float* __restrict__ dbeta_rows, float* __restrict__ coef, int HW, int C, int G,
                                         int nchunk, int row_stride, int L) {
  extern __shared__ __attribute__((aligned(16))) double sh[];   // [C][2], then [L][C][2]
  double* const lane_sums = sh + (size_t)C * 2;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < L * C; i += blockDim.x) {
    const int l = i / C, c = i - l * C;
    double a1 = 0, a2 = 0;
    for (int k = l; k < nchunk; k += L) {
      const double* p = partial + (((size_t)b * nchunk + k) * C + c) * 2;
      a1 += p[0];
      a2 += p[1];
    }
    lane_sums[(size_t)i * 2] = a1;
    lane_sums[(size_t)i * 2 + 1] = a2;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double a1 = 0, a2 = 0;
    for (int l = 0; l < L; ++l) {
      a1 += lane_sums[((size_t)l * C + c) * 2];
      a2 += lane_sums[((size_t)l * C + c) * 2 + 1];
    }
    sh[c * 2] = a1;
    sh[c * 2 + 1] = a2;
    dbeta_rows[(size_t)b * row_stride + c] = (float)a1;
    dgamma_rows[(size_t)b * row_stride + c] = (float)a2;
  }
  __syncthreads();
  const int cpg = C / G;
  const double n = (double)HW * cpg;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cpg;
    double A = 0, Bq = 0;
    for (int j = 0; j < cpg; ++j) {
      const int cc = g * cpg + j;
      A += (double)gamma[cc] * sh[cc * 2];
      Bq += (double)gamma[cc] * sh[cc * 2 + 1];
    }
    const float rstd = rs[(size_t)b * C + c];
    coef[((size_t)b * 3 + 0) * C + c] = rstd * gamma[c];
    coef[((size_t)b * 3 + 1) * C + c] = rstd * (float)(A / n);
    coef[((size_t)b * 3 + 2) * C + c] = rstd * (float)(Bq / n);
  }
}

// pass 3: dx, elementwise; thread = (pixel row, 4 channels) with its coefficients in registers
__global__ __launch_bounds__(TN_THREADS) void gn_bwd_apply_nhwc_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ rs, const float* __restrict__ ms, const float* __restrict__ coef, const float* __restrict__ add,
    float* __restrict__ dx, int HW, int C, int act, int nchunk) {
  const int C4 = C >> 2;
  const int rows = TN_THREADS / C4;
  const int tid = threadIdx.x;
  const int row = tid / C4, c = (tid - row * C4) * 4;
  if (row >= rows) return;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  float rr[4], mm[4], gg[4], bb[4], ca[4], cb[4], cc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rr[j] = rs[(size_t)b * C + c + j]; mm[j] = ms[(size_t)b * C + c + j];
    gg[j] = gamma[c + j]; bb[j] = beta[c + j];
    ca[j] = coef[((size_t)b * 3 + 0) * C + c + j];
    cb[j] = coef[((size_t)b * 3 + 1) * C + c + j];
    cc[j] = coef[((size_t)b * 3 + 2) * C + c + j];
  }
  const size_t base = (size_t)b * HW * C + c;
  for (int p = p0 + row; p < p1; p += rows) {
    const float4 xv = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
    const float4 dv = *reinterpret_cast<const float4*>(dy + base + (size_t)p * C);
    const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = xa[j] * rr[j] + mm[j];
      const float du = da[j] * t_dact(xh * gg[j] + bb[j], act);
      o[j] = du * ca[j] - cb[j] - xh * cc[j];
    }
    if (add) {                                     // a second gradient path into the same tensor (residual shortcut), fused
      const float4 av = *reinterpret_cast<const float4*>(add + base + (size_t)p * C);
      o[0] += av.x; o[1] += av.y; o[2] += av.z; o[3] += av.w;
    }
    *reinterpret_cast<float4*>(dx + base + (size_t)p * C) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// out[b][p][c] = x[b][p][c] + bias[b][c]
__global__ void bias_add_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out,
                                     int HW, int C4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / C4;
    const int c4 = (int)(i - pix * C4);
    const size_t b = pix / HW;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 bv = reinterpret_cast<const float4*>(bias)[b * C4 + c4];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
  }
}

// per-(sample, chunk) column sums of [HW, C] -> partial [B][nchunk][C] (fp64), then folded by sum_pixels_final
__global__ __launch_bounds__(TN_THREADS) void sum_pixels_nhwc_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                                     int HW, int C, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) double sred[];   // [rows][C]
  const int C4 = C >> 2;
  const int rows = TN_THREADS / C4;
  const int tid = threadIdx.x;
  const int row = tid / C4, c = (tid - row * C4) * 4;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  if (row < rows) {
    double s[4] = {0, 0, 0, 0};
    const size_t base = (size_t)b * HW * C + c;
    for (int p = p0 + row; p < p1; p += rows) {
      const float4 v = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sred[(size_t)row * C + c + j] = s[j];
  }
  __syncthreads();
  for (int i = tid; i < C; i += TN_THREADS) {
    double a = 0;
    for (int r = 0; r < rows; ++r) a += sred[(size_t)r * C + i];
    partial[((size_t)b * nchunk + chunk) * C + i] = a;
  }
}
// 32 channels x 8 chunk lanes per workgroup (lane l: chunks l, l + 8, ...; lanes folded in lane order)
__global__ __launch_bounds__(256) void sum_pixels_final_kernel(const double* __restrict__ partial, float* __restrict__ out, int C, int nchunk) {
  __shared__ double sh[8][32];
  const int b = blockIdx.y;
  const int tx = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  double a = 0;
  if (c < C)
    for (int k = l; k < nchunk; k += 8) a += partial[((size_t)b * nchunk + k) * C + c];
  sh[l][tx] = a;
  __syncthreads();
  if (l == 0 && c < C) {
    double t = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += sh[j][tx];
    out[(size_t)b * C + c] = (float)t;
  }
}

// z[b][2y+1][2x+1][c] = dy[b][y][x][c], zeros elsewhere (data gradient of the pad-(0,1,0,1) stride-2 Downsample conv)
__global__ void zero_insert_odd_nhwc_kernel(const float* __restrict__ dy, float* __restrict__ z, int h, int w, int C4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int X = (int)(r % (2 * w)); r /= (2 * w);
    const int Y = (int)(r % (2 * h));
    const size_t b = r / (2 * h);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((X & 1) && (Y & 1)) v = reinterpret_cast<const float4*>(dy)[((b * h + (Y >> 1)) * w + (X >> 1)) * C4 + c4];
    reinterpret_cast<float4*>(z)[i] = v;
  }
}
// out[b][y][x][c] = sum of the 2x2 block of in (data gradient of the nearest x2 upsample in front of a conv)
__global__ void sumpool2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int h, int w, int C4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int X = (int)(r % w); r /= w;
    const int Y = (int)(r % h);
    const size_t b = r / h;
    const float4* src = reinterpret_cast<const float4*>(in);
    const size_t row0 = ((b * 2 * h + 2 * Y) * 2 * w + 2 * X) * C4 + c4, row1 = row0 + (size_t)2 * w * C4;
    const float4 a = src[row0], bq = src[row0 + C4], cq = src[row1], d = src[row1 + C4];
    reinterpret_cast<float4*>(out)[i] = make_float4(a.x + bq.x + cq.x + d.x, a.y + bq.y + cq.y + d.y, a.z + bq.z + cq.z + d.z,
                                                    a.w + bq.w + cq.w + d.w);
  }
}

inline unsigned ew_grid(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 65536); }

int tn_chunks(int B, int HW, int C) {
  int nchunk = cdiv(2048, B);
  const int rows = TN_THREADS / (C / 4);
  const int maxc = std::max(1, HW / std::max(1, 4 * rows));
  return std::max(1, std::min(nchunk, maxc));
}

}  // namespace

extern "C" size_t csd_groupnorm_nhwc_scratch_bytes(int B, int C, int HW) {
  GNPlan g;
  if (gn_plan(&g, B, HW, C, 0, 1)) return 0;
  const size_t gn = gn_partial_bytes(g) * 64;
  const size_t bw = (size_t)B * tn_chunks(B, HW, C) * C * 2 * sizeof(double) + (size_t)B * 3 * C * sizeof(float);
  return std::max(gn, bw) + 2 * al64((size_t)B * C) * sizeof(float) + 1024;
}

// y = act(GroupNorm(x)); rs / ms [B, C] receive the statistics (rstd, -mean*rstd per channel) the backward needs
extern "C" int csd_groupnorm_act_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* rs, float* ms,
                                      int B, int C, int HW, int groups, float eps, int act, void* scratch, void* stream) {
  return csd::groupnorm_act_dropout_nhwc(x, gamma, beta, y, rs, ms, nullptr, 0.f, 0, 0, B, C, HW, groups, eps, act, scratch, stream);
}

int csd::groupnorm_act_dropout_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* rs, float* ms, float* mask,
                                    float p_drop, uint64_t seed, uint64_t stream_id, int B, int C, int HW, int groups, float eps,
                                    int act, void* scratch, void* stream, void* planes, int plane_count) {
  CSD_REQUIRE(x && gamma && beta && y && rs && ms && scratch, "groupnorm_act_nhwc: null argument");
  CSD_REQUIRE(mask == nullptr || (p_drop > 0.f && p_drop < 1.f && ((size_t)B * HW * C) % 4 == 0), "groupnorm_act_nhwc: dropout p = %f", (double)p_drop);
  hipStream_t s = (hipStream_t)stream;
  GNPlan g;
  int rc = gn_plan(&g, B, HW, C, 0, groups);
  if (rc) return rc;
  float* f = static_cast<float*>(scratch);
  float* sc = f; f += al64((size_t)B * C);
  float* sh = f; f += al64((size_t)B * C);
  double* partial = reinterpret_cast<double*>(f);
  if ((rc = gn_stats_launch(g, x, nullptr, partial, s))) return rc;
  if ((rc = gn_finalize_launch(g, partial, gamma, beta, eps, sc, sh, s, rs, ms))) return rc;
  void* hi = plane_count > 0 ? planes : nullptr;
  void* lo = plane_count > 1 ? static_cast<void*>(static_cast<char*>(planes) + (size_t)B * HW * C * 2) : nullptr;
  return gn_apply_launch(x, sc, sh, y, B, HW, C, act, s, mask, p_drop, seed, stream_id, hi, lo);
}

// dx = GroupNorm(+act) backward of dy (+ add, when given: the gradient arriving over the residual shortcut)
int csd::groupnorm_act_backward_nhwc_add(const float* x, const float* gamma, const float* beta, const float* rs, const float* ms,
                                         const float* dy, const float* add, float* dx, float* dgamma_rows, float* dbeta_rows,
                                         int row_stride, int B, int C, int HW, int groups, int act, void* scratch, void* stream) {
  CSD_REQUIRE(x && gamma && beta && rs && ms && dy && dx && dgamma_rows && dbeta_rows && scratch, "groupnorm_act_backward_nhwc: null argument");
  CSD_REQUIRE(C % 4 == 0 && C <= 1024 && groups > 0 && C % groups == 0, "groupnorm_act_backward_nhwc: C=%d groups=%d unsupported", C, groups);
  CSD_REQUIRE(row_stride >= C, "groupnorm_act_backward_nhwc: row_stride %d < C", row_stride);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = tn_chunks(B, HW, C);
  const int rows = TN_THREADS / (C / 4);
  double* partial = static_cast<double*>(scratch);
  float* coef = reinterpret_cast<float*>(partial + (size_t)B * nchunk * C * 2);
  hipLaunchKernelGGL(gn_bwd_stats_nhwc_kernel, dim3(nchunk, B), dim3(TN_THREADS), (size_t)rows * C * 2 * sizeof(double), s, x, dy,
                     gamma, beta, rs, ms, partial, HW, C, act, nchunk);
  CSD_LAUNCH_CHECK();
  const int fl = std::max(1, std::min(std::min(16, nchunk), 1024 / C));      // chunk lanes per channel of the fold
  hipLaunchKernelGGL(gn_bwd_final_nhwc_kernel, dim3(B), dim3(std::min(1024, std::max(256, fl * C))), (size_t)C * 2 * (1 + fl) * sizeof(double), s,
                     partial, gamma, rs, dgamma_rows, dbeta_rows, coef, HW, C, groups, nchunk, row_stride, fl);
  CSD_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_apply_nhwc_kernel, dim3(nchunk, B), dim3(TN_THREADS), 0, s, x, dy, gamma, beta, rs, ms, coef, add, dx,
                     HW, C, act, nchunk);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_groupnorm_act_backward_nhwc(const float* x, const float* gamma, const float* beta, const float* rs,
                                               const float* ms, const float* dy, float* dx, float* dgamma_rows,
                                               float* dbeta_rows, int row_stride, int B, int C, int HW, int groups, int act,
                                               void* scratch, void* stream) {
  return groupnorm_act_backward_nhwc_add(x, gamma, beta, rs, ms, dy, nullptr, dx, dgamma_rows, dbeta_rows, row_stride, B, C, HW, groups,
                                         act, scratch, stream);
}

extern "C" int csd_bias_add_nhwc(const float* x, const float* bias, float* out, int B, int HW, int C, void* stream) {
  CSD_REQUIRE(x && bias && out && C % 4 == 0, "bias_add_nhwc: bad arguments");
  const size_t total4 = (size_t)B * HW * C / 4;
  hipLaunchKernelGGL(bias_add_nhwc_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, bias, out, HW, C / 4, total4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" size_t csd_sum_pixels_scratch_bytes(int B, int HW, int C) {
  return (size_t)B * tn_chunks(B, HW, C) * C * sizeof(double) + 256;
}

// out[b][c] = sum over the pixels of x [B, HW, C]  (a batch reduction on top is csd_sum_rows: one serial sweep over batch x chunks
// in the final stage measured 10x slower than the two balanced stages)
extern "C" int csd_sum_pixels_nhwc(const float* x, float* out, int B, int HW, int C, void* scratch, void* stream) {
  CSD_REQUIRE(x && out && scratch && C % 4 == 0 && C <= 1024, "sum_pixels_nhwc: bad arguments (C=%d)", C);
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = tn_chunks(B, HW, C);
  const int rows = TN_THREADS / (C / 4);
  double* partial = static_cast<double*>(scratch);
  hipLaunchKernelGGL(sum_pixels_nhwc_kernel, dim3(nchunk, B), dim3(TN_THREADS), (size_t)rows * C * sizeof(double), s, x, partial, HW,
                     C, nchunk);
  CSD_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_pixels_final_kernel, dim3(cdiv(C, 32), B), dim3(256), 0, s, partial, out, C, nchunk);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_zero_insert_odd_nhwc(const float* dy, float* z, int B, int h, int w, int C, void* stream) {
  CSD_REQUIRE(dy && z && C % 4 == 0, "zero_insert_odd_nhwc: bad arguments");
  const size_t total4 = (size_t)B * 4 * h * w * C / 4;
  hipLaunchKernelGGL(zero_insert_odd_nhwc_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, z, h, w, C / 4, total4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_sumpool2_nhwc(const float* in, float* out, int B, int h, int w, int C, void* stream) {
  CSD_REQUIRE(in && out && C % 4 == 0, "sumpool2_nhwc: bad arguments");
  const size_t total4 = (size_t)B * h * w * C / 4;
  hipLaunchKernelGGL(sumpool2_nhwc_kernel, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, in, out, h, w, C / 4, total4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// attention core on a packed qkv tensor [B, L, 3C] (q at +0, k at +C, v at +2C): out [B, L, C]
extern "C" int csd_attention_nhwc(const float* qkv, float* out, int B, int L, int C, void* stream) {
  CSD_REQUIRE(qkv && out, "attention_nhwc: null argument");
  return attention_launch(qkv, 3 * C, out, B, L, C, (hipStream_t)stream);
}

// the same core in the arithmetic of a precision mode: CSD_PREC_F32 -> the fp32 MFMA kernel; F16X3 / F16F8 -> split fp16 operands
// (3 MFMAs per product, fp32-class); F16 -> plain fp16 operands (what csd_unet_forward runs in that mode)
extern "C" int csd_attention_nhwc_prec(const float* qkv, float* out, int B, int L, int C, int precision, void* stream) {
  CSD_REQUIRE(qkv && out, "attention_nhwc: null argument");
  const int ns = precision_ns(precision);
  if (!ns) return attention_launch(qkv, 3 * C, out, B, L, C, (hipStream_t)stream);
  return attention16_launch(qkv, 3 * C, out, B, L, C, ns >= 2 ? 2 : 1, (hipStream_t)stream);
}
