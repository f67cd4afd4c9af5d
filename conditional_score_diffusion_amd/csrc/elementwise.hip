// elementwise.hip - the small HBM-/latency-bound kernels around the contraction kernels:
// time embedding + dense layers, boundary layout changes (NCHW <-> NHWC), and the reference's
// two native ops (upfirdn2d, fused_bias_act) re-written for gfx950.
#include <algorithm>

#include "common.h"

namespace csd {

__device__ __forceinline__ float ew_act(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: return v / (1.0f + expf(-v));
    case CSD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case CSD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// ---- get_timestep_embedding (models/layers.py:524-538) ------------------------------------
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int B,
                                          int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim;
  float v = 0.f;   // odd dim: last column zero-padded
  if (j < 2 * half) {
    const int kidx = j < half ? j : j - half;
    // emb = log(10000)/(half-1) in Python double, then exp(arange * -emb), t*emb, sin/cos in fp32.
    // Each fp32 transcendental is evaluated in fp64 and rounded once (= correctly rounded), the
    // closest a different libm can get to the CPU reference's own <=1-ulp results: at t~999 one
    // ulp of the frequency already moves sin() by ~5e-5.
    const float e = (float)kidx * -(float)(9.210340371976184 / (double)(half - 1));
    const float w = (float)exp((double)e);
    const float arg = t[b] * w;
    v = j < half ? (float)sin((double)arg) : (float)cos((double)arg);
  }
  out[i] = v;
}

int timestep_embedding_launch(const float* t, float* out, int B, int dim, hipStream_t s) {
  CSD_REQUIRE(dim >= 4, "timestep_embedding: dim=%d too small", dim);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv(B * dim, 256)), dim3(256), 0, s, t, out, B, dim);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---- nn.Linear on [B, K] (temb MLP models/ddpm.py:155-159; Dense_0 models/layers.py:666) ----
// A workgroup = 64 samples x 16 output features: lane = sample, wave w owns features 4w .. 4w+3.  The (activated) inputs are staged
// transposed in LDS ([k][sample], K in chunks of 256), the weight rows are wave-uniform (scalar loads), every output is ONE sequential
// fp32 sum over k - the same bits whatever the batch size.  (The first version ran one wave per (feature, sample): 405 k waves for the
// 6336 Dense_0 columns of SR3-160 at B = 64, 168 us per evaluation.)
#define LIN_KC 256
// VEC: K % 4 == 0 (rows 16-byte aligned: every net of the reference) - float4 loads; else four clamped dword loads per item.
// Straight-line staging: clamped addresses + selects, no branch around a load (a guarded load sits in its own basic block behind an
// s_waitcnt: the 20 loads of a chunk would run one round trip each instead of all in flight).
template <bool VEC>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int B, int K, int N, int act_in, int act_out) {
  __shared__ float xs[LIN_KC][65];                               // [k][sample] (+1: the transposing store is conflict-free)
  __shared__ __attribute__((aligned(16))) float ws[16][LIN_KC];  // [feature][k]: read as wave-wide broadcasts
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * 64;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  auto load4 = [&](const float* row, int k, int kc) __attribute__((always_inline)) {      // row[k .. k+3], zeros from kc on
    float4 v;
    if (VEC) {
      v = *reinterpret_cast<const float4*>(row + (k < kc ? k : 0));
    } else {
      v.x = row[min(k, kc - 1)]; v.y = row[min(k + 1, kc - 1)]; v.z = row[min(k + 2, kc - 1)]; v.w = row[min(k + 3, kc - 1)];
    }
    v.x = k < kc ? v.x : 0.f; v.y = k + 1 < kc ? v.y : 0.f; v.z = k + 2 < kc ? v.z : 0.f; v.w = k + 3 < kc ? v.w : 0.f;
    return v;
  };
  for (int k0 = 0; k0 < K; k0 += LIN_KC) {
    const int kc = min(LIN_KC, K - k0);
    __syncthreads();
    float4 xv[16], wv4[4];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + u * 256;                  // float4 index: sample bb, k4 (k fastest)
      const int bb = i / (LIN_KC / 4), k = (i % (LIN_KC / 4)) * 4;
      xv[u] = load4(in + (size_t)min(b0 + bb, B - 1) * K + k0, k, kc);      // (samples past B: a valid row, never stored)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      const int f = i / (LIN_KC / 4), k = (i % (LIN_KC / 4)) * 4;
      wv4[u] = load4(W + (size_t)min(n0 + f, N - 1) * K + k0, k, kc);
    }
    if (act_in == CSD_ACT_SWISH) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        xv[u].x = ew_act(xv[u].x, CSD_ACT_SWISH); xv[u].y = ew_act(xv[u].y, CSD_ACT_SWISH);
        xv[u].z = ew_act(xv[u].z, CSD_ACT_SWISH); xv[u].w = ew_act(xv[u].w, CSD_ACT_SWISH);
      }
    } else if (act_in != CSD_ACT_NONE) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        xv[u].x = ew_act(xv[u].x, act_in); xv[u].y = ew_act(xv[u].y, act_in);
        xv[u].z = ew_act(xv[u].z, act_in); xv[u].w = ew_act(xv[u].w, act_in);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + u * 256;
      const int bb = i / (LIN_KC / 4), k = (i % (LIN_KC / 4)) * 4;
      xs[k][bb] = xv[u].x; xs[k + 1][bb] = xv[u].y; xs[k + 2][bb] = xv[u].z; xs[k + 3][bb] = xv[u].w;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      *reinterpret_cast<float4*>(&ws[i / (LIN_KC / 4)][(i % (LIN_KC / 4)) * 4]) = wv4[u];
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kc; k += 4) {
      float4 wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const float4*>(&ws[wave * 4 + q][k]);
      const float x0 = xs[k][lane], x1 = xs[k + 1][lane], x2 = xs[k + 2][lane], x3 = xs[k + 3][lane];
#pragma unroll
      for (int q = 0; q < 4; ++q) {               // one sequential fp32 sum over k per output
        acc[q] += x0 * wv[q].x;
        acc[q] += x1 * wv[q].y;
        acc[q] += x2 * wv[q].z;
        acc[q] += x3 * wv[q].w;
      }
    }
  }
  const int b = b0 + lane;
  if (b < B) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wave * 4 + q;
      if (n < N) out[(size_t)b * N + n] = ew_act(acc[q] + (bias ? bias[n] : 0.f), act_out);
    }
  }
}

int linear_launch(const float* in, const float* W, const float* bias, float* out, int B, int K, int N,
                  int act_in, hipStream_t s, int act_out) {
  const dim3 grid(cdiv(N, 16), cdiv(B, 64));
  if (K % 4 == 0) hipLaunchKernelGGL(linear_kernel<true>, grid, dim3(256), 0, s, in, W, bias, out, B, K, N, act_in, act_out);
  else hipLaunchKernelGGL(linear_kernel<false>, grid, dim3(256), 0, s, in, W, bias, out, B, K, N, act_in, act_out);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---- network input: cat(x, y [+ sigma*z]) , 2v-1, NCHW -> NHWC padded to Cpad channels --------
// (models/ddpm.py:163-168,283; sampling/conditional.py:104-110)
__global__ void assemble_input_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                      const float* __restrict__ yn, float ysig, float* __restrict__ out,
                                      int Cx, int Cy, int HW, int Cpad, int centered, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / Cpad;             // b*HW + p
    const int c = (int)(i - pix * Cpad);
    const size_t b = pix / HW;
    const size_t p = pix - b * HW;
    float v = 0.f;
    if (c < Cx) {
      v = x[(b * Cx + c) * HW + p];
      if (!centered) v = 2.f * v - 1.f;
    } else if (c < Cx + Cy) {
      const size_t j = (b * Cy + (c - Cx)) * HW + p;
      v = y[j];
      if (yn) v = v + yn[j] * ysig;
      if (!centered) v = 2.f * v - 1.f;
    }
    out[i] = v;
  }
}

int assemble_input_launch(const float* x, const float* y, const float* y_noise, float y_sigma, float* out,
                          int B, int Cx, int Cy, int HW, int Cpad, int centered, hipStream_t s) {
  const size_t total = (size_t)B * HW * Cpad;
  const int grid = (int)std::min<size_t>(cdiv64(total, 256), 8192);
  hipLaunchKernelGGL(assemble_input_kernel, dim3(grid), dim3(256), 0, s, x, y, y_noise, y_sigma, out, Cx, Cy,
                     HW, Cpad, centered, total);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---- layout changes at the per-operator C-ABI boundary (LDS-tiled transposes) -----------------
// in [B, C, HW] -> out [B, HW, ld]: channels [0, Cw) are written (zeros beyond C)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, int HW, int Cpad, int ld) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? in[((size_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int p = p0 + j, c = c0 + tx;
    if (p < HW && c < Cpad) out[((size_t)b * HW + p) * ld + c] = tile[tx][j];
  }
}

int nchw_to_nhwc_launch(const float* in, float* out, int B, int C, int HW, int Cw, int ld, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(Cw, 32), B), dim3(256), 0, s, in, out, C,
                     HW, Cw, ld);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// in [B, HW, Cstride] (first C channels) -> out [B, C, HW]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, int HW, int Cstride) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (c < C && p < HW) ? in[((size_t)b * HW + p) * Cstride + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) out[((size_t)b * C + c) * HW + p] = tile[tx][j];
  }
}

int nhwc_to_nchw_launch(const float* in, float* out, int B, int C, int HW, int Cstride, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, s, in, out, C, HW,
                     Cstride);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---- resampling without convolution (resamp_with_conv=False: models/layers.py:601,626) --------
__global__ void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C,
                                size_t total4) {
  const int C4 = C >> 2, OH = H >> 1, OW = W >> 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const size_t b = r / OH;
    const float4* p = reinterpret_cast<const float4*>(in) + ((b * H + 2 * oy) * W + 2 * ox) * C4 + c4;
    const float4 a = p[0], bq = p[C4], c = p[(size_t)W * C4], d = p[(size_t)W * C4 + C4];
    float4 o;
    o.x = (a.x + bq.x + c.x + d.x) * 0.25f;
    o.y = (a.y + bq.y + c.y + d.y) * 0.25f;
    o.z = (a.z + bq.z + c.z + d.z) * 0.25f;
    o.w = (a.w + bq.w + c.w + d.w) * 0.25f;
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

int avgpool2_launch(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  const size_t total4 = (size_t)B * (H / 2) * (W / 2) * C / 4;
  const int grid = (int)std::min<size_t>(cdiv64(total4, 256), 8192);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid), dim3(256), 0, s, in, out, H, W, C, total4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

__global__ void nearest_up2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                        int C, size_t total4) {
  const int C4 = C >> 2, OH = H * 2, OW = W * 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const size_t b = r / OH;
    reinterpret_cast<float4*>(out)[i] =
        reinterpret_cast<const float4*>(in)[((b * H + (oy >> 1)) * W + (ox >> 1)) * C4 + c4];
  }
}

int nearest_up2_nhwc_launch(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
  const size_t total4 = (size_t)B * H * 2 * W * 2 * C / 4;
  const int grid = (int)std::min<size_t>(cdiv64(total4, 256), 8192);
  hipLaunchKernelGGL(nearest_up2_nhwc_kernel, dim3(grid), dim3(256), 0, s, in, out, H, W, C, total4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---- nearest x2 on NCHW (C-ABI csd_nearest_up2) ---------------------------------------------------
__global__ void nearest_up2_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                        size_t total) {
  const int OW = 2 * W, OH = 2 * H;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    size_t r = i / OW;
    const int oy = (int)(r % OH);
    const size_t nc = r / OH;
    out[i] = in[(nc * H + (oy >> 1)) * W + (ox >> 1)];
  }
}

// ---- upfirdn2d (op/upfirdn2d_kernel.cu:107-207; semantics of op/upfirdn2d.py:161-202) --------------
// out[n,c,oy,ox] = sum_{ky,kx} xpad_up[oy*down_y + ky, ox*down_x + kx] * kflip[ky,kx], where xpad_up
// is x zero-stuffed by `up` and padded by (pad0, pad1) (negative pads crop).  One workgroup computes
// a 16x64 output tile of one (n,c) plane; the source window and the flipped FIR taps are staged in
// LDS; fp32 accumulate (the reference's half path accumulates in half - SURVEY.md App. B).
#define UFD_TH 16
#define UFD_TW 64
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const float* __restrict__ x, const float* __restrict__ kern,
                                                        float* __restrict__ out, int H, int W, int OH, int OW,
                                                        int kh, int kw, int upx, int upy, int dnx, int dny,
                                                        int px0, int py0, int tiles_x) {
  extern __shared__ float sm[];
  float* sk = sm;                 // [kh*kw] flipped taps
  float* sx = sm + kh * kw;       // source window
  const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
  const size_t plane = blockIdx.y;
  const int oy0 = tile_y * UFD_TH, ox0 = tile_x * UFD_TW;
  // window of source rows/cols that can contribute to this tile
  auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
  const int my0 = oy0 * dny - py0, mx0 = ox0 * dnx - px0;             // first up-sampled coord
  const int iy0 = fdiv(my0, upy);                                    // floor: safe lower bound
  const int ix0 = fdiv(mx0, upx);
  const int wh = ((UFD_TH - 1) * dny + kh - 1) / upy + 2;
  const int ww = ((UFD_TW - 1) * dnx + kw - 1) / upx + 2;
  for (int i = threadIdx.x; i < kh * kw; i += 256) sk[i] = kern[kh * kw - 1 - i];
  const float* xp = x + plane * H * W;
  for (int i = threadIdx.x; i < wh * ww; i += 256) {
    const int ry = i / ww, rx = i - ry * ww;
    const int iy = iy0 + ry, ix = ix0 + rx;
    sx[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xp[(size_t)iy * W + ix] : 0.f;
  }
  __syncthreads();
  float* op = out + plane * OH * OW;
  for (int i = threadIdx.x; i < UFD_TH * UFD_TW; i += 256) {
    const int ty = i / UFD_TW, tx = i - ty * UFD_TW;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= OH || ox >= OW) continue;
    float acc = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int my = oy * dny + ky - py0;          // coordinate in the zero-stuffed image
      if (my < 0 || my % upy != 0) continue;
      const int iy = my / upy - iy0;
      for (int kx = 0; kx < kw; ++kx) {
        const int mx = ox * dnx + kx - px0;
        if (mx < 0 || mx % upx != 0) continue;
        const int ix = mx / upx - ix0;
        acc += sx[iy * ww + ix] * sk[ky * kw + kx];
      }
    }
    op[(size_t)oy * OW + ox] = acc;
  }
}

// ---- fused_bias_act (op/fused_bias_act_kernel.cu:18-49) -----------------------------------------
__global__ void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                      const float* __restrict__ ref, float* __restrict__ out, size_t numel,
                                      int C, size_t inner, int act, int grad, float alpha, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel;
       i += (size_t)gridDim.x * blockDim.x) {
    float v = x[i];
    if (b) v += b[(i / inner) % C];
    const float r = ref ? ref[i] : 0.f;
    float y;
    if (act == 3) {   // leaky relu
      if (grad == 0) y = (v > 0.f ? v : v * alpha) * scale;
      else if (grad == 1) y = (r > 0.f ? v : v * alpha) * scale;
      else y = 0.f;
    } else {          // linear
      y = (grad == 2) ? 0.f : v * scale;
    }
    out[i] = y;
  }
}

// ---- FIR resampling of an NHWC tensor by 2 with a separable 4-tap kernel: upsample_2d / downsample_2d of
// models/up_or_down_sampling.py:196-257 (= upfirdn2d with up = 2, pad (2, 1) / down = 2, pad (1, 1)); used inside the
// NCSN++ graph executor, where activations never leave the NHWC layout ----
struct Fir16 { float k[16]; };      // the 2-D kernel as upfirdn2d receives it (gain applied), row-major
__global__ void fir_resample_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int C, int up,
                                         Fir16 f, size_t total) {
  const int OH = up ? H * 2 : H / 2, OW = up ? W * 2 : W / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t b = r / OH;
    const float* src = in + b * (size_t)H * W * C + c;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      // coordinate in the zero-stuffed (up) or plain (down) image; upfirdn2d correlates with the FLIPPED kernel
      const int my = up ? oy + ky - 2 : oy * 2 + ky - 1;
      if (my < 0 || (up && (my & 1))) continue;
      const int iy = up ? my >> 1 : my;
      if (iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int mx = up ? ox + kx - 2 : ox * 2 + kx - 1;
        if (mx < 0 || (up && (mx & 1))) continue;
        const int ix = up ? mx >> 1 : mx;
        if (ix >= W) continue;
        acc += src[((size_t)iy * W + ix) * C] * f.k[(3 - ky) * 4 + (3 - kx)];
      }
    }
    out[i] = acc;
  }
}
// float4-over-channels variant (C % 4 == 0): one thread per (output pixel, 4 channels)
__global__ void fir_resample_nhwc4_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int C4, int up,
                                          Fir16 f, size_t total) {
  const int OH = up ? H * 2 : H / 2, OW = up ? W * 2 : W / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    size_t r = i / C4;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t b = r / OH;
    const float4* src = in + b * (size_t)H * W * C4 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int my = up ? oy + ky - 2 : oy * 2 + ky - 1;
      if (my < 0 || (up && (my & 1))) continue;
      const int iy = up ? my >> 1 : my;
      if (iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int mx = up ? ox + kx - 2 : ox * 2 + kx - 1;
        if (mx < 0 || (up && (mx & 1))) continue;
        const int ix = up ? mx >> 1 : mx;
        if (ix >= W) continue;
        const float4 v = src[((size_t)iy * W + ix) * C4];
        const float w = f.k[(3 - ky) * 4 + (3 - kx)];
        acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
      }
    }
    out[i] = acc;
  }
}
// the up / down ResnetBlockBigGANpp's two resampled tensors in ONE pass over x (layerspp.py:242-260): out_x = FIR(x) and
// out_h = FIR(act(x * scale + shift)) - the GroupNorm'ed, activated tensor is never materialised (it used to be written, read by its FIR pass,
// and x read a second time by the other: 3 launches, 2.2 x the bytes).  Same taps, same accumulation order as the single-tensor kernel.
__global__ void fir_resample2_nhwc4_kernel(const float4* __restrict__ in, const float* __restrict__ nscale, const float* __restrict__ nshift,
                                           float4* __restrict__ out_x, float4* __restrict__ out_h, int H, int W, int C4, int up, int act,
                                           Fir16 f, size_t total) {
  const int OH = up ? H * 2 : H / 2, OW = up ? W * 2 : W / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    size_t r = i / C4;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t b = r / OH;
    const float4* src = in + b * (size_t)H * W * C4 + c;
    const float4 sc = *reinterpret_cast<const float4*>(nscale + (b * C4 + c) * 4);
    const float4 sh = *reinterpret_cast<const float4*>(nshift + (b * C4 + c) * 4);
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), ah = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int my = up ? oy + ky - 2 : oy * 2 + ky - 1;
      if (my < 0 || (up && (my & 1))) continue;
      const int iy = up ? my >> 1 : my;
      if (iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int mx = up ? ox + kx - 2 : ox * 2 + kx - 1;
        if (mx < 0 || (up && (mx & 1))) continue;
        const int ix = up ? mx >> 1 : mx;
        if (ix >= W) continue;
        const float4 v = src[((size_t)iy * W + ix) * C4];
        const float w = f.k[(3 - ky) * 4 + (3 - kx)];
        ax.x += v.x * w; ax.y += v.y * w; ax.z += v.z * w; ax.w += v.w * w;
        // (SiLU in the conv prologues' form - v_exp_f32 + v_rcp_f32, 1 ulp each: with libm's expf and an IEEE division the 16 taps of an
        // output made the pass vector-bound: 739 us at 160^2 for 190 us of HBM traffic)
        auto actf = [&](float t) __attribute__((always_inline)) {
          return act == CSD_ACT_SWISH ? t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)) : ew_act(t, act);
        };
        const float hx = actf(v.x * sc.x + sh.x), hy = actf(v.y * sc.y + sh.y);
        const float hz = actf(v.z * sc.z + sh.z), hw = actf(v.w * sc.w + sh.w);
        ah.x += hx * w; ah.y += hy * w; ah.z += hz * w; ah.w += hw * w;
      }
    }
    out_x[i] = ax;
    out_h[i] = ah;
  }
}
// LDS-tiled form of the same pass for the UPSAMPLING blocks (C % 16 == 0, whole 16 x 16 output tiles): a workgroup stages the raw 10 x 10 input
// patch x 16 channels, activates every patch element ONCE (the per-output form recomputes the SiLU for each of an element's 16 uses and stays
// vector-bound: 713 us at 80^2 -> 160^2 against 372 here), then every thread gathers its taps of both tensors from LDS.  Same taps, same
// accumulation order.  (Downsampling measured faster on the per-output kernel - 467 against 563 us - whose taps already hit L1.)
__global__ __launch_bounds__(256) void fir_upsample2_tiled_kernel(const float4* __restrict__ in, const float* __restrict__ nscale,
                                                                  const float* __restrict__ nshift, float4* __restrict__ out_x,
                                                                  float4* __restrict__ out_h, int H, int W, int C4, int act, Fir16 f) {
  constexpr int TO = 16;                                // output tile
  constexpr int PH = TO / 2 + 2;                        // input patch (rows = columns)
  __shared__ float4 raw[PH * PH * 4], actv[PH * PH * 4];
  const int OH = H * 2, OW = W * 2;
  const int tiles_x = OW / TO, tiles_y = OH / TO, cgs = C4 / 4;
  int t = blockIdx.x;
  const int cg = t % cgs; t /= cgs;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const size_t b = t / tiles_y;
  const int oy0 = ty * TO, ox0 = tx * TO;
  const int iy0 = oy0 / 2 - 1, ix0 = ox0 / 2 - 1;
  const int tid = threadIdx.x;
  const int c4 = tid & 3;                               // this thread's float4 of the 16-channel group (patch staging AND outputs)
  const float4* src = in + b * (size_t)H * W * C4 + cg * 4 + c4;
  const float4 sc = *reinterpret_cast<const float4*>(nscale + (b * C4 + cg * 4 + c4) * 4);
  const float4 sh = *reinterpret_cast<const float4*>(nshift + (b * C4 + cg * 4 + c4) * 4);
  auto actf = [&](float v) __attribute__((always_inline)) {
    return act == CSD_ACT_SWISH ? v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)) : ew_act(v, act);
  };
  for (int p = tid >> 2; p < PH * PH; p += 64) {
    const int py = p / PH, px = p - py * PH;
    const int iy = iy0 + py, ix = ix0 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), h = v;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {      // (outside the image both tensors contribute nothing)
      v = src[((size_t)iy * W + ix) * C4];
      h = make_float4(actf(v.x * sc.x + sh.x), actf(v.y * sc.y + sh.y), actf(v.z * sc.z + sh.z), actf(v.w * sc.w + sh.w));
    }
    raw[p * 4 + c4] = v;
    actv[p * 4 + c4] = h;
  }
  __syncthreads();
  for (int o = tid >> 2; o < TO * TO; o += 64) {
    const int ly = o / TO, lx = o - ly * TO;
    const int oy = oy0 + ly, ox = ox0 + lx;
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), ah = ax;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int my = oy + ky - 2;
      if (my < 0 || (my & 1)) continue;
      const int iy = my >> 1;
      if (iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int mx = ox + kx - 2;
        if (mx < 0 || (mx & 1)) continue;
        const int ix = mx >> 1;
        if (ix >= W) continue;
        const int p = (iy - iy0) * PH + (ix - ix0);
        const float4 v = raw[p * 4 + c4], h = actv[p * 4 + c4];
        const float w = f.k[(3 - ky) * 4 + (3 - kx)];
        ax.x += v.x * w; ax.y += v.y * w; ax.z += v.z * w; ax.w += v.w * w;
        ah.x += h.x * w; ah.y += h.y * w; ah.z += h.z * w; ah.w += h.w * w;
      }
    }
    const size_t oi = ((b * OH + oy) * OW + ox) * C4 + cg * 4 + c4;
    out_x[oi] = ax;
    out_h[oi] = ah;
  }
}

int fir_resample2_nhwc_launch(const float* in, const float* nscale, const float* nshift, float* out_x, float* out_h, int B, int H, int W,
                              int C, const float* taps4, int up, int act, hipStream_t s) {
  CSD_REQUIRE(C % 4 == 0, "fir_resample2: C = %d is not a multiple of 4", C);
  Fir16 f;
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) { f.k[a * 4 + b] = taps4[a] * taps4[b]; sum += f.k[a * 4 + b]; }
  for (int a = 0; a < 16; ++a) f.k[a] = f.k[a] / sum * (up ? 4.f : 1.f);
  const size_t total = (size_t)B * (up ? H * 2 : H / 2) * (up ? W * 2 : W / 2) * C;
  if (up && C % 16 == 0 && (H * 2) % 16 == 0 && (W * 2) % 16 == 0) {
    const size_t nwg = (size_t)B * (H * 2 / 16) * (W * 2 / 16) * (C / 16);
    hipLaunchKernelGGL(fir_upsample2_tiled_kernel, dim3((unsigned)nwg), dim3(256), 0, s, reinterpret_cast<const float4*>(in), nscale, nshift,
                       reinterpret_cast<float4*>(out_x), reinterpret_cast<float4*>(out_h), H, W, C / 4, act, f);
    CSD_LAUNCH_CHECK();
    return CSD_OK;
  }
  hipLaunchKernelGGL(fir_resample2_nhwc4_kernel, dim3((unsigned)std::min<size_t>(cdiv64(total / 4, 256), 65536)), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(in), nscale, nshift, reinterpret_cast<float4*>(out_x), reinterpret_cast<float4*>(out_h),
                     H, W, C / 4, up, act, f, total / 4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int fir_resample_nhwc_launch(const float* in, float* out, int B, int H, int W, int C, const float* taps4, int up, hipStream_t s,
                             float scale, int flip) {
  // _setup_kernel (up_or_down_sampling.py:181-189): outer product, normalised, times the gain (factor^2 when upsampling)
  Fir16 f;
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) { f.k[a * 4 + b] = taps4[flip ? 3 - a : a] * taps4[flip ? 3 - b : b]; sum += f.k[a * 4 + b]; }
  for (int a = 0; a < 16; ++a) f.k[a] = f.k[a] / sum * (up ? 4.f : 1.f) * scale;
  const size_t total = (size_t)B * (up ? H * 2 : H / 2) * (up ? W * 2 : W / 2) * C;
  if (C % 4 == 0) {
    hipLaunchKernelGGL(fir_resample_nhwc4_kernel, dim3((unsigned)std::min<size_t>(cdiv64(total / 4, 256), 65536)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), H, W, C / 4, up, f, total / 4);
  } else {
    hipLaunchKernelGGL(fir_resample_nhwc_kernel, dim3((unsigned)std::min<size_t>(cdiv64(total, 256), 65536)), dim3(256), 0, s, in,
                       out, H, W, C, up, f, total);
  }
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---- small per-op kernels for graphs orchestrated above the C ABI (NCSN++: models/ncsnpp.py of the reference) ----
// Gaussian Fourier features (models/layerspp.py:32-41): out[b] = [sin(a_bk), cos(a_bk)], a = ((t*W)*2)*pi evaluated
// in fp32 in that order (the argument reaches ~1e3, so its fp32 rounding is part of the result), sin/cos in fp64
__global__ void fourier_embedding_kernel(const float* __restrict__ t, const float* __restrict__ W, float* __restrict__ out,
                                         int B, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, k = i - b * E;
  float a = t[b] * W[k];
  a = a * 2.0f;
  a = a * 3.14159265358979323846f;
  out[(size_t)b * 2 * E + k] = (float)sin((double)a);
  out[(size_t)b * 2 * E + E + k] = (float)cos((double)a);
}
int fourier_embedding_launch(const float* t, const float* W, float* out, int B, int E, hipStream_t s) {
  hipLaunchKernelGGL(fourier_embedding_kernel, dim3(cdiv(B * E, 256)), dim3(256), 0, s, t, W, out, B, E);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// out = (alpha*a + beta*b + gamma) * post   (b may be null): residual adds, (x + h)/sqrt(2), Combine 'sum', 2x - 1
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, float alpha,
                             float beta, float gamma, float post, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = alpha * a[i];
    if (b) v += beta * b[i];
    out[i] = (v + gamma) * post;
  }
}
int axpby_launch(const float* a, const float* b, float* out, float alpha, float beta, float gamma, float post, size_t n,
                 hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)std::min<size_t>(cdiv64(n, 256), 65536)), dim3(256), 0, s, a, b, out, alpha,
                     beta, gamma, post, n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// out[b,c,:] = act(x[b,c,:] + bias[b*bias_stride + c])   (h += Dense_0(act(temb))[:, :, None, None])
__global__ void bias_add_nchw_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out,
                                     int C, size_t inner, int bias_stride, int act, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t bc = i / inner;
    const size_t b = bc / C;
    const int c = (int)(bc - b * C);
    out[i] = ew_act(x[i] + bias[b * bias_stride + c], act);
  }
}
int bias_add_nchw_launch(const float* x, const float* bias, float* out, int B, int C, size_t inner, int bias_stride, int act,
                         hipStream_t s) {
  const size_t n = (size_t)B * C * inner;
  hipLaunchKernelGGL(bias_add_nchw_kernel, dim3((unsigned)std::min<size_t>(cdiv64(n, 256), 65536)), dim3(256), 0, s, x, bias,
                     out, C, inner, bias_stride, act, n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

}  // namespace csd

// =================================================================================================
// C ABI for the stand-alone operators
// =================================================================================================
using namespace csd;

extern "C" int csd_upfirdn2d(const float* x, const float* kernel, float* out, int N, int C, int H, int W, int kh,
                             int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                             int pad_y0, int pad_y1, void* stream) {
  CSD_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "upfirdn2d: up/down must be >= 1");
  const int OH = (H * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  const int OW = (W * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  CSD_REQUIRE(OH >= 1 && OW >= 1, "upfirdn2d: empty output");
  const int tiles_x = cdiv(OW, UFD_TW), tiles_y = cdiv(OH, UFD_TH);
  const int wh = ((UFD_TH - 1) * down_y + kh - 1) / up_y + 2;
  const int ww = ((UFD_TW - 1) * down_x + kw - 1) / up_x + 2;
  const size_t lds = (size_t)(kh * kw + wh * ww) * sizeof(float);
  CSD_REQUIRE(lds <= 64 * 1024, "upfirdn2d: filter/window too large for the tiled kernel");
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3(tiles_x * tiles_y, N * C), dim3(256), lds, (hipStream_t)stream, x,
                     kernel, out, H, W, OH, OW, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, tiles_x);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_fused_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t numel,
                                  int C, int64_t inner, int act, int grad, float alpha, float scale,
                                  void* stream) {
  CSD_REQUIRE(act == 1 || act == 3, "fused_bias_act: act must be 1 (linear) or 3 (lrelu)");
  CSD_REQUIRE(grad >= 0 && grad <= 2, "fused_bias_act: grad must be 0..2");
  if (numel == 0) return CSD_OK;
  const int grid = (int)std::min<int64_t>(cdiv64(numel, 256), 8192);
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, ref, out,
                     (size_t)numel, C, (size_t)inner, act, grad, alpha, scale);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_nearest_up2(const float* x, float* out, int N, int C, int H, int W, void* stream) {
  const size_t total = (size_t)N * C * 4 * H * W;
  if (total == 0) return CSD_OK;
  const int grid = (int)std::min<size_t>(cdiv64(total, 256), 8192);
  hipLaunchKernelGGL(nearest_up2_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, H, W, total);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_timestep_embedding(const float* t, float* out, int B, int dim, void* stream) {
  return timestep_embedding_launch(t, out, B, dim, (hipStream_t)stream);
}
