// backward.hip - gradient kernels of the score network's layers (the autograd backward
// of conv / NIN / Linear / GroupNorm(+act) / attention / dropout that the reference gets from torch) behind the
// NCHW fp32 C ABI of include/csd.h.  The data gradient of a convolution is itself a convolution with the flipped,
// transposed weight and runs on the forward kernels (csd_conv2d); what is new here:
//   conv_wgrad_kernel   dW = dY^T (x) X as an implicit GEMM over pixels on the fp32 matrix cores, split-K over
//                       workgroups with a deterministic two-stage reduction
//   gn_bwd_kernel       GroupNorm(+activation) backward, one workgroup per (sample, group), fp64 statistics
//   bgemm_kernel        strided batched fp32 GEMM (attention backward, Linear backward)
//   softmax / dsoftmax rows, row / column sums, activation fwd/bwd, dropout (Philox4x32-10), elementwise product
#include <stdlib.h>

#include <algorithm>

#include "common.h"

using namespace csd;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

inline size_t al64(size_t v) { return (v + 63) / 64 * 64; }

__device__ __forceinline__ float bw_act(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: return v / (1.0f + expf(-v));
    case CSD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case CSD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
// d act(v) / dv
__device__ __forceinline__ float bw_dact(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: { const float s = 1.0f / (1.0f + expf(-v)); return s * (1.0f + v * (1.0f - s)); }
    case CSD_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? 1.f : 0.2f;
    case CSD_ACT_ELU: return v > 0.f ? 1.f : expf(v);
    default: return 1.f;
  }
}

__device__ __forceinline__ double block_sum(double v, double* sh) {   // 256 threads; result broadcast to all
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------------------------
// convolution weight gradient.  x [B, IH, IW, Cs] NHWC (Cin real channels), dy [B, OH, OW, Cout] NHWC.
// GEMM: M = 32 couts (A = dY), N = 32 cins (B = X shifted by the tap), K = pixels, v_mfma_f32_32x32x2_f32
// (2 pixels per instruction).  A wave keeps the 32x32 tile of EVERY tap (TAPS x 16 accumulators): dY is read once
// per pixel, X once per tap.  Workgroup = 4 waves on interleaved pixel pairs of one K-split, reduced through LDS;
// partial [split][Cout][Cin][TAPS] is summed in split order by wgrad_reduce_kernel (bit-reproducible).
// ---------------------------------------------------------------------------------------------------------------
template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ partial, int B, int IH, int IW, int Cs,
                                                         int Cin, int OH, int OW, int Cout, int stride, int pad, int up,
                                                         int per_split, int n_ci, int n_co) {
  constexpr int KS = TAPS == 9 ? 3 : 1;
  __shared__ float red[TAPS * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, kk = lane >> 5;
  // XCD-aware work order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (one L2 each).  The
  // (cout tile, cin tile) workgroups of one K-split read the same rows of x and dy, so they are made neighbours ON ONE XCD:
  // work item w = (id % 8) * per_xcd + id / 8, split = w / tiles, tile = w % tiles.
  const unsigned tiles = (unsigned)n_ci * n_co, total = gridDim.x;
  const unsigned per_xcd = (total + 7) / 8;
  unsigned wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((total & 7) != 0) wid = blockIdx.x;            // (grids that do not fill the 8 XCDs evenly keep the plain order)
  const unsigned split_id = wid / tiles, tile_id = wid - split_id * tiles;
  const int tz = (int)(tile_id / n_ci), ty = (int)(tile_id - (unsigned)tz * n_ci);
  const int co = tz * 32 + m, ci = ty * 32 + m;
  const bool cov = co < Cout, civ = ci < Cin;
  // K runs over output rows r = b*OH + oy (a K-split is a run of rows; wave w takes rows w, w+4, ...) and, inside a row,
  // over pixel pairs: all addresses advance by constants, so a step costs ~60 VALU instructions next to 9 MFMAs.
  const unsigned nrows = (unsigned)B * OH;
  const unsigned r_begin = split_id * (unsigned)per_split;
  const unsigned r_end = r_begin + per_split < nrows ? r_begin + per_split : nrows;
  const int EH = IH << up, EW = IW << up;
  floatx16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const unsigned cio = civ ? ci : 0, coo = cov ? co : 0;

  for (unsigned r = r_begin + wave; r < r_end; r += 4) {
    const unsigned b = r / (unsigned)OH, oy = r % (unsigned)OH;
    unsigned rowoff[KS];
    bool rowok[KS];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int iy = (int)oy * stride + ky - pad;
      rowok[ky] = civ && iy >= 0 && iy < EH;
      rowoff[ky] = ((b * (unsigned)IH + (rowok[ky] ? (unsigned)(iy >> up) : 0u)) * (unsigned)IW) * (unsigned)Cs + cio;
    }
    const float* dyp = dy + (size_t)(r * (unsigned)OW) * Cout + coo;
    // U pixel pairs per step: all their operand loads (U*(1+TAPS) dwords per lane, unconditional on clamped addresses) are
    // issued before the first MFMA, so ~10 KB per wave is in flight - the stream comes from HBM/L2 with ~2 us latency and
    // the second wave of the SIMD covers it with its own U*TAPS MFMAs.
    constexpr int U = 4;
    for (int ox0 = 0; ox0 < OW; ox0 += 2 * U) {
      float a[U], bv[U][TAPS];
      bool va[U], vb[U][TAPS];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ox = ox0 + 2 * u + kk;
        const bool v = ox < OW;
        a[u] = dyp[(unsigned)(v ? ox : 0) * (unsigned)Cout];
        va[u] = v && cov;
        unsigned xo[KS];
        bool okx[KS];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int ix = ox * stride + kx - pad;
          okx[kx] = v && ix >= 0 && ix < EW;
          xo[kx] = okx[kx] ? (unsigned)(ix >> up) * (unsigned)Cs : 0u;
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          bv[u][t] = x[(size_t)(rowoff[t / KS] + xo[t % KS])];
          vb[u][t] = rowok[t / KS] && okx[t % KS];
        }
      }
      __builtin_amdgcn_sched_barrier(0);          // keep every load of the step ahead of its MFMAs
      // masks first, then U*TAPS matrix instructions back to back (a v_cndmask in front of every MFMA held the pipe back)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = va[u] ? a[u] : 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) bv[u][t] = vb[u][t] ? bv[u][t] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u][t], acc[t], 0, 0, 0);
      }
    }
  }

  // cross-wave reduction (fixed order: wave 0, 1, 2, 3)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* d = &red[(t * 16 + r) * 64 + lane];
          *d = (w == 0) ? acc[t][r] : *d + acc[t][r];
        }
    }
    __syncthreads();
  }
  // partial[split][co][ci][tap]; accumulator r of lane holds D[row = (r&3) + 8*(r>>2) + 4*kk][col = m]
  float* dst = partial + (size_t)split_id * Cout * Cin * TAPS;
  for (int e = threadIdx.x; e < TAPS * 1024; e += 256) {
    const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
    const int rco = tz * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int rci = ty * 32 + (l & 31);
    if (rco < Cout && rci < Cin) dst[((size_t)rco * Cin + rci) * TAPS + t] = red[e];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide-load schedule of the weight gradient (Cin % 4 == 0).  The 32x32x9 kernel above turned out to be bound by the texture
// addresser's instruction rate (40 dword wave-loads per 36 MFMAs, ~16 cycles each, 8 waves per CU), not by MFMA or bandwidth.
// Here a lane loads a float4 = 4 consecutive input channels of one pixel: ONE 1 KiB wave-load feeds FOUR MFMAs (MFMA j takes
// element j, so column n of tile j is input channel 4n + j), i.e. a wave covers 32 couts x 128 cins.  To keep the accumulators
// in registers the taps are split over workgroups by kernel row ky: a wave holds KS x 4 tiles (192 accumulators for 3x3).
// One wave per SIMD, software-pipelined: the 4 + 12 loads of the next step (U = 4 pixel pairs, 13 KiB) are in flight while
// the 48 MFMAs (3072 cycles) of the current step run.
// ---------------------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void conv_wgrad4_kernel(const float* x, const float* dy,   // (not __restrict__: see the loop)
                                                          float* partial, int B, int IH, int IW, int Cs,
                                                          int Cin, int OH, int OW, int Cout, int stride, int pad, int up,
                                                          int per_split, int n_ci, int n_co) {
  constexpr int TAPS = KS * KS, U = 4;
  __shared__ float red[KS * 4 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, kk = lane >> 5;
  // XCD-aware order (see conv_wgrad_kernel): the tiles of one K-split are neighbours on one XCD
  const unsigned tiles = (unsigned)n_ci * n_co * KS, total = gridDim.x;
  const unsigned per_xcd = (total + 7) / 8;
  unsigned wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((total & 7) != 0) wid = blockIdx.x;
  const unsigned split_id = wid / tiles;
  unsigned tile_id = wid - split_id * tiles;
  const int ky = (int)(tile_id % KS); tile_id /= KS;
  const int ty = (int)(tile_id % n_ci), tz = (int)(tile_id / n_ci);
  const int co = tz * 32 + m, ci = ty * 128 + 4 * m;
  const bool cov = co < Cout, civ = ci < Cin;
  const unsigned cio = civ ? ci : 0, coo = cov ? co : 0;
  const unsigned nrows = (unsigned)B * OH;
  const unsigned r_begin = split_id * (unsigned)per_split;
  const unsigned r_end = r_begin + per_split < nrows ? r_begin + per_split : nrows;
  const int EH = IH << up, EW = IW << up;
  floatx16 acc[KS][4];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

  // operands of step (row r, pixel pairs ox0 .. ox0 + 2U): a[u] = dy of this lane's pixel, bq[u][kx] = 4 channels of x under tap
  // (ky, kx); mask bit u*4 + kx (bit 3 of each nibble: the dy value) says whether the operand exists.  Loads are unconditional
  // on clamped addresses.
  auto fetch = [&](unsigned r, int ox0, float (&a)[U], float4 (&bq)[U][KS], unsigned& mask) {
    const bool rv = r < r_end;
    const unsigned rc = rv ? r : r_begin;
    const unsigned b = rc / (unsigned)OH, oy = rc - b * (unsigned)OH;
    const int iy = (int)oy * stride + ky - pad;
    const bool rowok = rv && civ && iy >= 0 && iy < EH;
    const unsigned rowoff = ((b * (unsigned)IH + (rowok ? (unsigned)(iy >> up) : 0u)) * (unsigned)IW) * (unsigned)Cs + cio;
    const float* dyp = dy + (size_t)(rc * (unsigned)OW) * Cout + coo;
    mask = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ox = ox0 + 2 * u + kk;
      const bool v = rv && ox < OW;
      a[u] = dyp[(unsigned)(v ? ox : 0) * (unsigned)Cout];
      if (v && cov) mask |= 8u << (4 * u);
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int ix = ox * stride + kx - pad;
        const bool ok = v && rowok && ix >= 0 && ix < EW;
        bq[u][kx] = *reinterpret_cast<const float4*>(x + (size_t)(rowoff + (ok ? (unsigned)(ix >> up) * (unsigned)Cs : 0u)));
        if (ok) mask |= 1u << (4 * u + kx);
      }
    }
  };
  // the masks are applied when a step's operands are moved into place, so the MFMA block below is 48 back-to-back matrix
  // instructions with no VALU between them (a v_cndmask in front of every MFMA held the pipe at ~40 %)
  auto settle = [&](const float (&a)[U], const float4 (&bq)[U][KS], unsigned mask, float (&ao)[U], float4 (&bo)[U][KS]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ao[u] = ((mask >> (4 * u + 3)) & 1u) ? a[u] : 0.f;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const bool ok = (mask >> (4 * u + kx)) & 1u;
        bo[u][kx] = ok ? bq[u][kx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  unsigned r = r_begin + wave;
  int ox0 = 0;
  float a_c[U];
  float4 b_c[U][KS];
  {
    float a_n[U];
    float4 b_n[U][KS];
    unsigned m_n;
    fetch(r, ox0, a_n, b_n, m_n);
    settle(a_n, b_n, m_n, a_c, b_c);
  }
  while (r < r_end) {
    unsigned rn = r;
    int oxn = ox0 + 2 * U;
    if (oxn >= OW) { oxn = 0; rn = r + 4; }
    float a_n[U];
    float4 b_n[U][KS];
    unsigned m_n;
    fetch(rn, oxn, a_n, b_n, m_n);                 // (masked to nothing past the last row)
    // the loads of step i+1 must be ISSUED before the MFMAs of step i: a compiler-level memory fence (no instruction) keeps the
    // optimiser from sinking them to the end of the iteration, the sched_barrier keeps the machine scheduler from doing it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        acc[kx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[u], b_c[u][kx].x, acc[kx][0], 0, 0, 0);
        acc[kx][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[u], b_c[u][kx].y, acc[kx][1], 0, 0, 0);
        acc[kx][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[u], b_c[u][kx].z, acc[kx][2], 0, 0, 0);
        acc[kx][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[u], b_c[u][kx].w, acc[kx][3], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    settle(a_n, b_n, m_n, a_c, b_c);
    r = rn;
    ox0 = oxn;
  }

  // cross-wave reduction (fixed order: wave 0, 1, 2, 3)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float* d = &red[((t * 4 + j) * 16 + q) * 64 + lane];
            *d = (w == 0) ? acc[t][j][q] : *d + acc[t][j][q];
          }
    }
    __syncthreads();
  }
  // partial[split][co][ci][tap]: accumulator q of lane l in tile (kx, j) is D[row (q&3) + 8*(q>>2) + 4*(l>>5)][channel 4*(l&31) + j]
  float* dst = partial + (size_t)split_id * Cout * Cin * TAPS;
  for (int e = threadIdx.x; e < KS * 4 * 1024; e += 256) {
    const int l = e & 63, q = (e >> 6) & 15, j = (e >> 10) & 3, kx = e >> 12;
    const int rco = tz * 32 + (q & 3) + 8 * (q >> 2) + 4 * (l >> 5);
    const int rci = ty * 128 + 4 * (l & 31) + j;
    if (rco < Cout && rci < Cin) dst[((size_t)rco * Cin + rci) * TAPS + ky * KS + kx] = red[e];
  }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, size_t n, int splits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < splits; ++k) s += (double)partial[(size_t)k * n + i];
    dw[i] = (float)s;
  }
}

inline bool wgrad_wide(int Cin) { return Cin % 4 == 0 && getenv("CSD_WGRAD_WIDE"); }   // A/B switch: the 32x32x9 schedule measures faster

int wgrad_splits(int B, int OH, int OW, int Cin, int Cout, int ksize, int* per_split) {
  // K-splits are runs of output rows; ~1024 workgroups per launch, at least 4 rows (one per wave) and ~64 pixels each
  const long long nrows = (long long)B * OH;
  const int tiles = wgrad_wide(Cin) ? cdiv(Cout, 32) * cdiv(Cin, 128) * ksize : cdiv(Cout, 32) * cdiv(Cin, 32);
  long long S = std::max(1, 1024 / tiles);
  const long long min_rows = std::max<long long>(4, cdiv(64, OW));
  const long long max_s = std::max<long long>(1, nrows / min_rows);
  if (S > max_s) S = max_s;
  long long per = (nrows + S - 1) / S;
  S = (nrows + per - 1) / per;
  // prefer a split count that makes the grid a multiple of 8 (the XCD-aware order needs it): shrink `per` a little if that works
  for (long long p2 = per; p2 >= std::max<long long>(min_rows, per - 8); --p2) {
    const long long s2 = (nrows + p2 - 1) / p2;
    if ((s2 * tiles) % 8 == 0) { per = p2; S = s2; break; }
  }
  *per_split = (int)per;
  return (int)S;
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm (+activation) backward on NCHW: a group of one sample is one contiguous run of cpg*HW floats.
//   u = xhat*gamma + beta, y = act(u);  du = dy*act'(u);  dgamma_c = sum du*xhat, dbeta_c = sum du (per sample here,
//   summed over the batch by csd_sum_rows);  dx = rstd*(du*gamma - mean(du*gamma) - xhat*mean(du*gamma*xhat))
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ dy,
                                                     float* __restrict__ dx, float* __restrict__ dgamma_s,
                                                     float* __restrict__ dbeta_s, int C, int HW, int G, float eps,
                                                     int act) {
  __shared__ double sh[4];
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const size_t base = ((size_t)b * C + (size_t)g * cpg) * HW;
  const size_t n = (size_t)cpg * HW;
  double s = 0.0, q = 0.0;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    const double v = x[base + i];
    s += v;
    q += v * v;
  }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  const double mean = s / (double)n;
  double var = q / (double)n - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)mean;
  double A = 0.0, Bq = 0.0;       // sum dxhat, sum dxhat*xhat over the group
  for (int c = 0; c < cpg; ++c) {
    const int ch = g * cpg + c;
    const float ga = gamma[ch], be = beta[ch];
    double s1 = 0.0, s2 = 0.0;
    const size_t cb = base + (size_t)c * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float xh = (x[cb + i] - mu) * rstd;
      const float du = dy[cb + i] * bw_dact(xh * ga + be, act);
      s1 += du;
      s2 += (double)du * xh;
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) {
      dbeta_s[(size_t)b * C + ch] = (float)s1;
      dgamma_s[(size_t)b * C + ch] = (float)s2;
    }
    A += (double)ga * s1;
    Bq += (double)ga * s2;
  }
  const float mA = (float)(A / (double)n), mB = (float)(Bq / (double)n);
  for (int c = 0; c < cpg; ++c) {
    const int ch = g * cpg + c;
    const float ga = gamma[ch], be = beta[ch];
    const size_t cb = base + (size_t)c * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float xh = (x[cb + i] - mu) * rstd;
      const float du = dy[cb + i] * bw_dact(xh * ga + be, act);
      dx[cb + i] = rstd * (du * ga - mA - xh * mB);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// strided batched GEMM on the fp32 matrix cores:  C[z][m][n] = alpha * sum_k A[z][m*sam + k*sak] * B[z][k*sbk + n*sbn]
// 64x64 tile per workgroup, four waves x one 32x32 quadrant on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate),
// K step 16 through LDS (one ds_read_b32 per operand per MFMA; the FMA form of this kernel spent its time on 8 LDS reads per 16
// FMAs: 3.3 ms of a 42 ms training step, now ~0.5).  Any strides: the loader picks the unit-stride index as the fast one.
// Used by the attention backward (L <= 400) and the Linear backward ([B, <= 2048]).
// ---------------------------------------------------------------------------------------------------------------
struct GemmDesc {
  int M, N, K;
  long long sam, sak, sbk, sbn, scm, scn, za, zb, zc;
  float alpha;
};

typedef float bg_f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void bgemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                    float* __restrict__ Cm, GemmDesc d) {
  __shared__ float As[16][65], Bs[16][65];
  const int z = blockIdx.z;
  A += (size_t)z * d.za; Bm += (size_t)z * d.zb; Cm += (size_t)z * d.zc;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = (w >> 1) * 32, wn = (w & 1) * 32;
  const int lk = lane >> 5, li = lane & 31;
  bg_f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // loader mapping: pick the thread->element order that makes the unit-stride index the fast one
  const bool a_mfast = d.sam == 1 || (d.sak != 1 && d.sam < d.sak);
  const bool b_nfast = d.sbn == 1 || (d.sbk != 1 && d.sbn < d.sbk);
  for (int k0 = 0; k0 < d.K; k0 += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                 // all eight loads of the step in flight before the first LDS write
      const int e = threadIdx.x + r * 256;
      const int mm = a_mfast ? (e & 63) : (e >> 4), kk = a_mfast ? (e >> 6) : (e & 15);
      const int gm = m0 + mm, gk = k0 + kk;
      av[r] = (gm < d.M && gk < d.K) ? A[(long long)gm * d.sam + (long long)gk * d.sak] : 0.f;
      const int nn = b_nfast ? (e & 63) : (e >> 4), kb = b_nfast ? (e >> 6) : (e & 15);
      const int gn = n0 + nn, gkb = k0 + kb;
      bv[r] = (gn < d.N && gkb < d.K) ? Bm[(long long)gkb * d.sbk + (long long)gn * d.sbn] : 0.f;
    }
    __syncthreads();                              // the previous step's MFMA reads are done
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = threadIdx.x + r * 256;
      As[a_mfast ? (e >> 6) : (e & 15)][a_mfast ? (e & 63) : (e >> 4)] = av[r];
      Bs[b_nfast ? (e >> 6) : (e & 15)][b_nfast ? (e & 63) : (e >> 4)] = bv[r];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * j + lk][wm + li], Bs[2 * j + lk][wn + li], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {                  // D[(i/4)*8 + (lane/32)*4 + i%4][lane%32]
    const int gm = m0 + wm + (i >> 2) * 8 + lk * 4 + (i & 3), gn = n0 + wn + li;
    if (gm < d.M && gn < d.N) Cm[(long long)gm * d.scm + (long long)gn * d.scn] = d.alpha * acc[i];
  }
}

int bgemm_launch(const float* A, const float* Bm, float* Cm, const GemmDesc& d, int batch, hipStream_t s) {
  hipLaunchKernelGGL(bgemm_kernel, dim3(cdiv(d.N, 64), cdiv(d.M, 64), batch), dim3(256), 0, s, A, Bm, Cm, d);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// softmax over the rows of [R, L] in place (one wave per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, long long R, int L) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  float* row = S + r * L;
  float mx = -INFINITY;
  for (int j = lane; j < L; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float inv = 1.0f / sum;
  for (int j = lane; j < L; j += 64) row[j] *= inv;
}

// dS = P * (dP - sum_j P*dP), written over dP
__global__ __launch_bounds__(256) void dsoftmax_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, long long R,
                                                            int L) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + r * L;
  float* g = dP + r * L;
  float dot = 0.f;
  for (int j = lane; j < L; j += 64) dot += p[j] * g[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
  for (int j = lane; j < L; j += 64) g[j] = p[j] * (g[j] - dot);
}

// out[r] = sum over `inner` contiguous floats of row r (fp64 accumulation); one wave per row
__global__ __launch_bounds__(256) void sum_inner_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows,
                                                        long long inner) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + r * inner;
  double s = 0.0;
  for (long long i = lane; i < inner; i += 64) s += (double)p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[r] = (float)s;
}

// out[c] = sum_r x[r][c]  (fp64 accumulation, fixed order): 64 columns x 4 row lanes per workgroup, rows r = lane, lane+4, ...
// (blockIdx.y = 1: the second (x1, out1) pair of the same shape - the dgamma / dbeta rows of one GroupNorm backward in one launch)
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C,
                                                       const float* __restrict__ x1, float* __restrict__ out1) {
  if (blockIdx.y) { x = x1; out = out1; }
  __shared__ double sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double s = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int r = ty; r < R; r += 4) s += (double)x[(size_t)r * C + c];
  }
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) out[c] = (float)(sh[0][tx] + sh[1][tx] + sh[2][tx] + sh[3][tx]);
}

// dy == null: out = act(x);  else out = dy * act'(x)
__global__ void act_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int act,
                           size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = dy ? dy[i] * bw_dact(x[i], act) : bw_act(x[i], act);
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = a[i] * b[i];
}

// Philox4x32-10 (same generator as csd_randn): 4 uniforms per counter
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// nn.Dropout(p) in training mode: mask = (u >= p) / (1 - p), out = x * mask
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ mask, float p,
                               uint64_t seed, uint64_t stream_id, size_t n) {
  const float keep = 1.0f / (1.0f - p);
  const size_t n4 = (n + 3) / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t e = i * 4 + j;
      if (e < n) {
        const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
        const float mk = u >= p ? keep : 0.f;
        mask[e] = mk;
        out[e] = x[e] * mk;
      }
    }
  }
}

inline unsigned ew_grid(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 65536); }

}  // namespace

// ===============================================================================================================
// C ABI
// ===============================================================================================================
extern "C" size_t csd_conv_wgrad_scratch_bytes(int B, int Cin, int Cout, int H, int W, int ksize, int stride, int up2) {
  const int OH = (H << (up2 ? 1 : 0)) / stride, OW = (W << (up2 ? 1 : 0)) / stride;
  int per;
  const int S = wgrad_splits(B, OH, OW, Cin, Cout, ksize, &per);
  size_t ext = 0, S2 = 0;
  if (ksize == 3 && (stride == 2 || up2)) {      // resampling convs in the split-bf16 modes: operands rebuilt on the fine grid (wgrad_impl)
    const int FH = H << (up2 ? 1 : 0), FW = W << (up2 ? 1 : 0);
    int per2;
    S2 = (size_t)wgrad_splits(B, FH, FW, Cin, Cout, 3, &per2);
    ext = al64((size_t)B * FH * FW * (up2 ? Cin : Cout));
  }
  return (al64((size_t)B * H * W * Cin) + al64((size_t)B * OH * OW * Cout) + ext +
          al64(std::max((size_t)S, S2) * Cout * Cin * ksize * ksize)) * sizeof(float) + 1024;
}

// layout bit 0: x is NHWC [B,H,W,Cin]; bit 1: dy is NHWC [B,OH,OW,Cout]; bit 2: split-bf16 arithmetic allowed (else exact fp32)
static int wgrad_impl(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W, int ksize, int stride,
                      int pad_mode, int up2, int layout, void* scratch, void* stream) {
  CSD_REQUIRE(x && dy && dw && scratch, "conv2d_wgrad: null argument");
  CSD_REQUIRE(ksize == 1 || ksize == 3, "conv2d_wgrad: ksize must be 1 or 3");
  CSD_REQUIRE(stride == 1 || stride == 2, "conv2d_wgrad: stride must be 1 or 2");
  CSD_REQUIRE(!(up2 && stride != 1), "conv2d_wgrad: up2 requires stride 1");
  int pad;
  if (stride == 2) {
    CSD_REQUIRE(pad_mode == 1 && ksize == 3 && H % 2 == 0 && W % 2 == 0, "conv2d_wgrad: stride 2 is the reference Downsample");
    pad = 0;
  } else {
    CSD_REQUIRE(pad_mode == 0, "conv2d_wgrad: stride 1 uses symmetric padding");
    pad = ksize / 2;
  }
  hipStream_t s = (hipStream_t)stream;
  const int up = up2 ? 1 : 0;
  const int OH = (H << up) / stride, OW = (W << up) / stride;
  CSD_REQUIRE((double)B * H * W * Cin < 4.0e9 && (double)B * OH * OW * Cout < 4.0e9, "conv2d_wgrad: tensor exceeds 32-bit indexing");
  int per;
  const int S = wgrad_splits(B, OH, OW, Cin, Cout, ksize, &per);
  float* f = static_cast<float*>(scratch);
  float* xh = f; f += al64((size_t)B * H * W * Cin);
  float* dyh = f; f += al64((size_t)B * OH * OW * Cout);
  float* partial = f;
  int rc;
  if (layout & 1) xh = const_cast<float*>(x);
  else if ((rc = nchw_to_nhwc_launch(x, xh, B, Cin, H * W, Cin, Cin, s))) return rc;
  if (layout & 2) dyh = const_cast<float*>(dy);
  else if ((rc = nchw_to_nhwc_launch(dy, dyh, B, Cout, OH * OW, Cout, Cout, s))) return rc;
  if ((layout & 4) && ksize == 3 && (stride == 2 || up) && Cin % 4 == 0 && Cout % 4 == 0 && !wgrad_wide(Cin) && !getenv("CSD_WGRAD_FP32") &&
      !CSD_TUNE_ENV("CSD_WGRAD_RESAMPLE_FP32")) {
    // resampling convs on the bf16 kernel too (it knows stride 1 only): rebuild ONE operand on the fine grid.
    //   Upsample (nearest x2, then 3x3): x' = nearest_up2(x), the plain stride-1 gradient of (x', dy).
    //   Downsample (pad (0,1,0,1), stride 2): y[p] = sum_k W[k] x[2p + k]  ->  dW[k] = sum_q dy'[q] x[q + k - 1] with dy'[2p + 1] = dy[p],
    //   zeros elsewhere: the stride-1, pad-1 gradient of (x, dy') (x[H] = 0 is that gradient's own padding).
    // 4x the matrix work of a direct kernel (3/4 zeros / repeats), still half the time of the fp32 one (322 us -> ~150 us per layer
    // of the 64 x 64 training net).
    const int FH = H << up, FW = W << up;
    int per2;
    const int S2 = wgrad_splits(B, FH, FW, Cin, Cout, 3, &per2);
    float* ext = partial;
    float* partial2 = ext + al64((size_t)B * FH * FW * (up ? Cin : Cout));
    if (up) { if ((rc = nearest_up2_nhwc_launch(xh, ext, B, H, W, Cin, s))) return rc; }
    else if ((rc = csd_zero_insert_odd_nhwc(dyh, ext, B, OH, OW, Cout, s))) return rc;
    if ((rc = wgrad_bf16_launch(up ? ext : xh, up ? dyh : ext, partial2, B, FH, FW, Cin, Cout, 3, S2, per2, s))) return rc;
    const size_t n = (size_t)Cout * Cin * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, s, partial2, dw, n, S2);
    CSD_LAUNCH_CHECK();
    return CSD_OK;
  }
  if ((layout & 4) && stride == 1 && !up && !wgrad_wide(Cin) && !getenv("CSD_WGRAD_FP32")) {
    // split-bf16 operands on the bf16 matrix cores (wgrad_bf16.hip): the fp16 precision modes of the training step
    if ((rc = wgrad_bf16_launch(xh, dyh, partial, B, H, W, Cin, Cout, ksize, S, per, s))) return rc;
    const size_t n = (size_t)Cout * Cin * ksize * ksize;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, s, partial, dw, n, S);
    CSD_LAUNCH_CHECK();
    return CSD_OK;
  }
  if (wgrad_wide(Cin)) {
    const int n_ci = cdiv(Cin, 128), n_co = cdiv(Cout, 32);
    const dim3 grid((unsigned)S * n_ci * n_co * ksize);
    if (ksize == 3)
      hipLaunchKernelGGL(conv_wgrad4_kernel<3>, grid, dim3(256), 0, s, xh, dyh, partial, B, H, W, Cin, Cin, OH, OW, Cout, stride,
                         pad, up, per, n_ci, n_co);
    else
      hipLaunchKernelGGL(conv_wgrad4_kernel<1>, grid, dim3(256), 0, s, xh, dyh, partial, B, H, W, Cin, Cin, OH, OW, Cout, stride,
                         pad, up, per, n_ci, n_co);
    CSD_LAUNCH_CHECK();
    const size_t n = (size_t)Cout * Cin * ksize * ksize;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, s, partial, dw, n, S);
    CSD_LAUNCH_CHECK();
    return CSD_OK;
  }
  const int n_ci = cdiv(Cin, 32), n_co = cdiv(Cout, 32);
  const dim3 grid((unsigned)S * n_ci * n_co);
  if (ksize == 3)
    hipLaunchKernelGGL(conv_wgrad_kernel<9>, grid, dim3(256), 0, s, xh, dyh, partial, B, H, W, Cin, Cin, OH, OW, Cout, stride,
                       pad, up, per, n_ci, n_co);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, s, xh, dyh, partial, B, H, W, Cin, Cin, OH, OW, Cout, stride,
                       pad, up, per, n_ci, n_co);
  CSD_LAUNCH_CHECK();
  const size_t n = (size_t)Cout * Cin * ksize * ksize;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, s, partial, dw, n, S);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                int ksize, int stride, int pad_mode, int up2, void* scratch, void* stream) {
  return wgrad_impl(x, dy, dw, B, Cin, Cout, H, W, ksize, stride, pad_mode, up2, 0, scratch, stream);
}

extern "C" int csd_conv2d_wgrad_ex(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                   int ksize, int stride, int pad_mode, int up2, int layout, void* scratch, void* stream) {
  return wgrad_impl(x, dy, dw, B, Cin, Cout, H, W, ksize, stride, pad_mode, up2, layout, scratch, stream);
}

extern "C" int csd_groupnorm_act_backward(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                                          float* dgamma_rows, float* dbeta_rows, int B, int C, int H, int W, int groups,
                                          float eps, int act, void* stream) {
  CSD_REQUIRE(x && gamma && beta && dy && dx && dgamma_rows && dbeta_rows, "groupnorm_act_backward: null argument");
  CSD_REQUIRE(groups > 0 && C % groups == 0, "groupnorm_act_backward: %d channels not divisible into %d groups", C, groups);
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, dy, dx, dgamma_rows,
                     dbeta_rows, C, H * W, groups, eps, act);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_bgemm(const float* A, const float* Bm, float* Cm, int M, int N, int K, int64_t sam, int64_t sak,
                         int64_t sbk, int64_t sbn, int64_t scm, int64_t scn, int batch, int64_t za, int64_t zb, int64_t zc,
                         float alpha, void* stream) {
  CSD_REQUIRE(A && Bm && Cm && M > 0 && N > 0 && K > 0 && batch > 0, "bgemm: bad arguments");
  GemmDesc d{M, N, K, sam, sak, sbk, sbn, scm, scn, za, zb, zc, alpha};
  return bgemm_launch(A, Bm, Cm, d, batch, (hipStream_t)stream);
}

extern "C" size_t csd_attention_backward_scratch_bytes(int B, int C, int H, int W) {
  const size_t L = (size_t)H * W;
  return 2 * al64((size_t)B * L * L) * sizeof(float) + 1024;
}

// q, k, v, dout, dq, dk, dv: [B, C, H, W]; the forward is csd_attention (models/layers.py:584-588)
extern "C" int csd_attention_backward(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                                      float* dv, int B, int C, int H, int W, void* scratch, void* stream) {
  CSD_REQUIRE(q && k && v && dout && dq && dk && dv && scratch, "attention_backward: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int L = H * W;
  const long long zq = (long long)C * L, zs = (long long)L * L;
  const float scale = 1.0f / sqrtf((float)C);
  float* P = static_cast<float*>(scratch);
  float* dP = P + al64((size_t)B * L * L);
  int rc;
  // S[i][j] = scale * sum_c q[c][i] k[c][j]
  if ((rc = bgemm_launch(q, k, P, GemmDesc{L, L, C, 1, L, L, 1, L, 1, zq, zq, zs, scale}, B, s))) return rc;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64((long long)B * L, 4)), dim3(256), 0, s, P, (long long)B * L, L);
  CSD_LAUNCH_CHECK();
  // dV[c][j] = sum_i dO[c][i] P[i][j]
  if ((rc = bgemm_launch(dout, P, dv, GemmDesc{C, L, L, L, 1, L, 1, L, 1, zq, zs, zq, 1.f}, B, s))) return rc;
  // dP[i][j] = sum_c dO[c][i] v[c][j]
  if ((rc = bgemm_launch(dout, v, dP, GemmDesc{L, L, C, 1, L, L, 1, L, 1, zq, zq, zs, 1.f}, B, s))) return rc;
  hipLaunchKernelGGL(dsoftmax_rows_kernel, dim3((unsigned)cdiv64((long long)B * L, 4)), dim3(256), 0, s, P, dP, (long long)B * L,
                     L);
  CSD_LAUNCH_CHECK();
  // dq[c][i] = scale * sum_j k[c][j] dS[i][j]
  if ((rc = bgemm_launch(k, dP, dq, GemmDesc{C, L, L, L, 1, 1, L, L, 1, zq, zs, zq, scale}, B, s))) return rc;
  // dk[c][j] = scale * sum_i q[c][i] dS[i][j]
  return bgemm_launch(q, dP, dk, GemmDesc{C, L, L, L, 1, L, 1, L, 1, zq, zs, zq, scale}, B, s);
}

// the same on the packed NHWC tensors of the training graph: qkv, dqkv [B, L, 3C] (q | k | v per pixel), dout [B, L, C]
extern "C" int csd_attention_backward_nhwc(const float* qkv, const float* dout, float* dqkv, int B, int L, int C, void* scratch,
                                           void* stream) {
  CSD_REQUIRE(qkv && dout && dqkv && scratch, "attention_backward_nhwc: null argument");
  hipStream_t s = (hipStream_t)stream;
  const long long C3 = 3LL * C, zq = (long long)L * C3, zo = (long long)L * C, zs = (long long)L * L;
  const float scale = 1.0f / sqrtf((float)C);
  const float *q = qkv, *k = qkv + C, *v = qkv + 2 * C;
  float *dq = dqkv, *dk = dqkv + C, *dv = dqkv + 2 * C;
  float* P = static_cast<float*>(scratch);
  float* dP = P + al64((size_t)B * L * L);
  int rc;
  if ((rc = bgemm_launch(q, k, P, GemmDesc{L, L, C, C3, 1, 1, C3, L, 1, zq, zq, zs, scale}, B, s))) return rc;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64((long long)B * L, 4)), dim3(256), 0, s, P, (long long)B * L, L);
  CSD_LAUNCH_CHECK();
  if ((rc = bgemm_launch(dout, P, dv, GemmDesc{C, L, L, 1, C, L, 1, 1, C3, zo, zs, zq, 1.f}, B, s))) return rc;
  if ((rc = bgemm_launch(dout, v, dP, GemmDesc{L, L, C, C, 1, 1, C3, L, 1, zo, zq, zs, 1.f}, B, s))) return rc;
  hipLaunchKernelGGL(dsoftmax_rows_kernel, dim3((unsigned)cdiv64((long long)B * L, 4)), dim3(256), 0, s, P, dP, (long long)B * L,
                     L);
  CSD_LAUNCH_CHECK();
  if ((rc = bgemm_launch(k, dP, dq, GemmDesc{C, L, L, 1, C3, 1, L, 1, C3, zq, zs, zq, scale}, B, s))) return rc;
  return bgemm_launch(q, dP, dk, GemmDesc{C, L, L, 1, C3, L, 1, 1, C3, zq, zs, zq, scale}, B, s);
}

extern "C" int csd_sum_inner(const float* x, float* out, int64_t rows, int64_t inner, void* stream) {
  CSD_REQUIRE(x && out && rows > 0 && inner > 0, "sum_inner: bad arguments");
  hipLaunchKernelGGL(sum_inner_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long long)rows, (long long)inner);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_sum_rows(const float* x, float* out, int R, int C, void* stream) {
  CSD_REQUIRE(x && out && R > 0 && C > 0, "sum_rows: bad arguments");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, out, R, C, nullptr, nullptr);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int csd::sum_rows2(const float* x0, float* out0, const float* x1, float* out1, int R, int C, void* stream) {
  CSD_REQUIRE(x0 && out0 && x1 && out1 && R > 0 && C > 0, "sum_rows2: bad arguments");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv(C, 64), 2), dim3(256), 0, (hipStream_t)stream, x0, out0, R, C, x1, out1);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_act(const float* x, const float* dy, float* out, int act, int64_t n, void* stream) {
  CSD_REQUIRE(x && out && n > 0, "act: bad arguments");
  hipLaunchKernelGGL(act_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, dy, out, act, (size_t)n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
  CSD_REQUIRE(a && b && out && n > 0, "mul: bad arguments");
  hipLaunchKernelGGL(mul_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

extern "C" int csd_dropout(const float* x, float* out, float* mask, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                           void* stream) {
  CSD_REQUIRE(x && out && mask && n > 0 && p >= 0.f && p < 1.f, "dropout: bad arguments");
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid((size_t)(n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, mask, p, seed,
                     stream_id, (size_t)n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}
