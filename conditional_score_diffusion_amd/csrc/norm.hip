// norm.hip - GroupNorm statistics on NHWC fp32 (HBM-bound streaming reduction).
//
// Replaces nn.GroupNorm(num_groups, C, eps=1e-6) as used at models/layers.py:571,638,646 and
// models/ddpm.py:145.  The normalisation itself is NOT a separate pass: gn_finalize emits a
// per-(sample, channel) scale/shift pair which the consuming convolution applies (together with
// the SiLU) while staging its source patch (conv_f32.hip) - the normalised tensor never
// touches HBM.  gn_apply exists for the stand-alone csd_groupnorm_act() entry point only.
//
//   pass 1  gn_stats    : grid (nchunk, B); each workgroup streams a slab of pixels x ALL
//                         channels (float4, fully coalesced rows of the NHWC tensor) and keeps
//                         per-channel sum / sum-of-squares in fp64 registers (fp64 accumulate is
//                         free under an HBM-bound stream and makes E[x^2]-E[x]^2 safe), then
//                         folds them to per-group partials - deterministic, no atomics.
//   pass 2  gn_finalize : grid B; folds the nchunk partials, writes scale = rstd*gamma and
//                         shift = beta - mean*rstd*gamma.
// The source may be a virtual channel-concat of two tensors (skip connections, models/ddpm.py:197)
// - groups may straddle the seam, which is why statistics are kept per channel first.
#include "common.h"

namespace csd {

#define GN_THREADS 256

__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(
    const float* __restrict__ src0, const float* __restrict__ src1, double* __restrict__ partial,
    int HW, int C0, int C1, int G, int nchunk, int per_channel) {
  extern __shared__ __attribute__((aligned(16))) double sred[];   // [2][C]
  const int C = C0 + C1;
  const int C4 = C >> 2;
  const int rows = GN_THREADS / C4;            // pixel rows processed per sweep (>=1: C <= 1024)
  const int tid = threadIdx.x;
  const int row = tid / C4;
  const int c4 = tid - row * C4;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = chunk * per;
  const int p1 = min(HW, p0 + per);
  const bool active = row < rows;

  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (active) {
    const int c = c4 * 4;
    const float* src;
    int Cs, coff;
    if (c < C0) { src = src0; Cs = C0; coff = c; }
    else { src = src1; Cs = C1; coff = c - C0; }
    const float* base = src + (size_t)b * HW * Cs + coff;
    // eight rows in flight per thread (one load per dependent fp64 chain left the kernel latency-bound: 2.6 TB/s on the 160^2 x 96
    // Upsample output); the additions keep their order - same bits
    int p = p0 + row;
    for (; p + 7 * rows < p1; p += 8 * rows) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)(p + u * rows) * Cs);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s[0] += v[u].x; q[0] += (double)v[u].x * v[u].x;
        s[1] += v[u].y; q[1] += (double)v[u].y * v[u].y;
        s[2] += v[u].z; q[2] += (double)v[u].z * v[u].z;
        s[3] += v[u].w; q[3] += (double)v[u].w * v[u].w;
      }
    }
    for (; p < p1; p += rows) {
      const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * Cs);
      s[0] += v.x; q[0] += (double)v.x * v.x;
      s[1] += v.y; q[1] += (double)v.y * v.y;
      s[2] += v.z; q[2] += (double)v.z * v.z;
      s[3] += v.w; q[3] += (double)v.w * v.w;
    }
  }
  // fold the `rows` row-partials per channel in a fixed order (deterministic)
  double* ssum = sred;
  double* ssq = sred + C;
  for (int r = 0; r < rows; ++r) {
    if (active && row == r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c4 * 4 + j;
        if (r == 0) { ssum[c] = s[j]; ssq[c] = q[j]; }
        else { ssum[c] += s[j]; ssq[c] += q[j]; }
      }
    }
    __syncthreads();
  }
  if (per_channel) {      // the layout of the conv epilogues' tile partials ([B * nchunk][C][2]): gn_finalize_tiles folds them with another source's
    for (int c = tid; c < C; c += GN_THREADS) {
      double* dst = partial + (((size_t)b * nchunk + chunk) * C + c) * 2;
      dst[0] = ssum[c];
      dst[1] = ssq[c];
    }
    return;
  }
  const int cpg = C / G;
  for (int g = tid; g < G; g += GN_THREADS) {
    double a = 0, bsum = 0;
    for (int j = 0; j < cpg; ++j) { a += ssum[g * cpg + j]; bsum += ssq[g * cpg + j]; }
    double* dst = partial + (((size_t)b * nchunk + chunk) * G + g) * 2;
    dst[0] = a;
    dst[1] = bsum;
  }
}

__global__ void gn_finalize_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float* __restrict__ nscale,
                                   float* __restrict__ nshift, int HW, int C, int G, int nchunk, float* __restrict__ rs,
                                   float* __restrict__ ms) {
  const int b = blockIdx.x;
  const int cpg = C / G;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cpg;
    double s = 0, q = 0;
    for (int k = 0; k < nchunk; ++k) {
      const double* p = partial + (((size_t)b * nchunk + k) * G + g) * 2;
      s += p[0];
      q += p[1];
    }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = rstd * (gamma ? gamma[c] : 1.f);          // gamma == null: the plain statistics (rstd, -mean*rstd)
    nscale[(size_t)b * C + c] = sc;
    nshift[(size_t)b * C + c] = (beta ? beta[c] : 0.f) - (float)mean * sc;
    if (rs) {                                                  // the plain statistics next to the affine ones (training forward: one launch, not two)
      rs[(size_t)b * C + c] = rstd;
      ms[(size_t)b * C + c] = 0.f - (float)mean * rstd;
    }
  }
}

// finalize from conv-epilogue partials.  One workgroup per (block of GB groups, sample): thread (ti, c) strides over
// the tiles ti, ti+TL, ... of channel c - the cb = cpg*GB channels of one tile row are contiguous, so the reads
// coalesce - then fixed-order sums over the TL tile lanes and over the channels of each group (deterministic).
#define GNF_THREADS 256
__global__ void gn_finalize_tiles_kernel(const double* __restrict__ p0, int tpi0, int C0, const double* __restrict__ p1,
                                         int tpi1, int C1, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float eps, float* __restrict__ nscale, float* __restrict__ nshift, int HW, int G,
                                         int GB) {
  __shared__ double red[2][GNF_THREADS];
  __shared__ double chs[2][64];
  const int b = blockIdx.y, t = threadIdx.x;
  const int C = C0 + C1;
  const int cpg = C / G;
  const int cb = cpg * GB;                 // channels per workgroup (<= 64)
  const int TL = GNF_THREADS / cb;         // tile lanes
  const int cl = t % cb, ti = t / cb;
  const int c = blockIdx.x * cb + cl;
  double s = 0, q = 0;
  if (ti < TL && c < C) {
    const bool s1 = c >= C0;
    const double* p = s1 ? p1 : p0;
    const int Cs = s1 ? C1 : C0, cs = s1 ? c - C0 : c, tpi = s1 ? tpi1 : tpi0;
    // (sum, sumsq) pairs as one 16-byte load; unrolled so that 8 loads are in flight - the adds keep their order (the kernel was
    // latency-bound: ~17 us per launch whatever the size, one dependent load per add)
    const double2* e = reinterpret_cast<const double2*>(p + ((size_t)b * tpi * Cs + cs) * 2);
    int i = ti;
    for (; i + 7 * TL < tpi; i += 8 * TL) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = e[(size_t)(i + u * TL) * Cs];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += v[u].x; q += v[u].y; }
    }
    for (; i < tpi; i += TL) {
      const double2 v = e[(size_t)i * Cs];
      s += v.x;
      q += v.y;
    }
  }
  red[0][t] = s;
  red[1][t] = q;
  __syncthreads();
  if (t < cb) {
    double cs_ = 0, cq_ = 0;
    for (int j = 0; j < TL; ++j) { cs_ += red[0][j * cb + t]; cq_ += red[1][j * cb + t]; }
    chs[0][t] = cs_;
    chs[1][t] = cq_;
  }
  __syncthreads();
  if (t < cb && c < C) {
    const int g0 = (t / cpg) * cpg;
    double gs = 0, gq = 0;
    for (int j = 0; j < cpg; ++j) { gs += chs[0][g0 + j]; gq += chs[1][g0 + j]; }
    const double n = (double)HW * cpg;
    const double mean = gs / n;
    double var = gq / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = rstd * (gamma ? gamma[c] : 1.f);          // gamma == null: the plain statistics (rstd, -mean*rstd)
    nscale[(size_t)b * C + c] = sc;
    nshift[(size_t)b * C + c] = (beta ? beta[c] : 0.f) - (float)mean * sc;
  }
}

__device__ __forceinline__ float gn_act(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: return v / (1.0f + expf(-v));
    case CSD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case CSD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// (hi != null: the fp16 hi | lo planes of y as well - what gn_apply16 would make of y in a second pass; the training forward's conv reads them)
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gn_store_planes(half4_t* __restrict__ hi, half4_t* __restrict__ lo, size_t i, const float4& o) {
  half4_t h, l;
  h[0] = (_Float16)o.x; h[1] = (_Float16)o.y; h[2] = (_Float16)o.z; h[3] = (_Float16)o.w;
  l[0] = (_Float16)(o.x - (float)h[0]); l[1] = (_Float16)(o.y - (float)h[1]);
  l[2] = (_Float16)(o.z - (float)h[2]); l[3] = (_Float16)(o.w - (float)h[3]);
  hi[i] = h;
  if (lo) lo[i] = l;
}
__global__ void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ nscale,
                                const float* __restrict__ nshift, float* __restrict__ y, int HW, int C,
                                int act, size_t total4, half4_t* __restrict__ hi, half4_t* __restrict__ lo) {
  const int C4 = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / C4;
    const int c4 = (int)(i - pix * C4);
    const int b = (int)(pix / HW);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = *reinterpret_cast<const float4*>(nscale + (size_t)b * C + c4 * 4);
    const float4 sh = *reinterpret_cast<const float4*>(nshift + (size_t)b * C + c4 * 4);
    float4 o;
    o.x = gn_act(v.x * sc.x + sh.x, act);
    o.y = gn_act(v.y * sc.y + sh.y, act);
    o.z = gn_act(v.z * sc.z + sh.z, act);
    o.w = gn_act(v.w * sc.w + sh.w, act);
    reinterpret_cast<float4*>(y)[i] = o;
    if (hi) gn_store_planes(hi, lo, i, o);
  }
}

// the training forward's dropout(act(GroupNorm(x))) in one pass: the mask (0 or 1 / (1 - p), nn.Dropout) of float4 i is Philox4x32-10 counter
// i of (seed, stream) - element for element what csd_dropout (backward.hip) draws, so the fused and the two-launch form agree bitwise
__global__ void gn_apply_dropout_kernel(const float* __restrict__ x, const float* __restrict__ nscale, const float* __restrict__ nshift,
                                        float* __restrict__ y, float* __restrict__ mask, int HW, int C, int act, size_t total4, float p,
                                        uint64_t seed, uint64_t stream_id, half4_t* __restrict__ hi, half4_t* __restrict__ lo) {
  const int C4 = C >> 2;
  const float keep = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / C4;
    const int c4 = (int)(i - pix * C4);
    const int b = (int)(pix / HW);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = *reinterpret_cast<const float4*>(nscale + (size_t)b * C + c4 * 4);
    const float4 sh = *reinterpret_cast<const float4*>(nshift + (size_t)b * C + c4 * 4);
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
      c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float4 m;
    m.x = (float)(c[0] >> 8) * (1.0f / 16777216.0f) >= p ? keep : 0.f;
    m.y = (float)(c[1] >> 8) * (1.0f / 16777216.0f) >= p ? keep : 0.f;
    m.z = (float)(c[2] >> 8) * (1.0f / 16777216.0f) >= p ? keep : 0.f;
    m.w = (float)(c[3] >> 8) * (1.0f / 16777216.0f) >= p ? keep : 0.f;
    float4 o;
    o.x = gn_act(v.x * sc.x + sh.x, act) * m.x;
    o.y = gn_act(v.y * sc.y + sh.y, act) * m.y;
    o.z = gn_act(v.z * sc.z + sh.z, act) * m.z;
    o.w = gn_act(v.w * sc.w + sh.w, act) * m.w;
    reinterpret_cast<float4*>(y)[i] = o;
    reinterpret_cast<float4*>(mask)[i] = m;
    if (hi) gn_store_planes(hi, lo, i, o);
  }
}

// GroupNorm affine + activation + fp16 split, ONCE per element (consumer: conv_f16_kernel<IN16>).
// HBM-bound stream: grid (chunks, B); a thread owns 8 fixed channels (its scale/shift live in registers
// for the whole block), sweeps pixel rows: two float4 in, one 16-byte store per plane out.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
#define GA_THREADS 256
__global__ __launch_bounds__(GA_THREADS) void gn_apply16_kernel(
    const float* __restrict__ src0, const float* __restrict__ src1, int C0, int C1,
    const float* __restrict__ nscale, const float* __restrict__ nshift, half8_t* __restrict__ hi,
    half8_t* __restrict__ lo, int HW, int act, int nchunk, int f8) {
  const int C = C0 + C1;
  const int C8 = C >> 3;
  const int rows = GA_THREADS / C8;           // pixel rows per sweep (>= 1: C <= 2048)
  const int tid = threadIdx.x;
  const int row = tid / C8;
  const int c = (tid - row * C8) * 8;
  if (row >= rows) return;
  const int b = blockIdx.y;
  const int per = (HW + nchunk - 1) / nchunk;
  const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  const float* src;
  int Cs;
  if (c < C0) { src = src0 + (size_t)b * HW * C0 + c; Cs = C0; }
  else { src = src1 + (size_t)b * HW * C1 + (c - C0); Cs = C1; }
  // (nscale == nullptr: identity affine - a plain fp32 -> fp16 hi|lo split for convolutions without a GroupNorm)
  float sc[8], sh[8];
  if (nscale) {
    const float4 sca = *reinterpret_cast<const float4*>(nscale + (size_t)b * C + c);
    const float4 scb = *reinterpret_cast<const float4*>(nscale + (size_t)b * C + c + 4);
    const float4 sha = *reinterpret_cast<const float4*>(nshift + (size_t)b * C + c);
    const float4 shb = *reinterpret_cast<const float4*>(nshift + (size_t)b * C + c + 4);
    sc[0] = sca.x; sc[1] = sca.y; sc[2] = sca.z; sc[3] = sca.w; sc[4] = scb.x; sc[5] = scb.y; sc[6] = scb.z; sc[7] = scb.w;
    sh[0] = sha.x; sh[1] = sha.y; sh[2] = sha.z; sh[3] = sha.w; sh[4] = shb.x; sh[5] = shb.y; sh[6] = shb.z; sh[7] = shb.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
  }
  half8_t* hdst = hi + ((size_t)b * HW * C + c) / 8;
  half8_t* ldst = lo ? lo + ((size_t)b * HW * C + c) / 8 : nullptr;
  for (int p = p0 + row; p < p1; p += rows) {
    const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)p * Cs);
    const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)p * Cs + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    half8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j] * sc[j] + sh[j];
      switch (act) {
        case CSD_ACT_SWISH: t = t * __frcp_rn(1.0f + __expf(-t)); break;
        case CSD_ACT_RELU: t = t > 0.f ? t : 0.f; break;
        case CSD_ACT_LRELU: t = t > 0.f ? t : 0.2f * t; break;
        case CSD_ACT_ELU: t = t > 0.f ? t : expm1f(t); break;
        default: break;
      }
      h[j] = (_Float16)t;
      l[j] = (_Float16)(t - (float)h[j]);
      if (f8) {          // second plane = per channel the e4m3 byte pair (lo * 2^11, hi): operands of the fp8 correction MFMA
        const float hf = fminf(fmaxf((float)h[j], -448.f), 448.f), lf = fminf(fmaxf((t - (float)h[j]) * 2048.f, -448.f), 448.f);
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(lf, hf, 0, false);
        l[j] = __builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffff));
      }
    }
    hdst[(size_t)p * C8] = h;
    if (ldst) ldst[(size_t)p * C8] = l;
  }
}

// GroupNorm statistics + affine + activation + fp16 split in ONE launch, for maps small enough that a workgroup keeps its share of a
// sample in registers (the 10^2 / 5^2 levels and the narrower 20^2 tensors of SR3-160: three dependent launches of 5-10 us each -
// gn_stats, gn_finalize, gn_apply16 - were pure latency there; reference models/layers.py:638,646,659,667 h = act(GroupNorm(h))).
// One workgroup = (sample, GB consecutive groups = cb channels, cb % 8 == 0): thread (row, unit) owns the 8 channels of unit `unit` at
// pixels row, row + rows, ...: ONE sweep over HBM/L2, per-channel (sum, sum of squares) in fp64, rows folded in a fixed order through
// LDS, then mean / rstd per group, scale / shift per channel, and the arithmetic of gn_apply16_kernel on the registers.  Deterministic
// and per-sample: a sample's bits do not depend on the batch it runs in.
#define GNFU_THREADS 256
#define GNFU_MAXS 6                                  // pixel sweeps a thread holds (48 floats)
__global__ __launch_bounds__(GNFU_THREADS) void gn_fused16_kernel(
    const float* __restrict__ src0, const float* __restrict__ src1, int C0, int C1, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, half8_t* __restrict__ hi, half8_t* __restrict__ lo, int HW, int G, int GB, int act,
    int f8, float* __restrict__ nscale, float* __restrict__ nshift) {
  __shared__ double sred[2][GNFU_THREADS * 8];
  __shared__ double chs[2][GNFU_THREADS];
  __shared__ float ssc[GNFU_THREADS], ssh[GNFU_THREADS];
  const int C = C0 + C1, cpg = C / G, cb = cpg * GB, U = cb >> 3, rows = GNFU_THREADS / U;
  const int t = threadIdx.x, pr = t / U, u = t - pr * U;
  const int b = blockIdx.y, cbase = blockIdx.x * cb;
  const int c = cbase + u * 8;                       // first of this thread's 8 channels
  const bool active = pr < rows;
  const float* src;
  int Cs;
  if (c < C0) { src = src0 + (size_t)b * HW * C0 + c; Cs = C0; }
  else { src = src1 + (size_t)b * HW * C1 + (c - C0); Cs = C1; }
  float v[GNFU_MAXS][8];
  double s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0; q[j] = 0; }
  // every sweep's two loads are requested before the first is used: clamped addresses, no branch around a load (a guarded load sits in
  // its own basic block behind an s_waitcnt - six dependent L2 round trips per thread, ~5 us of a launch that moves 2 MB; round 6)
  float4 ld0[GNFU_MAXS], ld1[GNFU_MAXS];
#pragma unroll
  for (int k = 0; k < GNFU_MAXS; ++k) {
    const int p = min(pr + k * rows, HW - 1);
    ld0[k] = *reinterpret_cast<const float4*>(src + (size_t)p * Cs);
    ld1[k] = *reinterpret_cast<const float4*>(src + (size_t)p * Cs + 4);
  }
#pragma unroll
  for (int k = 0; k < GNFU_MAXS; ++k) {
    const bool valid = active && pr + k * rows < HW;
    v[k][0] = ld0[k].x; v[k][1] = ld0[k].y; v[k][2] = ld0[k].z; v[k][3] = ld0[k].w;
    v[k][4] = ld1[k].x; v[k][5] = ld1[k].y; v[k][6] = ld1[k].z; v[k][7] = ld1[k].w;
    if (valid) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[k][j]; q[j] += (double)v[k][j] * v[k][j]; }
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sred[0][pr * cb + u * 8 + j] = s[j]; sred[1][pr * cb + u * 8 + j] = q[j]; }
  }
  __syncthreads();
  if (t < cb) {                                      // rows folded per channel, in row order (eight LDS reads in flight, the adds in order)
    double a = 0, bq = 0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
      double x0[8], x1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { x0[i] = sred[0][(r + i) * cb + t]; x1[i] = sred[1][(r + i) * cb + t]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) { a += x0[i]; bq += x1[i]; }
    }
    for (; r < rows; ++r) { a += sred[0][r * cb + t]; bq += sred[1][r * cb + t]; }
    chs[0][t] = a;
    chs[1][t] = bq;
  }
  __syncthreads();
  if (t < cb) {
    const int g0 = (t / cpg) * cpg;
    double gs = 0, gq = 0;
    for (int j = 0; j < cpg; ++j) { gs += chs[0][g0 + j]; gq += chs[1][g0 + j]; }
    const double n = (double)HW * cpg;
    const double mean = gs / n;
    double var = gq / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = rstd * gamma[cbase + t];
    ssc[t] = sc;
    ssh[t] = beta[cbase + t] - (float)mean * sc;
    if (nscale) {                                    // statistics-only form: the per-(sample, channel) scale / shift for a consumer that applies them itself
      nscale[(size_t)b * C + cbase + t] = ssc[t];
      nshift[(size_t)b * C + cbase + t] = ssh[t];
    }
  }
  if (hi == nullptr) return;
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = ssc[u * 8 + j]; sh[j] = ssh[u * 8 + j]; }
  const int C8 = C >> 3;
  half8_t* hdst = hi + ((size_t)b * HW * C + c) / 8;
  half8_t* ldst = lo ? lo + ((size_t)b * HW * C + c) / 8 : nullptr;
#pragma unroll
  for (int k = 0; k < GNFU_MAXS; ++k) {
    const int p = pr + k * rows;
    if (p < HW) {
      half8_t h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {                  // (the arithmetic of gn_apply16_kernel, to the letter)
        float tv = v[k][j] * sc[j] + sh[j];
        switch (act) {
          case CSD_ACT_SWISH: tv = tv * __frcp_rn(1.0f + __expf(-tv)); break;
          case CSD_ACT_RELU: tv = tv > 0.f ? tv : 0.f; break;
          case CSD_ACT_LRELU: tv = tv > 0.f ? tv : 0.2f * tv; break;
          case CSD_ACT_ELU: tv = tv > 0.f ? tv : expm1f(tv); break;
          default: break;
        }
        h[j] = (_Float16)tv;
        l[j] = (_Float16)(tv - (float)h[j]);
        if (f8) {
          const float hf = fminf(fmaxf((float)h[j], -448.f), 448.f), lf = fminf(fmaxf((tv - (float)h[j]) * 2048.f, -448.f), 448.f);
          const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(lf, hf, 0, false);
          l[j] = __builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffff));
        }
      }
      hdst[(size_t)p * C8] = h;
      if (ldst) ldst[(size_t)p * C8] = l;
    }
  }
}

// groups per workgroup for the fused kernel (0: the tensor does not fit the one-sweep form)
int gn_fused16_groups(int HW, int C0, int C1, int G) {
  const int C = C0 + C1;
  if (G <= 0 || C % G || C0 % 8 || C1 % 8) return 0;
  const int cpg = C / G;
  for (int GB = 1; GB <= G; GB *= 2) {
    const int cb = cpg * GB;
    if (G % GB || cb % 8) continue;
    if (cb > GNFU_THREADS) return 0;
    const int rows = GNFU_THREADS / (cb / 8);
    return (HW + rows - 1) / rows <= GNFU_MAXS ? GB : 0;
  }
  return 0;
}

int gn_fused16_launch(const float* src0, const float* src1, int C0, int C1, const float* gamma, const float* beta, float eps,
                      void* hi, void* lo, int B, int HW, int G, int act, hipStream_t s, int f8, float* nscale, float* nshift) {
  const int GB = gn_fused16_groups(HW, C0, C1, G);
  CSD_REQUIRE(GB > 0, "gn_fused16: %d pixels x %d+%d channels in %d groups does not fit the one-sweep form", HW, C0, C1, G);
  hipLaunchKernelGGL(gn_fused16_kernel, dim3(G / GB, B), dim3(GNFU_THREADS), 0, s, src0, src1, C0, C1, gamma, beta, eps,
                     static_cast<half8_t*>(hi), static_cast<half8_t*>(lo), HW, G, GB, act, f8, nscale, nshift);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int gn_apply16_launch(const float* src0, const float* src1, int C0, int C1, const float* nscale, const float* nshift,
                      void* hi, void* lo, int B, int HW, int act, hipStream_t s, int f8) {
  const int C = C0 + C1;
  CSD_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0 && C / 8 <= GA_THREADS, "gn_apply16: channels must be multiples of 8, <= 2048");
  int nchunk = cdiv(4096, B);
  const int maxchunk = cdiv(HW, 16);
  if (nchunk > maxchunk) nchunk = maxchunk;
  if (nchunk < 1) nchunk = 1;
  hipLaunchKernelGGL(gn_apply16_kernel, dim3(nchunk, B), dim3(GA_THREADS), 0, s, src0, src1, C0, C1, nscale, nshift,
                     static_cast<half8_t*>(hi), static_cast<half8_t*>(lo), HW, act, nchunk, f8);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int gn_plan(GNPlan* p, int B, int HW, int C0, int C1, int G) {
  const int C = C0 + C1;
  CSD_REQUIRE(C % 4 == 0 && C0 % 4 == 0 && C <= 1024, "groupnorm: C=%d+%d unsupported", C0, C1);
  CSD_REQUIRE(G > 0 && C % G == 0, "groupnorm: %d channels not divisible into %d groups", C, G);
  p->B = B; p->HW = HW; p->C0 = C0; p->C1 = C1; p->G = G;
  // ~2048 workgroups per launch, each chunk at least ~64 pixel rows
  int nchunk = cdiv(2048, B);
  const int maxchunk = cdiv(HW, 64);
  if (nchunk > maxchunk) nchunk = maxchunk;
  if (nchunk < 1) nchunk = 1;
  p->nchunk = nchunk;
  return CSD_OK;
}

size_t gn_partial_bytes(const GNPlan& p) {
  return (size_t)p.B * p.nchunk * p.G * 2 * sizeof(double);
}

int gn_stats_launch(const GNPlan& p, const float* src0, const float* src1, double* partial, hipStream_t s, int per_channel) {
  const int C = p.C0 + p.C1;
  const size_t lds = (size_t)2 * C * sizeof(double);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(p.nchunk, p.B), dim3(GN_THREADS), lds, s, src0, src1, partial,
                     p.HW, p.C0, p.C1, p.G, p.nchunk, per_channel);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int gn_finalize_launch(const GNPlan& p, const double* partial, const float* gamma, const float* beta,
                       float eps, float* nscale, float* nshift, hipStream_t s, float* rs, float* ms) {
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.B), dim3(256), 0, s, partial, gamma, beta, eps, nscale,
                     nshift, p.HW, p.C0 + p.C1, p.G, p.nchunk, rs, ms);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int gn_finalize_tiles_launch(const double* p0, int tpi0, int C0, const double* p1, int tpi1, int C1, int B, int HW, int G,
                             const float* gamma, const float* beta, float eps, float* nscale, float* nshift,
                             hipStream_t s) {
  CSD_REQUIRE((C0 + C1) % G == 0 && (C0 + C1) / G <= 64, "gn finalize: %d channels in %d groups", C0 + C1, G);
  const int cpg = (C0 + C1) / G;
  int GB = 32 / cpg;                       // groups per workgroup: about 32 channels (a 512-byte row of partials)
  if (GB < 1) GB = 1;
  if (GB > G) GB = G;
  hipLaunchKernelGGL(gn_finalize_tiles_kernel, dim3(cdiv(G, GB), B), dim3(GNF_THREADS), 0, s, p0, tpi0, C0, p1, tpi1, C1,
                     gamma, beta, eps, nscale, nshift, HW, G, GB);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int gn_apply_launch(const float* x, const float* nscale, const float* nshift, float* y, int B, int HW, int C,
                    int act, hipStream_t s, float* mask, float p_drop, uint64_t seed, uint64_t stream_id, void* hi, void* lo) {
  const size_t total4 = (size_t)B * HW * C / 4;
  const int grid = (int)std::min<size_t>(cdiv64(total4, 256), 2048 * 4);
  if (mask)
    hipLaunchKernelGGL(gn_apply_dropout_kernel, dim3(grid), dim3(256), 0, s, x, nscale, nshift, y, mask, HW, C, act, total4, p_drop, seed,
                       stream_id, static_cast<half4_t*>(hi), static_cast<half4_t*>(lo));
  else
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid), dim3(256), 0, s, x, nscale, nshift, y, HW, C, act, total4, static_cast<half4_t*>(hi),
                       static_cast<half4_t*>(lo));
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

}  // namespace csd
