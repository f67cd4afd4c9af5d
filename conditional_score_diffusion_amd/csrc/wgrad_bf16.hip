// wgrad_bf16.hip - convolution weight gradient on the bf16 matrix cores with split operands (3x3 / 1x1, stride 1).
//
// dW[co][ci][tap] = sum over pixels of dY[p][co] * X[p + tap][ci]: the contraction index is the PIXEL, and
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive K values per lane - eight pixels of one channel.  Activations are NHWC
// fp32 (a wave-load of one pixel is 32 consecutive channels = 128 B per half-wave), so a lane gathers its 8 pixels with
// 8 dword loads (lanes 0-31: pixels 0..7 of the block, lanes 32-63: pixels 8..15) and converts on the fly:
//     hi = x with the low 16 bits cleared (= a bf16), lo = bf16(x - hi)            (x - hi is exact in fp32)
//     dY*X ~= hi*hi + hi*lo + lo*hi   -> 3 MFMAs, fp32 accumulate, ~2^-16 relative error per product
// bf16 rather than fp16 halves: gradients span the whole fp32 exponent range.  The fp32 kernel (backward.hip) needs
// 64 MFMA cycles per 2 pixels; this one 96 per 16: 5.3x fewer matrix-core cycles, and 38 instead of 80 wave-loads per 16
// pixels because a row's 10 input pixels are loaded ONCE and serve the three kx taps (register windows q..q+7).
//
// STAGED schedule (maps wider than 8, channels % 4 == 0): the gathers above are TA-bound - a 64-lane dword wave-load costs
// ~16 cycles of the CU's texture addresser however well it coalesces, 38 of them per 16 pixels per wave.  Instead ONE
// global_load_lds_dwordx4 wave-instruction brings 8 pixels x 32 channels (1 KiB, lane = pixel x channel quad) straight into a
// wave-private LDS patch in memory order [pixel][32 channels] (11 per step: 2 for dY, 3 per input row), and each lane gathers its
// 8 / 10 pixels of its channel from LDS with ds_read_b32 (bank = channel: conflict-free within a half wave).  No staging
// registers, no workgroup barrier (the patch is private to the wave): wait vmcnt(0), read, wait lgkmcnt(0), issue the next step's
// loads, then convert and run the MFMAs while they are in flight.
//
// Tile per wave: 32 couts x 32 cins x all taps (144 accumulators), K over output rows; 4 waves of a workgroup take
// interleaved rows of one K-split and are reduced through LDS in fixed order; split partials are summed in fp64 in split
// order by the caller (bit-reproducible).  Feature maps of width <= 8 put two ROWS into the two lane halves instead of
// two 8-pixel segments of one row.
#include <stdlib.h>

#include "common.h"

namespace csd {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi(float lo_elem, float hi_elem) {   // (bf16(hi_elem) << 16) | bf16(lo_elem), truncating
  return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }

template <int KS, bool STAGED, bool FULL>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const float* x, const float* dy, float* partial, int B, int IH,
                                                                 int IW, int Cin, int OH, int OW, int Cout, int per_split, int n_ci,
                                                                 int n_co) {
  constexpr int TAPS = KS * KS, PAD = KS / 2, NQ = 8 + KS - 1, NP = NQ / 2;
  constexpr int BPIX = KS == 3 ? 24 : 16;                    // staged pixels per input row (16 + halo, rounded up to 8)
  constexpr int WAVE_LDS = (16 + KS * BPIX) * 32;            // floats of one wave's patch: dY 16 pixels, KS rows of x
  constexpr int SM = (STAGED && 4 * WAVE_LDS > TAPS * 1024) ? 4 * WAVE_LDS : TAPS * 1024;
  __shared__ __attribute__((aligned(16))) float red[SM];     // the patches during the K loop, the cross-wave reduction after it
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = lane & 31, kg = lane >> 5;
  // XCD-aware order: the (cout tile, cin tile) workgroups of one K-split are neighbours on one XCD (they share x / dy rows)
  const unsigned tiles = (unsigned)n_ci * n_co, total = gridDim.x;
  const unsigned per_xcd = (total + 7) / 8;
  unsigned wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((total & 7) != 0) wid = blockIdx.x;
  const unsigned split_id = wid / tiles, tile_id = wid - split_id * tiles;
  const int tz = (int)(tile_id / n_ci), ty = (int)(tile_id - (unsigned)tz * n_ci);
  const int co = tz * 32 + m, ci = ty * 32 + m;
  const bool cov = co < Cout, civ = ci < Cin;
  const unsigned cio = civ ? ci : 0, coo = cov ? co : 0;
  const unsigned nrows = (unsigned)B * OH;
  const unsigned r_begin = split_id * (unsigned)per_split;
  const unsigned r_end = r_begin + per_split < nrows ? r_begin + per_split : nrows;
  const bool pair_rows = OW <= 8;          // lane halves = two rows (narrow maps) instead of two segments of one row
  floatx16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // this lane's 8 output pixels of a step: row `row`, columns oxb .. oxb+7
  auto lane_pos = [&](unsigned r, int ox0, unsigned& row, int& oxb) {
    if (pair_rows) { row = r + kg; oxb = 0; }
    else { row = r; oxb = ox0 + 8 * kg; }
  };
  // raw operands: a[j] = dy[row][oxb + j][co]; xin[ky][q] = x[b][oy + ky - PAD][oxb - PAD + q][ci]; loads are unconditional on
  // clamped addresses, masks (bit j of am, bit q of xm[ky]) are applied when the values are converted
  auto fetch = [&](unsigned r, int ox0, float (&a)[8], float (&xin)[KS][NQ], unsigned& am, unsigned (&xm)[KS]) {
    unsigned row;
    int oxb;
    lane_pos(r, ox0, row, oxb);
    const bool rv = row < r_end;
    const unsigned rc = rv ? row : r_begin;
    const unsigned b = rc / (unsigned)OH, oy = rc - b * (unsigned)OH;
    const float* dyp = dy + (size_t)(rc * (unsigned)OW) * Cout + coo;
    am = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ox = oxb + j;
      const bool v = rv && cov && ox < OW;
      a[j] = dyp[(unsigned)(v ? ox : 0) * (unsigned)Cout];
      am |= (v ? 1u : 0u) << j;
    }
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int iy = (int)oy + ky - PAD;
      const bool rowok = rv && civ && iy >= 0 && iy < IH;
      const float* xp = x + (size_t)((b * (unsigned)IH + (rowok ? (unsigned)iy : 0u)) * (unsigned)IW) * Cin + cio;
      xm[ky] = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ix = oxb - PAD + q;
        const bool ok = rowok && ix >= 0 && ix < IW;
        xin[ky][q] = xp[(unsigned)(ok ? ix : 0) * (unsigned)Cin];
        xm[ky] |= (ok ? 1u : 0u) << q;
      }
    }
  };

  // ---- STAGED: global -> LDS (direct), LDS -> registers -------------------------------------------------------------
  float* const patch = red + wave * WAVE_LDS;
  const int sp = lane >> 3, sq = (lane & 7) * 4;             // this lane's pixel (within an 8-pixel group) and channel quad
  auto glds16 = [&](const float* src, float* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // (VALU instructions are matrix-pipe time on this chip - tools/probes/mfma_valu_overlap.hip - so everything wave-uniform is kept on the
  // scalar unit: row bases in SGPRs, 32-bit element offsets, one 64-bit add per DMA)
  const unsigned coq = (unsigned)(tz * 32 + sq) < (unsigned)Cout ? (unsigned)(tz * 32 + sq) : 0u;
  const unsigned ciq = (unsigned)(ty * 32 + sq) < (unsigned)Cin ? (unsigned)(ty * 32 + sq) : 0u;
  auto stage = [&](unsigned r, int ox0) {                    // (row r is wave-uniform here: no pair mode)
    const unsigned rs = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    const int oxs = __builtin_amdgcn_readfirstlane(ox0);
    const unsigned rc = rs < r_end ? rs : r_begin;
    const unsigned b = rc / (unsigned)OH, oy = rc - b * (unsigned)OH;
    const float* const dyrow = dy + (size_t)rc * (unsigned)OW * (unsigned)Cout;          // scalar
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int ox = oxs + sp + 8 * i;
      ox = ox < OW ? ox : OW - 1;
      glds16(dyrow + ((unsigned)ox * (unsigned)Cout + coq), patch + i * 256);
    }
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      int iy = (int)oy + ky - PAD;
      iy = iy < 0 ? 0 : (iy >= IH ? IH - 1 : iy);
      const float* const xrow = x + (size_t)((b * (unsigned)IH + (unsigned)iy) * (unsigned)IW) * (unsigned)Cin;   // scalar
#pragma unroll
      for (int i = 0; i < BPIX / 8; ++i) {
        int ix = oxs - PAD + sp + 8 * i;
        ix = ix < 0 ? 0 : (ix >= IW ? IW - 1 : ix);
        glds16(xrow + ((unsigned)ix * (unsigned)Cin + ciq), patch + (16 + ky * BPIX) * 32 + i * 256);
      }
    }
  };
  // full channel tiles: instead of masking every gathered value in registers, the few out-of-image pixels of a step are zeroed in
  // the wave's patch (wave-uniform ranges: scalar branches, a handful of ds_writes on border steps only) and the gather is plain
  constexpr bool full_ch = FULL;                             // (host: Cout % 32 == 0 && Cin % 32 == 0, staged)
  auto zero_oob = [&](unsigned r, int ox0) {
    const unsigned rs = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    const int oxs = __builtin_amdgcn_readfirstlane(ox0);
    const unsigned b = rs / (unsigned)OH, oy = rs - b * (unsigned)OH;
    // dY pixels j >= OW - ox0
    for (int j = OW - oxs + kg; j < 16; j += 2) patch[j * 32 + m] = 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int iy = (int)oy + ky - PAD;
      float* const row = patch + (16 + ky * BPIX) * 32 + m;
      if (iy < 0 || iy >= IH) {
        for (int pp = kg; pp < 16 + KS - 1; pp += 2) row[pp * 32] = 0.f;
      } else {
        if (oxs - PAD < 0 && kg == 0) row[0] = 0.f;                                  // ix = -1
        for (int pp = IW - oxs + PAD + kg; pp < 16 + KS - 1; pp += 2) row[pp * 32] = 0.f;     // ix >= IW
      }
    }
  };
  // operands of this lane from the patch + the same masks as fetch()
  auto gather = [&](unsigned r, int ox0, float (&a)[8], float (&xin)[KS][NQ], unsigned& am, unsigned (&xm)[KS]) {
    const bool rv = r < r_end;
    const unsigned rc = rv ? r : r_begin;
    const unsigned b = rc / (unsigned)OH, oy = rc - b * (unsigned)OH;
    const int oxb = ox0 + 8 * kg;
    am = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = patch[(8 * kg + j) * 32 + m];
      am |= ((rv && cov && oxb + j < OW) ? 1u : 0u) << j;
    }
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int iy = (int)oy + ky - PAD;
      const bool rowok = rv && civ && iy >= 0 && iy < IH;
      xm[ky] = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int ix = oxb - PAD + q;
        xin[ky][q] = patch[(16 + ky * BPIX + 8 * kg + q) * 32 + m];
        xm[ky] |= ((rowok && ix >= 0 && ix < IW) ? 1u : 0u) << q;
      }
    }
  };

  unsigned r = r_begin + (pair_rows ? 2 * wave : wave);
  const unsigned r_step = pair_rows ? 8 : 4;
  int ox0 = 0;
  float a_c[8], x_c[KS][NQ];
  unsigned am_c, xm_c[KS];
  float a_n[8], x_n[KS][NQ];
  unsigned am_n, xm_n[KS];
  if (STAGED) stage(r, ox0);
  else fetch(r, ox0, a_c, x_c, am_c, xm_c);
  while (r < r_end) {
    unsigned rn = r;
    int oxn = ox0 + 16;
    if (pair_rows || oxn >= OW) { oxn = 0; rn = r + r_step; }
    if (STAGED) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this step's patch has landed
      if (full_ch) {
        zero_oob(r, ox0);
#pragma unroll
        for (int j = 0; j < 8; ++j) a_c[j] = patch[(8 * kg + j) * 32 + m];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int q = 0; q < NQ; ++q) x_c[ky][q] = patch[(16 + ky * BPIX + 8 * kg + q) * 32 + m];
        am_c = 0xffu;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) xm_c[ky] = (1u << NQ) - 1u;
      } else {
        gather(r, ox0, a_c, x_c, am_c, xm_c);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... and is in registers: the patch may be overwritten
      stage(rn, oxn);                                           // next step's 11 loads fly under the conversions + MFMAs
    } else {
      fetch(rn, oxn, a_n, x_n, am_n, xm_n);        // next step's 8 + KS*NQ loads in flight under this step's conversions + MFMAs
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // dY -> hi / lo bf16x8
    uintx4 ah, al;
    {
      float v[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = (FULL || ((am_c >> j) & 1u)) ? a_c[j] : 0.f;
        lo[j] = v[j] - trunc_bf16(v[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ah[j] = pack_hi(v[2 * j], v[2 * j + 1]);
        al[j] = pack_hi(lo[2 * j], lo[2 * j + 1]);
      }
    }
    const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Al = __builtin_bit_cast(bf16x8, al);
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      float v[NQ], lo[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        v[q] = (FULL || ((xm_c[ky] >> q) & 1u)) ? x_c[ky][q] : 0.f;
        lo[q] = v[q] - trunc_bf16(v[q]);
      }
      // even-aligned pairs P_i = (q = 2i, 2i+1) serve kx = 0 (P0..P3) and kx = 2 (P1..P4); odd-aligned Q_i = (2i+1, 2i+2) serve kx = 1
      unsigned ph[NP], pl[NP], qh[4], ql[4];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        ph[i] = pack_hi(v[2 * i], v[2 * i + 1]);
        pl[i] = pack_hi(lo[2 * i], lo[2 * i + 1]);
      }
      if (KS == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          qh[i] = pack_hi(v[2 * i + 1], v[2 * i + 2]);
          ql[i] = pack_hi(lo[2 * i + 1], lo[2 * i + 2]);
        }
      }
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        uintx4 bh, bl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bh[i] = (kx == 1) ? qh[i] : ph[i + kx / 2];
          bl[i] = (kx == 1) ? ql[i] : pl[i + kx / 2];
        }
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bl = __builtin_bit_cast(bf16x8, bl);
        const int t = ky * KS + kx;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (!STAGED) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a_c[j] = a_n[j];
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) x_c[ky][q] = x_n[ky][q];
        xm_c[ky] = xm_n[ky];
      }
      am_c = am_n;
    }
    r = rn;
    ox0 = oxn;
  }
  if (STAGED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the last, unused prefetch must not land in the reduction buffer)
    __syncthreads();                                            // every wave is done with its patch: red[] is reused below
  }

  // cross-wave reduction (fixed order: wave 0, 1, 2, 3)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float* d = &red[(t * 16 + q) * 64 + lane];
          *d = (w == 0) ? acc[t][q] : *d + acc[t][q];
        }
    }
    __syncthreads();
  }
  float* dst = partial + (size_t)split_id * Cout * Cin * TAPS;
  for (int e = threadIdx.x; e < TAPS * 1024; e += 256) {
    const int l = e & 63, q = (e >> 6) & 15, t = e >> 10;
    const int rco = tz * 32 + (q & 3) + 8 * (q >> 2) + 4 * (l >> 5);
    const int rci = ty * 32 + (l & 31);
    if (rco < Cout && rci < Cin) dst[((size_t)rco * Cin + rci) * TAPS + t] = red[e];
  }
}

int wgrad_bf16_launch(const float* xh, const float* dyh, float* partial, int B, int H, int W, int Cin, int Cout, int ksize, int S,
                      int per_split, hipStream_t s) {
  const int n_ci = cdiv(Cin, 32), n_co = cdiv(Cout, 32);
  const dim3 grid((unsigned)S * n_ci * n_co);
  const bool staged = W > 8 && Cin % 4 == 0 && Cout % 4 == 0 && !getenv("CSD_WGRAD_GATHER");
  const bool full = staged && Cin % 32 == 0 && Cout % 32 == 0 && !CSD_TUNE_ENV("CSD_WGRAD_MASKED");
#define WG_LAUNCH(KS_, ST_, FU_)                                                                                                   \
  hipLaunchKernelGGL((conv_wgrad_bf16_kernel<KS_, ST_, FU_>), grid, dim3(256), 0, s, xh, dyh, partial, B, H, W, Cin, H, W, Cout,   \
                     per_split, n_ci, n_co)
  if (ksize == 3) { if (full) WG_LAUNCH(3, true, true); else if (staged) WG_LAUNCH(3, true, false); else WG_LAUNCH(3, false, false); }
  else { if (full) WG_LAUNCH(1, true, true); else if (staged) WG_LAUNCH(1, true, false); else WG_LAUNCH(1, false, false); }
#undef WG_LAUNCH
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

}  // namespace csd
