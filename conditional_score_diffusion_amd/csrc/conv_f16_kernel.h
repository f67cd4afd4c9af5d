// conv_f16.hip - 3x3 (stride 1, optional fused nearest-x2) convolution as an implicit GEMM on the
// gfx950 fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate).
//
// Why: in fp32 the network is compute-bound at 157 TFLOP/s; the fp16 MFMA rate is
// 16x higher.  Two arithmetic modes share this kernel (template NS):
//   NS = 2  "F16X3": every operand is carried as hi + lo fp16 (hi = fp16(v), lo = fp16(v - hi)), and
//           each product is formed as ah*bh + ah*bl + al*bh (three MFMAs, the 2^-22-relative al*bl
//           term is dropped).  hi*hi is exact in fp32, accumulation is fp32 -> fp32-class results
//           (1.7e-6 on the whole network, tests/) at 1/3 of the fp16 rate = 5.3x the fp32 rate.
//   NS = 1  "F16":   plain fp16 operands, fp32 accumulate.
// Weights are pre-scaled by 2^8 at pack time (exact) so their lo parts stay in the fp16 normal
// range; the epilogue multiplies by 2^-8 (exact).
//
// Work decomposition ("waves split N"): a workgroup owns MT*32 output pixels (MT in {2,4,8}) of the
// virtual tall image x NW cout tiles; it has NW waves and wave w owns ALL the pixels x cout tile w.
//   * weights: every weight byte is fetched by exactly ONE wave of the workgroup, straight from L2
//     in MFMA-fragment order (1 KiB per wave-instruction), and reused for MT MFMAs per product term
//     - the earlier "waves split M" layout re-fetched each fragment in all 4 waves and was bound by
//     the 64 B/clk/CU L1 path (ablation in profiles/);
//   * activations: the staged halo patch (fp16 hi|lo planes, 16 channels per K chunk, built WHILE
//     staging with the fused GroupNorm affine + SiLU) is read by every wave - 1 KiB of ds_read_b128
//     per MFMA, i.e. 128 B/clk/CU at the full fp16 rate = half the LDS peak;
//   * staging is cut into pieces interleaved with the K steps of the current chunk, so its VALU work
//     (norm, SiLU, fp16 split) hides under the MFMAs and only 2 float4 per lane are in flight.
//   One barrier per K chunk (double-buffered patch); no barrier on the weight path.
#pragma once
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "common.h"

namespace csd {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

#ifndef CSD_CONV_ABLATE
#define CSD_CONV_ABLATE 0   // tuning aid: 1 no B loads, 2 no A LDS reads, 4 no staging/barriers, 8 no masks, 16 no norm/act
#endif
#ifdef CSD_C16_TIMING
#define C16_TSTAMP(i) do { if (k.a.dbg && tid == 0 && blockIdx.x < 4095) k.a.dbg[blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
#define C16_TSTAMP(i) do { } while (0)
#endif
#define C16_KC 16               // channels per K chunk (= one MFMA K step per tap)
#define C16_PIECE 2             // staging slots (float4 per lane) per interleaved piece
#define C16_MAX_PIECES 5
#define C16_WSCALE 256.0f       // weight pre-scale (power of two: exact)
// LDS patch row pitch in pixels is a template parameter (24 or 34): compile-time, so every tap offset is
// an immediate.  34 serves TW = 32 tiles (one 32-pixel row per M tile: conflict-free ds_read_b128).
#define C16_LDS_LIMIT (72 * 1024)   // patch double buffer budget: keeps 2 workgroups per CU

struct Conv16KArgs {
  ConvArgs a;
  int B, IH, IW, OH, OW, C0, C1, Cout;
  int stride, pad, up;
  int TH, TW, PH, PW, tiles_x, n_groups, nblocks;
  int nck, nw;              // K chunks; waves (= cout tiles) per workgroup
  int ntiles_n;             // 32-cout tiles of the layer
};

__device__ __forceinline__ float4 gload4f(const float* p) {
  const v4f_t v = *(const __attribute__((address_space(1))) v4f_t*)(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ half8 gload_h8(const char* p) {
  return *(const __attribute__((address_space(1))) half8*)(p);
}

__device__ __forceinline__ float act16(float v, int act) {
  switch (act) {
    case CSD_ACT_SWISH: return v * __frcp_rn(1.0f + __expf(-v));
    case CSD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case CSD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// MASK: per-tap validity masks are only needed when a pixel tile can straddle two images of the
// virtual tall image (TH does not divide OH: the 20x20 / 10x10 / 5x5 levels).  Otherwise every
// out-of-image tap reads a zero the staging wrote, and no VALU is spent on masking.
// IN16: the source is already fp16 (hi plane = g_src0, lo plane = g_src1 when NS == 2; one tensor of
// C0 channels) - written by gn_apply16_kernel (GroupNorm affine + activation + fp16 split applied ONCE
// per element) - and staging is a pure 16-byte copy.  Otherwise the source is fp32 NHWC (two-source
// virtual concat allowed) and norm/act/convert run while staging (measured: that VALU work, repeated for
// every halo pixel and every cout group, costs as much as all the MFMAs of the F16 mode).
// KCS (fp16 sources only): 16-channel sub-chunks per K STAGE.  At fp16 MFMA speed a 16-channel chunk
// is ~0.5 us of matrix work while an HBM round trip is ~3 us, so chunk-by-chunk double buffering cannot
// hide the staging latency.  With KCS > 1 the whole patch of a stage (48 channels in F16, 32 in F16X3:
// ~21-29 KB) is fetched in ONE burst - for Cin <= 96 that is the entire K extent - and the next stage's
// burst is issued a full stage (>= 1.4 us of MFMAs) before it is needed.  Single LDS buffer.
template <int MT, int NS, bool MASK, int PWC, bool IN16, int KCS>
__global__ __launch_bounds__(192, 2) void conv_f16_kernel(const void* __restrict__ g_src0v,
                                                         const void* __restrict__ g_src1v,
                                                         const char* __restrict__ g_wpack,
                                                         const Conv16KArgs k) {
  constexpr int TAPS = 9, KS = 3;
  static_assert(KCS == 1 || IN16, "multi-chunk stages need an fp16 source");
  constexpr int LO = 32 * KCS;               // byte offset of the lo plane inside a staged pixel
  constexpr int PSB = 32 * KCS * NS + 16;    // bytes per staged pixel: [hi KCS*16 ch][lo KCS*16 ch] + pad
  constexpr int NU = (KCS > 1) ? (NS == 1 ? 7 : 9) : C16_MAX_PIECES * C16_PIECE;    // staging units per lane (max)
  constexpr int NPIX = MT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  const float* const g_src0 = static_cast<const float*>(g_src0v);
  const float* const g_src1 = static_cast<const float*>(g_src1v);

  const int tid = threadIdx.x;
  C16_TSTAMP(0);
  const int nthr = k.nw * 64;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 5;                  // which 8 of the 16 K values this lane feeds

  int w;
  {
    const int bid = blockIdx.x, nb = k.nblocks;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ng = w % k.n_groups;
  const int tile = w / k.n_groups;
  const int tile_y = tile / k.tiles_x;
  const int tile_x = tile - tile_y * k.tiles_x;
  const int ov0 = tile_y * k.TH;
  const int ox0 = tile_x * k.TW;

  const int S = k.stride, P = k.pad, U = k.up;
  const int prow0 = (ov0 * S - P) >> U;
  const int pcol0 = (ox0 * S - P) >> U;
  const int patch_bytes = k.PH * PWC * PSB;
  char* const buf0 = smem16;
  char* const buf1 = (KCS > 1) ? smem16 : smem16 + patch_bytes;       // single buffer in staged mode
  int* const otab = reinterpret_cast<int*>(smem16 + ((KCS > 1) ? 1 : 2) * patch_bytes);   // [NPIX] output pixel index
  int* const btab = otab + NPIX;                                        // [NPIX] sample index
  const int npatch = k.PH * k.PW;
  int* const stab = btab + NPIX;          // [npatch] staging: source pixel index (or -1)
  int* const dtab = stab + npatch;        // [npatch] staging: LDS byte offset of the pixel inside a patch buffer
  int* const ntab = dtab + npatch;        // [npatch] staging: sample * Cin (GroupNorm scale/shift row)

  // ---- per-lane pixels: one per M tile.  LDS byte offset of tap (0,0) (+ this lane's K half),
  // 3+3 validity bits and (x2-upsample mode) the two parity bits that make the tap map non-affine.
  int base[MT];
  unsigned vbits[MT];         // bit r: row tap r valid; bit 3+s: col tap s valid; bit 6/7: row/col parity
  constexpr int rstride = PWC * PSB;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + (lane & 31);
    const int ty = m / k.TW;
    const int tx = m - ty * k.TW;
    const int ov = ov0 + ty, ox = ox0 + tx;
    const bool mv = (m < k.TH * k.TW) && (ov < k.B * k.OH) && (ox < k.OW);
    const int b = ov / k.OH;
    const int oy = ov - b * k.OH;
    if (wave == 0 && kg == 0) {
      otab[m] = mv ? (ov - ov0) * k.OW + ox : -1;   // relative to the tile's first row
      btab[m] = mv ? b : 0;
    }
    const int IHe = k.IH << U, IWe = k.IW << U;
    unsigned vb = 0;
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      const int iy = oy * S + r - P, ix = ox * S + r - P;
      vb |= ((mv && iy >= 0 && iy < IHe) ? 1u : 0u) << r;
      vb |= ((mv && ix >= 0 && ix < IWe) ? 1u : 0u) << (3 + r);
    }
    const int r0 = ov * S - P, c0 = ox * S - P;          // tap (0,0) coordinate before the >> U
    vb |= (unsigned)(r0 & U) << 6;
    vb |= (unsigned)(c0 & U) << 7;
    vbits[mt] = vb;
    base[mt] = mv ? ((r0 >> U) - prow0) * rstride + ((c0 >> U) - pcol0) * PSB + kg * 16 : kg * 16;
  }
  // tap (r,s) of pixel mt lives at base + roff(r) + coff(s):  U == 0: r*rstride, s*PSB
  //                                                          U == 1: ((r + parity) >> 1) * stride
  auto tap_off = [&](int mt, int r, int s) -> int {
    if (U == 0) return base[mt] + r * rstride + s * PSB;
    const int pr = (vbits[mt] >> 6) & 1, pc = (vbits[mt] >> 7) & 1;
    return base[mt] + ((r + pr) >> 1) * rstride + ((s + pc) >> 1) * PSB;
  };

  // ---- staging, in pieces of C16_PIECE 16-byte slots per lane ----
  // fp32 source: a slot = 4 channels of one patch pixel (4 slots per pixel per chunk)
  // fp16 source: a slot = 8 channels of one plane    (2*NS slots per pixel per chunk)
  // The pixel -> (source pixel, LDS destination, sample) map is the same for every K chunk: it is
  // computed once (integer divisions, image-border tests) into three small LDS tables.
  constexpr int SPP = IN16 ? 2 * NS * KCS : 4;    // slots per patch pixel per stage
  const int total4 = npatch * SPP;
  const int Cin = k.C0 + k.C1;
  for (int pix = tid; pix < npatch; pix += nthr) {
    const int pr = pix / k.PW;
    const int pc = pix - pr * k.PW;
    const int vr = prow0 + pr, col = pcol0 + pc;
    // Without per-tap masks (MASK == false: the tile lies inside ONE image) the halo must read zeros
    // outside THAT image - rows of the neighbouring image of the virtual tall tensor are padding too.
    const int img_lo = MASK ? 0 : (ov0 / k.OH) * k.IH;
    const int img_hi = MASK ? k.B * k.IH : img_lo + k.IH;
    const bool inimg = vr >= img_lo && vr < img_hi && col >= 0 && col < k.IW;
    stab[pix] = inimg ? vr * k.IW + col : -1;
    dtab[pix] = (pr * PWC + pc) * PSB;
    ntab[pix] = inimg ? (vr / k.IH) * Cin : 0;
  }
  __syncthreads();
  const int per_piece = nthr * C16_PIECE;
  const int npieces = (total4 + per_piece - 1) / per_piece;            // <= C16_MAX_PIECES (host checked)
  float4 stage[C16_PIECE], st_sc[C16_PIECE], st_sh[C16_PIECE];   // raw 16 bytes (+ GroupNorm scale/shift)
  int st_pix[C16_PIECE];      // source pixel index (or -1) of the slots of the piece in flight
  const bool has_norm = !IN16 && !(CSD_CONV_ABLATE & 16) && k.a.nscale != nullptr;
  // slot e of the patch -> {source address, LDS byte offset}
  auto slot_load = [&](int e, int cb, float4& v, float4& sc, float4& sh) -> int {
    int sp = -1;
    v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total4) {
      const int pix = e / SPP, sub = e - pix * SPP;
      sp = stab[pix];
      if (sp >= 0) {
        if (IN16) {     // sub = plane * (2*KCS) + sub-chunk * 2 + half
          const int pl = sub / (2 * KCS), rest = sub - pl * (2 * KCS);
          const _Float16* plane = static_cast<const _Float16*>(pl ? g_src1v : g_src0v);
          v = gload4f(reinterpret_cast<const float*>(plane + (size_t)sp * Cin + cb + rest * 8));
        } else {
          const float* src;
          int Cs, coff;
          if (cb < k.C0) { src = g_src0; Cs = k.C0; coff = cb; }
          else { src = g_src1; Cs = k.C1; coff = cb - k.C0; }
          v = gload4f(src + (size_t)sp * Cs + coff + sub * 4);
          if (has_norm) {     // issued together with the data: no dependent L2 round trip at conversion time
            const int nb = ntab[pix] + cb + sub * 4;
            sc = gload4f(k.a.nscale + nb);
            sh = gload4f(k.a.nshift + nb);
          }
        }
      }
    }
    return sp;
  };
  auto slot_store = [&](char* buf, int e, int sp, float4 v, const float4& sc, const float4& sh) {
    if (e >= total4) return;
    const int pix = e / SPP, sub = e - pix * SPP;
    char* dst = buf + dtab[pix];
    if (IN16) {
      const int pl = sub / (2 * KCS), rest = sub - pl * (2 * KCS);
      *reinterpret_cast<float4*>(dst + pl * LO + rest * 16) = v;     // raw copy of 8 halves
      return;
    }
    if (has_norm && sp >= 0) {
      v.x = act16(v.x * sc.x + sh.x, k.a.act);
      v.y = act16(v.y * sc.y + sh.y, k.a.act);
      v.z = act16(v.z * sc.z + sh.z, k.a.act);
      v.w = act16(v.w * sc.w + sh.w, k.a.act);
    }
    dst += sub * 8;
    half4 hi;
    hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
    *reinterpret_cast<half4*>(dst) = hi;
    if (NS == 2) {
      half4 lo;
      lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
      lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
      *reinterpret_cast<half4*>(dst + LO) = lo;
    }
  };
  auto stage_load = [&](int ck, int piece) {
#pragma unroll
    for (int j = 0; j < C16_PIECE; ++j)
      st_pix[j] = slot_load(piece * per_piece + j * nthr + tid, ck * C16_KC, stage[j], st_sc[j], st_sh[j]);
  };
  auto stage_write_slot = [&](char* buf, int ck, int piece, int j) {
    slot_store(buf, piece * per_piece + j * nthr + tid, st_pix[j], stage[j], st_sc[j], st_sh[j]);
  };
  auto stage_write = [&](char* buf, int ck, int piece) {
#pragma unroll
    for (int j = 0; j < C16_PIECE; ++j) stage_write_slot(buf, ck, piece, j);
  };

  floatx16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // weight stream of THIS wave's cout tile: [chunk][tap][plane hi|lo][lane][8 halves]; NS KiB per step
  constexpr int STEP_BYTES = NS * 1024;
  const size_t tile_stride = (size_t)k.nck * TAPS * STEP_BYTES;     // k.nck counts 16-channel chunks
  // (a wave beyond the last cout tile - tiny-Cout layers run 3 waves for the staging bandwidth - recomputes
  // the last tile and stores nothing: its columns are >= Cout)
  const int wtile = min(ng * k.nw + wave, k.ntiles_n - 1);
  const char* wstep = g_wpack + (size_t)wtile * tile_stride + lane * 16;

  // weight-fragment ring, BR-1 K steps ahead of the MFMAs: one step is only MT*32 (F16) / MT*96 (F16X3)
  // MFMA cycles while an L2 hit costs ~500-800, so the prefetch distance must be several steps.
  // BR divides the 9 steps of a chunk, which keeps the ring index a compile-time constant.
#ifdef CSD_C16_BR
  constexpr int BR = CSD_C16_BR;
#else
  constexpr int BR = 3;
#endif
  half8 breg[BR][NS];
#pragma unroll
  for (int q = 0; q < BR - 1; ++q)
#pragma unroll
    for (int p = 0; p < NS; ++p) breg[q][p] = gload_h8(wstep + (size_t)q * STEP_BYTES + p * 1024);
  wstep += (size_t)(BR - 2) * STEP_BYTES;     // points at the newest prefetched step

  C16_TSTAMP(1);   // tables built
  float4 sv_first[(KCS > 1 && NS == 1) ? NU : 1];   // staged mode: burst of stage 1, issued together with stage 0
  // first chunk: issue EVERY piece's loads before converting any (one HBM round trip, not one per
  // piece; the accumulators are not live yet, so the registers are free)
  {
    constexpr int N0 = NU;
    if constexpr (IN16) {
      float4 v0[N0], dz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < N0; ++j) slot_load(j * nthr + tid, 0, v0[j], dz, dz);
      if constexpr (KCS > 1 && NS == 1) {      // the second stage's burst goes out with the first: 2x the bytes in flight
        if (k.nck > KCS) {
#pragma unroll
          for (int j = 0; j < NU; ++j) slot_load(j * nthr + tid, KCS * C16_KC, sv_first[j], dz, dz);
        }
      }
#pragma unroll
      for (int j = 0; j < N0; ++j) slot_store(buf0, j * nthr + tid, -1, v0[j], dz, dz);
    } else {
      float4 v0[N0], sc0[N0], sh0[N0];
      int sp0[N0];
#pragma unroll
      for (int j = 0; j < N0; ++j) sp0[j] = slot_load(j * nthr + tid, 0, v0[j], sc0[j], sh0[j]);
#pragma unroll
      for (int j = 0; j < N0; ++j) slot_store(buf0, j * nthr + tid, sp0[j], v0[j], sc0[j], sh0[j]);
    }
  }
  __syncthreads();
  C16_TSTAMP(2);   // first stage in LDS

  if constexpr (KCS > 1) {
    // ================= staged mode (fp16 source): one burst per stage, single LDS buffer =================
    // A-fragment ring: a fragment must be requested an LDS latency (~130+ cycles under load) before its MFMAs;
    // one M-tile iteration is 32 (F16) / 96 (F16X3) MFMA cycles
#ifdef CSD_C16_SRING
    constexpr int RING = CSD_C16_SRING;
#else
    constexpr int RING = (NS == 1) ? 4 : 3;
#endif
    constexpr int SSTEPS = KCS * TAPS;            // K steps per stage (sub-chunk major, tap minor = stream order)
    const int nstage = k.nck / KCS;
    const char* buf = buf0;
    for (int stg = 0; stg < nstage; ++stg) {
      const bool more = stg + 1 < nstage;
      float4 sv[NU];
      if (more) {                                 // next stage's burst: in flight under this stage's MFMAs
        if (NS == 1 && stg == 0) {                // (stage 1 was requested in the prologue)
#pragma unroll
          for (int j = 0; j < NU; ++j) sv[j] = sv_first[j];
        } else {
          float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef CSD_C16_NOBURST   // tuning aid: no mid-kernel burst (results are garbage) - isolates its cost
#pragma unroll
          for (int j = 0; j < NU; ++j) sv[j] = dz;
#else
#pragma unroll
          for (int j = 0; j < NU; ++j) slot_load(j * nthr + tid, (stg + 1) * KCS * C16_KC, sv[j], dz, dz);
#endif
        }
      }
      half8 areg[RING][NS];
      auto load_frag = [&](int q) {      // q = s * MT + mt, s = sub * TAPS + tap (compile-time after unrolling)
        const int s_ = q / MT, mt_ = q % MT;
        const int sub_ = s_ / TAPS, tap_ = s_ % TAPS;
        const char* p = buf + tap_off(mt_, tap_ / KS, tap_ % KS) + sub_ * 32;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) areg[q % RING][pl] = *reinterpret_cast<const half8*>(p + pl * LO);
      };
#pragma unroll
      for (int q = 0; q < RING - 1; ++q) load_frag(q);
#pragma unroll
      for (int s = 0; s < SSTEPS; ++s) {
        const int bc = s % BR, bn = (s + BR - 1) % BR;
        const int tap = s % TAPS;
        const int r = tap / KS, sx = tap % KS;
        wstep += STEP_BYTES;
#pragma unroll
        for (int p = 0; p < NS; ++p) breg[bn][p] = gload_h8(wstep + p * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int q = s * MT + mt;
          const int ac = q % RING;
          if (q + RING - 1 < SSTEPS * MT) load_frag(q + RING - 1);
          __builtin_amdgcn_sched_barrier(0);
          half8 a[NS];
          const bool v = !MASK || (((vbits[mt] >> r) & 1u) && ((vbits[mt] >> (3 + sx)) & 1u));
#pragma unroll
          for (int pl = 0; pl < NS; ++pl) {
            a[pl] = areg[ac][pl];
            if (MASK && !v) {
#pragma unroll
              for (int e = 0; e < 8; ++e) a[pl][e] = (_Float16)0.f;
            }
          }
          if (NS == 2) {
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], breg[bc][0], acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], breg[bc][1], acc[mt], 0, 0, 0);
          }
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], breg[bc][0], acc[mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      static_assert((KCS * 9) % 3 == 0, "weight ring phase");
      if (stg == 0) C16_TSTAMP(3);                // stage 0 MFMAs done
      if (more) {
        __syncthreads();                          // everyone is done reading this stage's patch
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NU; ++j) slot_store(buf0, j * nthr + tid, -1, sv[j], d0, d0);
        __syncthreads();
      }
    }
  } else {
  for (int ck = 0; ck < k.nck; ++ck) {
    const char* buf = (ck & 1) ? buf1 : buf0;
    char* nbuf = (ck & 1) ? buf0 : buf1;
    const bool more = !(CSD_CONV_ABLATE & 4) && (ck + 1 < k.nck);
    // A-fragment ring, RING-1 fragments ahead of the MFMAs (statically indexed: everything below is
    // unrolled).  One M-tile iteration is 32 (F16) / 96 (F16X3) MFMA cycles against ~120+ cycles of
    // LDS latency, hence the deeper ring in F16 mode.
#ifdef CSD_C16_RING
    constexpr int RING = CSD_C16_RING;
#else
    constexpr int RING = 2;
#endif
    half8 areg[RING][NS];
    auto load_frag = [&](int q) {      // q = st * MT + mt (compile-time after unrolling)
      const int st_ = q / MT, mt_ = q % MT;
      const char* p = buf + tap_off(mt_, st_ / KS, st_ % KS);
#pragma unroll
      for (int pl = 0; pl < NS; ++pl) areg[q % RING][pl] = *reinterpret_cast<const half8*>(p + pl * LO);
    };
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) load_frag(q);
#pragma unroll
    for (int st = 0; st < TAPS; ++st) {
      const int bc = st % BR, bn = (st + BR - 1) % BR;
      const int r = st / KS, s = st % KS;
      // ---- weights of step st+BR-1 (BR-1 steps of slack past the end of the stream) ----
      wstep += STEP_BYTES;
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        if (CSD_CONV_ABLATE & 1) breg[bn][p] = breg[bc][p];
        else breg[bn][p] = gload_h8(wstep + p * 1024);
      }
      // ---- staging pieces of the NEXT chunk ride along: load at even steps, convert+store at odd ----
      if (more && (st & 1) == 0 && (st >> 1) < (TAPS >> 1) && (st >> 1) < npieces) stage_load(ck + 1, st >> 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int q = st * MT + mt;
        const int ac = q % RING;
        // prefetch the fragment RING-1 iterations ahead
        if (q + RING - 1 < TAPS * MT) {
          if (!(CSD_CONV_ABLATE & 2)) load_frag(q + RING - 1);
          else {
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) areg[(q + RING - 1) % RING][pl] = areg[ac][pl];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        half8 a[NS];
        const bool v = !MASK || (((vbits[mt] >> r) & 1u) && ((vbits[mt] >> (3 + s)) & 1u));
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
          a[pl] = areg[ac][pl];
          if (MASK && !(CSD_CONV_ABLATE & 8) && !v) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[pl][e] = (_Float16)0.f;
          }
        }
        if (NS == 2) {
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], breg[bc][0], acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], breg[bc][1], acc[mt], 0, 0, 0);
        }
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], breg[bc][0], acc[mt], 0, 0, 0);
        // odd steps: convert + store one staged slot of the next chunk INSIDE an MFMA block, so its
        // VALU work issues while the matrix pipe is busy
        if ((st & 1) == 1 && (st >> 1) < (TAPS >> 1)) {
#pragma unroll
          for (int j = 0; j < C16_PIECE; ++j)
            if (mt == (j * MT) / C16_PIECE + (MT > 2 ? 1 : 0) && more && (st >> 1) < npieces)
              stage_write_slot(nbuf, ck + 1, st >> 1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more && npieces > (TAPS >> 1)) {     // pieces that did not fit between the steps (large patches)
      for (int piece = TAPS >> 1; piece < npieces; ++piece) {
        stage_load(ck + 1, piece);
        stage_write(nbuf, ck + 1, piece);
      }
    }
    if (!(CSD_CONV_ABLATE & 4)) __syncthreads();
  }

  }

  C16_TSTAMP(4);   // K loop done
  // ---- epilogue: this wave's 32 couts for all MT*32 pixels ----
  // Straight-line and branch-free: every global access goes through a buffer descriptor based at the
  // tile's first output row, and a lane whose pixel / cout does not exist uses an out-of-range offset
  // (the hardware returns 0 for the load and drops the store).  Per-element `if`s here made the compiler
  // put each access in its own basic block behind an s_waitcnt vmcnt(0) - on gfx9 that counter also
  // tracks stores, so the 64 stores of a wave completed one HBM round trip at a time and the epilogue
  // was 60% of the kernel (tools/probes/phase_timing.py).
  const int ohw = k.OH * k.OW;
  const float wunscale = 1.0f / C16_WSCALE;
  const int col = (ng * k.nw + wave) * 32 + (lane & 31);
  const bool cv = col < k.Cout;
  const int colc = cv ? col : 0;
  const float bv = k.a.bias ? k.a.bias[colc] : 0.f;
  constexpr unsigned OOB = 0x80000000u;            // >= num_records of every descriptor below
  constexpr int RSRC_FLAGS = 0x00020000;           // raw dword buffer, gfx9 encoding
  const size_t o_base = (size_t)ov0 * k.OW;        // first pixel of the tile's first row (tall image)
  const int b0 = ov0 / k.OH;                       // first sample the tile touches
  const bool has_res = k.a.res != nullptr, has_temb = k.a.temb != nullptr, nchw = k.a.out_nchw != 0;
  float* const out_base = nchw ? k.a.out + (size_t)b0 * k.Cout * ohw : k.a.out + o_base * k.a.out_stride + k.a.out_coff;
  const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, OOB, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_res ? k.a.res + o_base * k.Cout : k.a.out), 0, OOB, RSRC_FLAGS);
  float tvu = 0.f;                                 // unmasked tiles lie inside one sample
  if (!MASK && has_temb) tvu = k.a.temb[(size_t)b0 * k.a.temb_stride + colc];
  const int pix0 = (int)(o_base - (size_t)b0 * ohw);   // tile origin relative to sample b0 (NCHW store)
#ifndef CSD_C16_GRP
#define CSD_C16_GRP 2
#endif
  constexpr int GRP = CSD_C16_GRP;         // M tiles per batch of epilogue accesses (16 per lane each)
  double st_s = 0.0, st_q = 0.0;           // GroupNorm partials of this lane's column over the tile (fp64, fixed order)
#pragma unroll
  for (int g0 = 0; g0 < MT; g0 += GRP) {
    int oidx[GRP][16];                     // pixel index relative to o_base, or -1
    float addv[GRP][16];
#pragma unroll
    for (int g = 0; g < GRP; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = otab[(g0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];
        oidx[g][r] = cv ? o : -1;
        addv[g][r] = tvu;
      }
    if (has_res) {
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = oidx[g][r];
          const unsigned off = o >= 0 ? (unsigned)(o * k.Cout + col) * 4u : OOB;
          const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, off, 0, 0));
          addv[g][r] = (MASK || !has_temb) ? v : v + tvu;     // (res and temb never meet in the network; order kept anyway)
        }
    }
    if (MASK && has_temb) {
      float tv[GRP][16];
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tv[g][r] = k.a.temb[(size_t)btab[(g0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] * k.a.temb_stride + colc];
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) addv[g][r] += tv[g][r];
    }
    if (nchw) {
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = oidx[g][r];
          const int db = btab[(g0 + g) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] - b0;
          // (acc*2^-8 + bias) + temb + residual: same association as the reference's h + Dense(temb), x + h
          const float val = ((acc[g0 + g][r] * wunscale + bv) + addv[g][r]) * k.a.out_scale;
          const unsigned off = o >= 0 ? (unsigned)((db * k.Cout + col) * ohw + (pix0 + o - db * ohw)) * 4u : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), out_r, off, 0, 0);
        }
    } else {
#pragma unroll
      for (int g = 0; g < GRP; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = oidx[g][r];
          const float val = ((acc[g0 + g][r] * wunscale + bv) + addv[g][r]) * k.a.out_scale;
          const unsigned off = o >= 0 ? (unsigned)(o * k.a.out_stride + col) * 4u : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), out_r, off, 0, 0);
          const double dv = o >= 0 ? (double)val : 0.0;
          st_s += dv;
          st_q = fma(dv, dv, st_q);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // GroupNorm statistics of the tensor being written, for the GroupNorm that consumes it next: one
  // (sum, sum of squares) pair per (tile, cout) - the finalize kernel reduces tiles in a fixed order, so
  // the result is deterministic and the separate statistics pass over the tensor disappears.  The host
  // sets `stats` only when a tile lies inside one sample (OH % TH == 0).
  if (k.a.stats) {
    st_s += __shfl_xor(st_s, 32);
    st_q += __shfl_xor(st_q, 32);
    if (lane < 32 && cv) {
      double* dst = k.a.stats + ((size_t)tile * k.Cout + col) * 2;
      dst[0] = st_s;
      dst[1] = st_q;
    }
  }
  C16_TSTAMP(5);
}

}  // namespace csd
