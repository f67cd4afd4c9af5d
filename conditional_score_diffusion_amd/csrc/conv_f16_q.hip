// conv_f16_q.hip - "quad" schedule of the fp16-source 3x3 stride-1 convolution for layers whose Cout is a
// multiple of 96 (every ResnetBlockDDPM conv of the DDPM-family nets: 96 / 192 / 288 channels; reference
// models/layers.py:ddpm_conv3x3 inside ResnetBlockDDPM, models/ddpm.py:149-213).
//
// Why: the 3-wave workgroups of conv_f16_kernel.h (one 32-cout tile per wave) put 6 MFMA waves on the 4
// SIMDs of a CU as (2,2,1,1).  A workgroup synchronises every K stage, so it runs at the pace of its slowest
// wave and two co-resident workgroups deliver barely more than one (tools/phase_timing*.py: a stage takes
// 7.6k cycles alone, 19k next to a second workgroup; the ideal is 6.9k).  Here a workgroup is FOUR waves in a
// 2 x 2 grid - M half (MQ 16-pixel tiles) x N half (48 couts = three 16-cout tiles) - on
// v_mfma_f32_16x16x32_f16, so two workgroups give every SIMD exactly two waves that cover each other's
// memory stalls.  Operands are swapped (A = weights, B = pixels): a lane's result is 4 consecutive couts of
// one pixel, so the epilogue is 16-byte buffer loads/stores (12 per wave instead of 64 dword ones), all
// reads issued before the first store (vmcnt retires in order and counts stores).
//
// LDS patch layout, staging bursts (KCS = 2: 32 channels per stage = one K step of the MFMA per tap) and
// the arithmetic (NS = 2: hi/lo split operands, 3 MFMAs per product; NS = 1: plain fp16) are those of
// conv_f16_kernel.h.  Weights: [Cout/96][N half][Cin/32][tap][16-cout tile 0..2][plane][lane][8 halves].
#include <stdlib.h>
#include <type_traits>

#include "conv_f16_kernel.h"

namespace csd {

typedef float floatx4q __attribute__((ext_vector_type(4)));
typedef unsigned int uint4q __attribute__((ext_vector_type(4)));

#define C16Q_THREADS 256

// sum of a double over the 16 lanes of a DPP row, result in every lane: quad_perm xor 1, xor 2, then row rotations by 4
// and 8 - VALU-rate data movement (a ds_bpermute butterfly of 24 doubles cost 8.6k cycles per workgroup, 15 % of the
// kernel: tools/probes/phase_timing_q.py)
__device__ __forceinline__ double dpp_row16_sum(double v) {
#define CSD_DPP_STEP(CTRL)                                                                                   \
  {                                                                                                          \
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);                        \
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);                        \
    v += __hiloint2double(hi, lo);                                                                           \
  }
  CSD_DPP_STEP(0xB1)      // quad_perm [1,0,3,2]
  CSD_DPP_STEP(0x4E)      // quad_perm [2,3,0,1]
  CSD_DPP_STEP(0x124)     // row_ror:4
  CSD_DPP_STEP(0x128)     // row_ror:8
#undef CSD_DPP_STEP
  return v;
}
__device__ __forceinline__ float dpp_row16_sum(float v) {
#define CSD_DPP_STEP_F(CTRL) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
  CSD_DPP_STEP_F(0xB1)
  CSD_DPP_STEP_F(0x4E)
  CSD_DPP_STEP_F(0x124)
  CSD_DPP_STEP_F(0x128)
#undef CSD_DPP_STEP_F
  return v;
}

// S: stride.  S = 2 is the reference Downsample (pad (0,1,0,1), models/layers.py:619-625): the patch of a TH x TW output
// tile is (2*TH+1) x (2*TW+1) input pixels and a lane's tap (0,0) sits at (2*ty, 2*tx).
// NTQ: 16-cout tiles per N half - 3 (Cout % 96 == 0: the nf = 96 nets) or 4 (Cout % 128 == 0: the nf = 128 nets)
// F8 (NS = 2 layouts): "fp16 + fp8 corrections" (CSD_PREC_F16F8).  hi*hi stays on v_mfma_f32_16x16x32_f16; the two correction products
// hi_w*lo_x + lo_w*hi_x run K-concatenated on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, block scale 2^-11): ONE instruction
// per cout tile per TWO taps x 32 channels - lanes 0-31 carry tap t (K block = channel half), lanes 32-63 tap t+1.  The second
// plane of the staged pixel holds, per channel, the byte pair (lo * 2^11, hi) in e4m3 (written by gn_apply16_kernel), the second
// plane of a weight step [channel half][cout row][16 x (hi, lo * 2^11)] - byte j of an A lane meets byte j of the B lane.
typedef int int8q __attribute__((ext_vector_type(8)));

// UP4 (S = 1): the reference Upsample = nearest x2 + 3x3 conv (models/layers.py:600-604) as FOUR 2x2 convolutions of the SOURCE image,
// one per output phase (a, b) = (row, column parity): output (2y+a, 2x+b) reads source rows {y-1, y} (a = 0) or {y, y+1} (a = 1) -
// taps that land on the same source pixel are summed when the weights are packed (conv16q_pack_up4_kernel).  4 taps instead of 9 per
// output pixel (2.25x fewer MACs); a workgroup = (source tile, phase, cout group), its outputs are written with stride 2.
// RAW (NS = 2, resampling convs and other layers without a GroupNorm in front): g_hi is the fp32 NHWC SOURCE tensor itself and the staging
// burst splits it into the hi | lo planes of the LDS patch on the way (two float4 in, two 16-byte pieces out per (pixel, 8 channels):
// the same registers as the two plane pieces) - the gn_apply16 pass that wrote the planes (fp32 read + 2 x fp16 written + read again)
// is gone; the arithmetic is that pass's: hi = fp16(x), lo = fp16(x - hi).
template <int MQ, int NS, bool MASK, int PWC, int S, int NTQ, bool F8, bool UP4 = false, bool RAW = false>
__global__ __launch_bounds__(C16Q_THREADS, 2) void conv_f16_q_kernel(const void* __restrict__ g_hi,
                                                                    const void* __restrict__ g_lo,
                                                                    const char* __restrict__ g_wpack,
                                                                    const Conv16KArgs k) {
  static_assert(!UP4 || (S == 1 && !F8), "the phase form is the stride-1 Upsample conv (a resampling conv: full split)");
  static_assert(!RAW || (NS == 2 && !F8), "the fp32-source form splits into two fp16 planes");
  constexpr int TAPS = UP4 ? 4 : 9, KS = UP4 ? 2 : 3, KCS = 2;
  constexpr int LO = 32 * KCS;               // byte offset of the lo plane inside a staged pixel
  constexpr int PSB = 32 * KCS * NS + 16;    // bytes per staged pixel
  constexpr int NPIX = 2 * MQ * 16;          // pixels per workgroup tile
  constexpr int SPP = 2 * NS * KCS;          // 16-byte slots per patch pixel per stage
  constexpr int NU0 = (S == 2) ? ((NS == 2) ? 10 : 5) : ((NS == 2) ? 7 : 4);     // staging slots per thread (host: patch * SPP <= NU0 * 256)
  constexpr int NU = RAW ? 2 * ((NU0 + 1) / 2) : NU0;                            // RAW: slot PAIRS (pixel, 8 channels), two registers quads each
  constexpr int rstride = PWC * PSB;
  constexpr int WSTEP = NTQ * NS * 1024;     // weight bytes per K step per wave
  constexpr int GC = 32 * NTQ, HC = 16 * NTQ; // couts per workgroup / per N half
  extern __shared__ __attribute__((aligned(16))) char smem16[];

  const int tid = threadIdx.x;
#ifdef CSD_C16_TIMING
  int ts_n = 0;
#define Q_TSTAMP() do { if (k.a.dbg && tid == 0 && blockIdx.x < 4095 && ts_n < 8) k.a.dbg[blockIdx.x * 8 + ts_n++] = clock64(); } while (0)
#else
#define Q_TSTAMP() do { } while (0)
#endif
  Q_TSTAMP();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = wave >> 1, ni = wave & 1;
  const int l16 = lane & 15, kq = lane >> 4;

  int w;
  {
    const int bid = blockIdx.x, nb = k.nblocks;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ng = w % k.n_groups;
  const int tile4 = w / k.n_groups;          // UP4: (source tile, phase); else the tile
  const int ph = UP4 ? (tile4 & 3) : 0, pa = ph >> 1, pb = ph & 1;
  const int tile = UP4 ? (tile4 >> 2) : tile4;
  const int tile_y = tile / k.tiles_x;
  const int ov0 = tile_y * k.TH, ox0 = (tile - tile_y * k.tiles_x) * k.TW;
  const int Py = UP4 ? 1 - pa : ((S == 2) ? 0 : 1), Px = UP4 ? 1 - pb : ((S == 2) ? 0 : 1);        // top / left padding
  const int prow0 = ov0 * S - Py, pcol0 = ox0 * S - Px;
  const int EH = (S == 2) ? k.IH : k.OH, EW = (S == 2) ? k.IW : k.OW;     // the image the patch coordinates live in
  const int Cin = k.C0;
  const int npatch = k.PH * k.PW;
  const int patch_bytes = k.PH * PWC * PSB;
  int* const otab = reinterpret_cast<int*>(smem16 + patch_bytes);   // [NPIX] output pixel (relative to the tile row) or -1
  int* const btab = otab + NPIX;                                     // [NPIX] sample
  int* const vtab = btab + NPIX;                                     // [NPIX] tap validity bits
  int* const stab = vtab + NPIX;                                     // [npatch] source pixel or -1
  int* const dtab = stab + npatch;                                   // [npatch] LDS byte offset

  // ---- tables ----
  for (int m = tid; m < NPIX; m += C16Q_THREADS) {
    const int ty = m / k.TW, tx = m - ty * k.TW;
    const int ov = ov0 + ty, ox = ox0 + tx;
    const bool mv = (m < k.TH * k.TW) && (ov < k.B * k.OH) && (ox < k.OW);
    const int b = ov / k.OH, oy = ov - b * k.OH;
    unsigned vb = 0;
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      const int iy = oy * S + r - Py, ix = ox * S + r - Px;
      vb |= ((mv && iy >= 0 && iy < EH) ? 1u : 0u) << r;      // (bounds of the image the conv runs on: upsampled when k.up)
      vb |= ((mv && ix >= 0 && ix < EW) ? 1u : 0u) << (3 + r);
    }
    // (UP4: k.OH / k.OW are the SOURCE size; the output pixel sits at (2 ov + a, 2 ox + b) of the 2 OW wide output)
    otab[m] = mv ? (UP4 ? (ov - ov0) * 4 * k.OW + 2 * ox : (ov - ov0) * k.OW + ox) : -1;
    btab[m] = mv ? b : 0;
    vtab[m] = (int)vb;
  }
  for (int pix = tid; pix < npatch; pix += C16Q_THREADS) {
    const int pr = pix / k.PW, pc = pix - pr * k.PW;
    const int vr = prow0 + pr, col = pcol0 + pc;
    // (MASK == false: the tile lies inside ONE sample; halo rows of its neighbours are padding)
    const int img_lo = MASK ? 0 : (ov0 / k.OH) * EH;
    const int img_hi = MASK ? k.B * EH : img_lo + EH;
    // (k.up: the conv runs on the nearest-x2 upsampled image (models/layers.py:600-604); the patch is in upsampled
    // coordinates and each of its pixels is fetched from source pixel (row >> 1, col >> 1) of the same sample)
    const bool in = vr >= img_lo && vr < img_hi && col >= 0 && col < EW;
    const int sb = vr / EH, sr = vr - sb * EH;
    stab[pix] = in ? (sb * k.IH + (sr >> k.up)) * k.IW + (col >> k.up) : -1;
    dtab[pix] = (pr * PWC + pc) * PSB;
  }
  // per-lane pixels of this wave's M half: LDS offset of tap (0,0) + this lane's 8-channel K slice
  int base[MQ];
#pragma unroll
  for (int j = 0; j < MQ; ++j) {
    const int m = (mi * MQ + j) * 16 + l16;
    const int ty = m / k.TW, tx = m - ty * k.TW;
    base[j] = (m < k.TH * k.TW) ? ty * S * rstride + tx * S * PSB + kq * 16 : kq * 16;
  }
  __syncthreads();
  Q_TSTAMP();
  unsigned vbits[MQ];
  if (MASK) {
#pragma unroll
    for (int j = 0; j < MQ; ++j) vbits[j] = (unsigned)vtab[(mi * MQ + j) * 16 + l16];
  }

  // ---- staging: slot e = pixel * SPP + plane * 2*KCS + 16-byte piece ----
  const int total4 = RAW ? npatch * 2 * KCS : npatch * SPP;
  // RAW: pair e = pixel * 2 KCS + 8-channel group; half = 0 | 1: the group's first | second float4
  auto raw_load = [&](int e, int cb, int half) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total4) {
      const int pix = e / (2 * KCS), grp = e - pix * (2 * KCS);
      const int sp = stab[pix];
      if (sp >= 0) v = gload4f(static_cast<const float*>(g_hi) + (size_t)sp * Cin + cb + grp * 8 + half * 4);
    }
    return v;
  };
  auto raw_store = [&](int e, const float4& a, const float4& b) {
    if (e >= total4) return;
    const int pix = e / (2 * KCS), grp = e - pix * (2 * KCS);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = (_Float16)v[j];
      l[j] = (_Float16)(v[j] - (float)h[j]);
    }
    *reinterpret_cast<half8*>(smem16 + dtab[pix] + grp * 16) = h;
    *reinterpret_cast<half8*>(smem16 + dtab[pix] + LO + grp * 16) = l;
  };
  auto slot_load = [&](int e, int cb) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < total4) {
      const int pix = e / SPP, sub = e - pix * SPP;
      const int sp = stab[pix];
      if (sp >= 0) {
        const int pl = sub / (2 * KCS), rest = sub - pl * (2 * KCS);
        const _Float16* plane = static_cast<const _Float16*>((NS == 2 && pl) ? g_lo : g_hi);
        v = gload4f(reinterpret_cast<const float*>(plane + (size_t)sp * Cin + cb + rest * 8));
      }
    }
    return v;
  };
  auto slot_store = [&](int e, const float4& v) {
    if (e >= total4) return;
    const int pix = e / SPP, sub = e - pix * SPP;
    const int pl = sub / (2 * KCS), rest = sub - pl * (2 * KCS);
    *reinterpret_cast<float4*>(smem16 + dtab[pix] + pl * LO + rest * 16) = v;
  };

  // ---- weight stream of this wave: 3 fragments (x NS planes) per K step, ring of BR steps ----
  // ring depth: the prefetch distance (BR - 1 steps of MQ x NTQ x 3 MFMAs) has to cover an L2 round trip.  With one 16-cout tile per
  // wave (the 5^2 level) a step is 6 MFMAs = ~100 cycles, so the whole stage's nine steps are kept in flight there (72 registers):
  // 26 -> 23.5 us per launch for ~7 us of arithmetic - the rest of those launches is the per-stage patch hand-over (a 32-channel
  // stage is 0.4 us of MFMAs: one stage of prefetch distance does not cover a global load), launch latency and the prologue tables
  constexpr int BR = UP4 ? 2 : (NTQ == 1 && !F8 ? 9 : 3);
  const int nk32 = Cin / 32;
  const char* wstep = g_wpack + ((size_t)((UP4 ? ph * k.n_groups * 2 : 0) + ng * 2 + ni) * nk32 * TAPS) * WSTEP + lane * 16;
  constexpr int NP = F8 ? 1 : NS;               // fp16 planes read per fragment
  half8 wreg[BR][NTQ][NP];
#pragma unroll
  for (int q = 0; q < BR - 1; ++q)
#pragma unroll
    for (int t = 0; t < NTQ; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p) wreg[q][t][p] = gload_h8(wstep + (size_t)q * WSTEP + (t * NS + p) * 1024);
  wstep += (size_t)(BR - 2) * WSTEP;
  // F8: correction operands of a tap pair, one pair ahead: lane (kq >> 1) selects the tap, (kq & 1) the channel half
  const char* const w8base = g_wpack + ((size_t)(ng * 2 + ni) * nk32 * TAPS + (kq >> 1)) * WSTEP + (lane & 31) * 32 + 1024;      // (never with UP4)
  int8q w8[1][NTQ];                              // (single buffer: refilled right after a pair's MFMAs, two taps before its next use)
  auto load_w8 = [&](int buf, int step) {       // step = global K step (stage * 9 + tap) of the pair's first tap
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
      const uint4q a = *reinterpret_cast<const uint4q*>(w8base + (size_t)step * WSTEP + t * NS * 1024);
      const uint4q c = *reinterpret_cast<const uint4q*>(w8base + (size_t)step * WSTEP + t * NS * 1024 + 16);
      w8[buf][t] = int8q{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)c.x, (int)c.y, (int)c.z, (int)c.w};
    }
  };
  if constexpr (F8) load_w8(0, 0);
  int base8[MQ];                                 // the lane's pixel record + second plane + channel half (tap added per pair)
#pragma unroll
  for (int j = 0; j < MQ; ++j) base8[j] = base[j] - kq * 16 + LO + (kq & 1) * 32;

  floatx4q acc[MQ][NTQ];
#pragma unroll
  for (int j = 0; j < MQ; ++j)
#pragma unroll
    for (int t = 0; t < NTQ; ++t) acc[j][t] = floatx4q{0.f, 0.f, 0.f, 0.f};

  // first stage: one burst
  {
    float4 v0[NU];
    if constexpr (RAW) {
#pragma unroll
      for (int j = 0; j < NU; ++j) v0[j] = raw_load((j >> 1) * C16Q_THREADS + tid, 0, j & 1);
#pragma unroll
      for (int j = 0; j < NU; j += 2) raw_store((j >> 1) * C16Q_THREADS + tid, v0[j], v0[j + 1]);
    } else {
#pragma unroll
      for (int j = 0; j < NU; ++j) v0[j] = slot_load(j * C16Q_THREADS + tid, 0);
#pragma unroll
      for (int j = 0; j < NU; ++j) slot_store(j * C16Q_THREADS + tid, v0[j]);
    }
  }
  __syncthreads();
  Q_TSTAMP();

  const int nstage = nk32;                      // one 32-channel K step per tap per stage
  for (int stg = 0; stg < nstage; ++stg) {
    const bool more = stg + 1 < nstage;
    float4 sv[NU];
    if (more) {                                 // next stage's burst: in flight under this stage's MFMAs
#pragma unroll
      for (int j = 0; j < NU; ++j)
        sv[j] = RAW ? raw_load((j >> 1) * C16Q_THREADS + tid, (stg + 1) * 32, j & 1) : slot_load(j * C16Q_THREADS + tid, (stg + 1) * 32);
    }
#ifdef CSD_Q_RING
    constexpr int RING = CSD_Q_RING;
#else
    constexpr int RING = 2;
#endif
    half8 preg[RING][NP];
    auto load_frag = [&](int q) {               // q = tap * MQ + j
      const int tap_ = q / MQ, j_ = q % MQ;
      const char* p = smem16 + base[j_] + (tap_ / KS) * rstride + (tap_ % KS) * PSB;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) preg[q % RING][pl] = *reinterpret_cast<const half8*>(p + pl * LO);
    };
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) load_frag(q);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int bc = tap % BR, bn = (tap + BR - 1) % BR;
      const int r = tap / KS, sx = tap % KS;
      wstep += WSTEP;
      // The weight stream through L1 / the texture addresser is the binding resource of this loop: a tuning build with a third
      // of these loads (CSD_Q_ABLATE=1, wrong results) runs the 3x3 class 20 % faster (36.3 -> 28.9 ms per PC step).  Each
      // fragment is pulled by BOTH waves of an N half.  Tried: 256-pixel tiles (MQ = 8) halve the bytes per MFMA but need 316
      // registers in the split mode (60 spilled: 66.4 vs 62.1 ms per step); sharing through an LDS ring needs a pair barrier per
      // tap or 36 KB more LDS per workgroup.
#ifndef CSD_Q_ABLATE
#define CSD_Q_ABLATE 0
#endif
#pragma unroll
      for (int t = 0; t < ((CSD_Q_ABLATE & 1) ? 1 : NTQ); ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) wreg[bn][t][p] = gload_h8(wstep + (t * NS + p) * 1024);

#pragma unroll
      for (int j = 0; j < MQ; ++j) {
        const int q = tap * MQ + j;
        if (q + RING - 1 < TAPS * MQ) load_frag(q + RING - 1);
        __builtin_amdgcn_sched_barrier(0);
        half8 b[NP];
        const bool v = !MASK || (((vbits[j] >> r) & 1u) && ((vbits[j] >> (3 + sx)) & 1u));
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          b[pl] = preg[q % RING][pl];
          if (MASK && !v) {
#pragma unroll
            for (int e = 0; e < 8; ++e) b[pl][e] = (_Float16)0.f;
          }
        }
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
          if constexpr (NS == 2 && !F8) {
            acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[bc][t][1], b[0], acc[j][t], 0, 0, 0);
            acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[bc][t][0], b[1], acc[j][t], 0, 0, 0);
          }
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[bc][t][0], b[0], acc[j][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (F8) {
        if (tap % 2 == 1 || tap == TAPS - 1) {     // the pair (t0, t0 + 1) is complete (or the last tap stands alone)
          const int t0 = tap % 2 == 1 ? tap - 1 : tap;
          const bool pair = t0 + 1 < TAPS;
          const int t1 = pair ? t0 + 1 : t0;
          const int o0 = (t0 / KS) * rstride + (t0 % KS) * PSB, o1 = (t1 / KS) * rstride + (t1 % KS) * PSB;
          const int off = (kq >> 1) ? o1 : o0;
          constexpr int wb = 0;
#pragma unroll
          for (int j = 0; j < MQ; ++j) {
            const char* xp = smem16 + base8[j] + off;
            const uint4q a = *reinterpret_cast<const uint4q*>(xp);
            const uint4q c = *reinterpret_cast<const uint4q*>(xp + 16);
            bool ok = pair || (kq >> 1) == 0;        // the upper K half of the unpaired tap multiplies zeros
            if (MASK) {
              const int tt = (kq >> 1) ? t1 : t0;
              ok = ok && ((vbits[j] >> (tt / KS)) & 1u) && ((vbits[j] >> (3 + tt % KS)) & 1u);
            }
            const int8q b8 = int8q{ok ? (int)a.x : 0, ok ? (int)a.y : 0, ok ? (int)a.z : 0, ok ? (int)a.w : 0,
                                   ok ? (int)c.x : 0, ok ? (int)c.y : 0, ok ? (int)c.z : 0, ok ? (int)c.w : 0};
#pragma unroll
            for (int t = 0; t < NTQ; ++t)
              acc[j][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8[wb][t], b8, acc[j][t], 0, 0, 0, 116, 0, 127);
          }
          // the next pair's correction weights (the last pair of a stage fetches the next stage's first): two taps of MFMAs to arrive
          const int nxt = t0 + 2 < TAPS ? stg * TAPS + t0 + 2 : (stg + 1) * TAPS;
          if (t0 + 2 < TAPS || more) load_w8(0, nxt);
        }
      }
    }
    static_assert(TAPS % BR == 0, "weight ring phase");
    Q_TSTAMP();
    if (more) {
      __syncthreads();                          // everyone is done reading this stage's patch
      if constexpr (RAW) {
#pragma unroll
        for (int j = 0; j < NU; j += 2) raw_store((j >> 1) * C16Q_THREADS + tid, sv[j], sv[j + 1]);
      } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) slot_store(j * C16Q_THREADS + tid, sv[j]);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: lane = pixel l16 of each 16-pixel tile, 4 consecutive couts per 16-cout tile ----
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  const float wunscale = 1.0f / C16_WSCALE;
  const size_t o_base = UP4 ? ((size_t)(2 * ov0 + pa) * 2 * k.OW + pb) : (size_t)ov0 * k.OW;
  const int b0 = ov0 / k.OH;
  const bool has_res = k.a.res != nullptr, has_temb = k.a.temb != nullptr;
  const __amdgpu_buffer_rsrc_t out_r =
      __builtin_amdgcn_make_buffer_rsrc(k.a.out + o_base * k.a.out_stride + k.a.out_coff, 0, OOB, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_res ? k.a.res + o_base * k.Cout : k.a.out), 0, OOB, RSRC_FLAGS);
  const int c_base = ng * GC + ni * HC + kq * 4;            // this lane's first cout of 16-cout tile 0
  int oidx[MQ];
#pragma unroll
  for (int j = 0; j < MQ; ++j) oidx[j] = otab[(mi * MQ + j) * 16 + l16];
  // every read before the first store
  float4 bias4[NTQ], tvu4[NTQ];
#pragma unroll
  for (int t = 0; t < NTQ; ++t) {
    const int c0 = c_base + t * 16;
    bias4[t] = k.a.bias ? gload4f(k.a.bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    tvu4[t] = (!MASK && has_temb) ? gload4f(k.a.temb + (size_t)b0 * k.a.temb_stride + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 addv[MQ][NTQ];
#pragma unroll
  for (int j = 0; j < MQ; ++j)
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
      float4 a = tvu4[t];
      if (has_res) {
        const unsigned off = oidx[j] >= 0 ? (unsigned)(oidx[j] * k.Cout + c_base + t * 16) * 4u : OOB;
        const uint4q rv = __builtin_amdgcn_raw_buffer_load_b128(res_r, off, 0, 0);
        const float4 rr = make_float4(__uint_as_float(rv.x), __uint_as_float(rv.y), __uint_as_float(rv.z), __uint_as_float(rv.w));
        // residual first, then the time-embedding column: the order of conv_f16_kernel.h
        a = (MASK || !has_temb) ? rr : make_float4(rr.x + a.x, rr.y + a.y, rr.z + a.z, rr.w + a.w);
      }
      addv[j][t] = a;
    }
  if (MASK && has_temb) {
#pragma unroll
    for (int j = 0; j < MQ; ++j) {
      const int b = btab[(mi * MQ + j) * 16 + l16];
#pragma unroll
      for (int t = 0; t < NTQ; ++t) {
        const float4 tv = gload4f(k.a.temb + (size_t)b * k.a.temb_stride + c_base + t * 16);
        addv[j][t] = make_float4(addv[j][t].x + tv.x, addv[j][t].y + tv.y, addv[j][t].z + tv.z, addv[j][t].w + tv.w);
      }
    }
  }
  // GroupNorm partials of this lane's 12 columns over its MQ pixels.  The phase-decomposed Upsample (UP4) keeps them in fp32: a partial
  // covers 16 MQ pixels and is widened to fp64 where it is stored - the finalize kernel folds the (tile, phase) partials in fp64 as
  // ever.  (In fp64 the statistics cost that kernel more than the streaming pass over its output they replace: round 5.)
  using st_t = typename std::conditional<UP4, float, double>::type;
  st_t st_s[NTQ][4], st_q[NTQ][4];
#pragma unroll
  for (int t = 0; t < NTQ; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) { st_s[t][i] = 0; st_q[t][i] = 0; }
#pragma unroll
  for (int j = 0; j < MQ; ++j)
#pragma unroll
    for (int t = 0; t < NTQ; ++t) {
      const float b4[4] = {bias4[t].x, bias4[t].y, bias4[t].z, bias4[t].w};
      const float a4[4] = {addv[j][t].x, addv[j][t].y, addv[j][t].z, addv[j][t].w};
      float o4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (acc*2^-8 + bias) + temb + residual: same association as the reference's h + Dense(temb), x + h
        o4[i] = ((acc[j][t][i] * wunscale + b4[i]) + a4[i]) * k.a.out_scale;
        const st_t dv = oidx[j] >= 0 ? (st_t)o4[i] : (st_t)0;
        st_s[t][i] += dv;
        st_q[t][i] = fma(dv, dv, st_q[t][i]);
      }
      uint4q ov;
      ov.x = __float_as_uint(o4[0]); ov.y = __float_as_uint(o4[1]); ov.z = __float_as_uint(o4[2]); ov.w = __float_as_uint(o4[3]);
      const unsigned off = oidx[j] >= 0 ? (unsigned)(oidx[j] * k.a.out_stride + c_base + t * 16) * 4u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(ov, out_r, off, 0, 0);
    }
  // GroupNorm statistics of the written tensor: one (sum, sumsq) pair per (tile, M half, cout); the host sets
  // `stats` only when a tile lies inside one sample.  Fixed-order DPP reduction over the 16 pixel lanes.
  if (k.a.stats) {
#pragma unroll
    for (int t = 0; t < NTQ; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double s = (double)dpp_row16_sum(st_s[t][i]), q = (double)dpp_row16_sum(st_q[t][i]);
        if (l16 == 0)      // (one 16-byte store per pair)
          *reinterpret_cast<double2*>(k.a.stats + (((size_t)tile4 * 2 + mi) * k.Cout + c_base + t * 16 + i) * 2) = make_double2(s, q);
      }
  }
  Q_TSTAMP();
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool conv16q_supported(const ConvPlan& p, int ns) {
  return (ns >= 1 && ns <= 3) && p.taps == 9 && ((p.stride == 1 && (p.up == 0 || p.up == 1)) || (p.stride == 2 && p.up == 0)) &&
         p.C1 == 0 && p.C0 % 32 == 0 &&
         (p.qnt == 1 ? (p.Cout % 32 == 0 && ns >= 2 && p.stride == 1 && p.up == 0) : (p.Cout % 96 == 0 || (p.Cout % 128 == 0 && ns != 3)));
}

// the phase-decomposed Upsample layers whose tile shape does not depend on the batch (conv16q_plan_tiles): their epilogue statistics
// are a function of the sample alone
bool conv16q_up4_stats_ok(const ConvPlan& p) { return p.up == 2 && (long)p.IH * p.IW >= 1600 && p.IH % p.TH == 0; }

// ConvPlan.up == 2 selects the phase-decomposed Upsample (UP4): same source / output sizes as up == 1
bool conv16q_up4_supported(const ConvPlan& p, int ns) {
  ConvPlan q = p;
  q.up = 1;
  return (ns == 1 || ns == 2) && p.stride == 1 && conv16q_supported(q, ns) && !CSD_TUNE_ENV("CSD_NO_UP4");
}

size_t conv16q_up4_packed_bytes(const ConvPlan& p, int ns) {
  const int ntq = p.Cout % 96 == 0 ? 3 : 4;
  return (size_t)4 * (p.Cout / (32 * ntq)) * 2 * (p.C0 / 32) * 4 * ntq * ns * 1024 + (size_t)4 * ntq * ns * 1024;   // 4 phases x 4 taps
}

// phase weights: W_ab[i][j] = sum of the 3x3 taps (r, s) whose upsampled source row / column is i / j of the phase's 2x2 window:
// a = 0: i = 0 <- {r = 0}, i = 1 <- {1, 2};  a = 1: i = 0 <- {0, 1}, i = 1 <- {2}  (columns alike); summed in fp32, then split.
// layout [phase][cout group][N half][cin / 32][tap 0..3][16-cout tile][plane][lane][8 halves]
__global__ void conv16q_pack_up4_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int Cin, int Cout, int ns, int ntq) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Cout * Cin * 16;
  {
    const size_t body32 = (size_t)4 * (Cout / (32 * ntq)) * 2 * (Cin / 32) * 4 * ntq * ns * 256, slack32 = (size_t)ntq * ns * 1024;
    if (idx < slack32) reinterpret_cast<uint32_t*>(wpack)[body32 + idx] = 0u;
  }
  if (idx >= total) return;
  const int pt = (int)(idx % 16), ph = pt >> 2, tap = pt & 3;
  const int cin = (int)((idx / 16) % Cin);
  const int cout = (int)(idx / ((size_t)16 * Cin));
  const int a = ph >> 1, b = ph & 1, i = tap >> 1, j = tap & 1;
  const int r0 = a == 0 ? (i == 0 ? 0 : 1) : (i == 0 ? 0 : 2), r1 = a == 0 ? (i == 0 ? 0 : 2) : (i == 0 ? 1 : 2);
  const int s0 = b == 0 ? (j == 0 ? 0 : 1) : (j == 0 ? 0 : 2), s1 = b == 0 ? (j == 0 ? 0 : 2) : (j == 0 ? 1 : 2);
  float v = 0.f;
  for (int r = r0; r <= r1; ++r)
    for (int sx = s0; sx <= s1; ++sx) v += w[((size_t)cout * Cin + cin) * 9 + r * 3 + sx];
  v *= C16_WSCALE;
  const int gc = 32 * ntq, hc = 16 * ntq;
  const int ng = cout / gc, ni = (cout % gc) / hc, t = (cout % hc) / 16, rr = cout % 16;
  const int kb = cin / 32, kq = (cin % 32) / 8, q = cin % 8;
  const int lane = kq * 16 + rr;
  const size_t step = (((size_t)ph * (Cout / gc) * 2 + ng * 2 + ni) * (Cin / 32) + kb) * 4 + tap;
  _Float16* dst = wpack + (step * ntq + t) * (size_t)ns * 512 + lane * 8 + q;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  if (ns == 2) dst[512] = (_Float16)(v - (float)hi);
}

int conv16q_pack_weight_up4(const ConvPlan& p, int ns, const float* w, void* wpack, hipStream_t s) {
  const size_t total = (size_t)p.Cout * p.C0 * 16;
  hipLaunchKernelGGL(conv16q_pack_up4_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, (_Float16*)wpack, p.C0, p.Cout, ns,
                     p.Cout % 96 == 0 ? 3 : 4);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// 16-cout tiles per N half: groups of 96 couts where that divides (the layout the nf = 96 nets have always had), else of 128
// (ConvPlan.qnt = 1: groups of 32 couts - the 5 x 5 level, where 96-cout groups leave 81 workgroups for 256 CUs and the weight
// stream of a layer goes through 81 texture paths; 243 workgroups of a third of the weights each run the level's convs 2x faster)
static inline int q_ntq(const ConvPlan& p) { return p.qnt ? p.qnt : (p.Cout % 96 == 0 ? 3 : 4); }

// ns = 3 (fp16 + fp8 corrections): two planes like ns = 2, the second one holds e4m3 byte pairs
size_t conv16q_packed_bytes(const ConvPlan& p, int ns) {
  if (ns == 3) ns = 2;
  const int ntq = q_ntq(p);
  return (size_t)(p.Cout / (32 * ntq)) * 2 * (p.C0 / 32) * 9 * ntq * ns * 1024 + (size_t)8 * ntq * ns * 1024;   // + prefetch slack (8 K steps: the deepest weight ring)
}

__global__ void conv16q_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int layout, int cin_src,
                                    int cout_src, int cout_off, int Cin, int Cout, int ns, int ntq, int f8) {
  // one thread per (cout in [cout_off, cout_off + cout_src), cin, tap)
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)cout_src * Cin * 9;
  if (idx >= total) return;
  {                                                    // the prefetch slack behind the last fragment (read, never multiplied) is zeroed here
    const size_t body32 = (size_t)(Cout / (32 * ntq)) * 2 * (Cin / 32) * 9 * ntq * ns * 256, slack32 = (size_t)2 * ntq * ns * 1024;
    if (cout_off == 0 && idx < slack32) reinterpret_cast<uint32_t*>(wpack)[body32 + idx] = 0u;
  }
  const int tap = (int)(idx % 9);
  const int cin = (int)((idx / 9) % Cin);
  const int co = (int)(idx / ((size_t)9 * Cin));
  const int cout = cout_off + co;
  if (cin >= cin_src || cout >= Cout) return;          // padding stays zero
  // layout 0: OIHW; 1: NIN [Cin][Cout]; 2: the transposed convolution's OIHW weight [Cin][Cout][9], spatially flipped (data gradient)
  const float v = ((layout == 0) ? w[((size_t)co * cin_src + cin) * 9 + tap]
                   : (layout == 1) ? w[(size_t)cin * cout_src + co]
                                   : w[((size_t)cin * cout_src + co) * 9 + (8 - tap)]) * C16_WSCALE;
  const int gc = 32 * ntq, hc = 16 * ntq;
  const int ng = cout / gc, ni = (cout % gc) / hc, t = (cout % hc) / 16, r = cout % 16;
  const int kb = cin / 32, kq = (cin % 32) / 8, q = cin % 8;
  const int lane = kq * 16 + r;
  const size_t step = ((size_t)(ng * 2 + ni) * (Cin / 32) + kb) * 9 + tap;
  _Float16* dst = wpack + (step * ntq + t) * (size_t)ns * 512 + lane * 8 + q;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  if (ns == 2 && !f8) dst[512] = (_Float16)(v - (float)hi);
  if (f8) {                                            // second plane: [channel half][cout row][16 x (hi, lo * 2^11)] in e4m3
    const float hf = fminf(fmaxf((float)hi, -448.f), 448.f), lf = fminf(fmaxf((v - (float)hi) * 2048.f, -448.f), 448.f);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(hf, lf, 0, false);
    unsigned char* c8 = reinterpret_cast<unsigned char*>(wpack + ((step * ntq + t) * (size_t)ns + 1) * 512) +
                        (((cin % 32) / 16) * 16 + r) * 32 + (cin % 16) * 2;
    c8[0] = (unsigned char)(pk & 255);
    c8[1] = (unsigned char)((pk >> 8) & 255);
  }
}

__global__ void conv16q_zero_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

int conv16q_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                        void* wpack, hipStream_t s) {
  const int f8 = ns == 3;
  if (f8) ns = 2;
  if (cout_off == 0 && (cin_src != p.C0 || cout_src != p.Cout)) {     // (a weight that fills the packed layout needs no zero fill)
    const size_t n32 = conv16q_packed_bytes(p, ns) / 4;
    hipLaunchKernelGGL(conv16q_zero_kernel, dim3((unsigned)cdiv64(n32, 256)), dim3(256), 0, s, (uint32_t*)wpack, n32);
    CSD_LAUNCH_CHECK();
  }
  const size_t total = (size_t)cout_src * p.C0 * 9;
  hipLaunchKernelGGL(conv16q_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, (_Float16*)wpack, layout,
                     cin_src, cout_src, cout_off, p.C0, p.Cout, ns, q_ntq(p), f8);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

static int in_coord_q(int t, int stride = 1) { return stride * (t - 1) + 3; }      // patch extent of a t-pixel tile edge (3x3)
static int in_coord_k(int t, int stride, int ks) { return stride * (t - 1) + ks; }

int conv16q_plan_tiles(ConvPlan* p, int ns) {
  const bool up4 = p->up == 2;
  CSD_REQUIRE(up4 ? conv16q_up4_supported(*p, ns) : conv16q_supported(*p, ns), "conv16q: unsupported shape (Cin=%d Cout=%d)", p->C0, p->Cout);
  // (UP4 tiles the SOURCE image: one workgroup per (source tile, phase, cout group), 2x2 taps)
  const int OHt = up4 ? p->IH : p->OH, OWt = up4 ? p->IW : p->OW, ks = up4 ? 2 : 3;
  p->KC = C16_KC;
  p->CoutPad = p->Cout;
  p->NT = q_ntq(*p);
  p->n_groups = p->Cout / (32 * p->NT);
  p->KCS = 2;
  p->LC = 0;
  if (ns == 3) ns = 2;                          // (same staged-pixel geometry)
  const int psb = 32 * 2 * ns + 16;
  const int spp = 2 * ns * 2;
  const int st = p->stride;
  const int nu = st == 2 ? (ns == 2 ? 10 : 5) : (ns == 2 ? 7 : 4);
  auto pitch = [&](int tw) { return in_coord_k(tw, st, ks) <= 24 ? 24 : 34; };
  int best_mq = 0, best_tw = 0, best_th = 0;
  for (int mq = p->qnt == 1 ? 2 : 4; mq >= 2; mq >>= 1) {        // 128- or 64-pixel tiles (32-cout groups: 64-pixel tiles only)
    const int npix = 2 * mq * 16;
    int btw = 0, bth = 0, blds = 1 << 30;
    double bcov = -1;
    bool bunm = false;
    for (int tw = 1; tw <= 32 && tw <= OWt; ++tw) {
      if (OWt % tw != 0 && !(tw == 32 && OWt > 32)) continue;
      if (in_coord_k(tw, st, ks) > 34) continue;
      // the tallest tile of this width that fits LDS and the staging slots (th * tw may be < npix: spare pixels are masked)
      for (int th = npix / tw; th >= 1; --th) {
        const int patch = in_coord_k(th, st, ks) * in_coord_k(tw, st, ks);
        const int lds = in_coord_k(th, st, ks) * pitch(tw) * psb;
        if (lds > 72 * 1024 || patch * spp > nu * C16Q_THREADS) continue;
        const double cov = (double)th * tw * OWt / ((double)cdiv(OWt, tw) * tw);
        const bool unm = (OHt % th) == 0;
        if (cov > bcov + 1e-9 || (cov > bcov - 1e-9 && ((unm && !bunm) || (unm == bunm && lds < blds)))) {
          bcov = cov; blds = lds; btw = tw; bth = th; bunm = unm;
        }
        break;
      }
    }
    if (btw == 0) continue;
    if (mq > 2 && bcov < 0.75 * npix) continue;        // a large tile that would be mostly masked: try the smaller one
    best_mq = mq; best_tw = btw; best_th = bth;
    const long nwg = (long)cdiv(OWt, btw) * cdiv(p->B * OHt, bth) * p->n_groups * (up4 ? 4 : 1);
    // (the weight fragments come through the CU's texture path once per WAVE: what a layer costs is the weight traffic of its busiest
    // CU.  128-pixel tiles halve that traffic as soon as they still put a workgroup on every CU: measured at the 20^2 level (B = 64:
    // 400 instead of 800 workgroups) -0.6 ms per PC step; at 10^2 (162 workgroups) the 64-pixel tiles stay ahead.)
    static const long min_wg = CSD_TUNE_ENV("CSD_Q_MINWG") ? atol(CSD_TUNE_ENV("CSD_Q_MINWG")) : 300;
    // (a phase-decomposed Upsample of a source of >= 40^2 pixels keeps the tile it gets at a large batch WHATEVER the batch: its epilogue
    // leaves the next GroupNorm's statistics per (tile, M half, phase) - unet.hip reads up_stats_ok() - and a sample's bits must not
    // depend on the batch it is run in.  At B = 1 such a layer then puts 120 - 200 workgroups on the chip instead of 240 - 400.)
    if (up4 && (long)OHt * OWt >= 1600) break;
    if (nwg >= min_wg || mq == 2) break;
  }
  CSD_REQUIRE(best_tw > 0, "conv16q: no feasible tile for OW=%d", p->OW);
  p->TW = best_tw; p->TH = best_th;
  p->PH = in_coord_k(p->TH, st, ks); p->PW = in_coord_k(p->TW, st, ks);
  p->tiles_x = cdiv(OWt, p->TW);
  p->tiles_y = cdiv(p->B * OHt, p->TH);
  p->MT = best_mq;           // (field reused: 16-pixel tiles per wave)
  p->lds_bytes = (size_t)p->PH * pitch(p->TW) * psb + (size_t)3 * 2 * best_mq * 16 * sizeof(int) +
                 (size_t)2 * p->PH * p->PW * sizeof(int);
  return CSD_OK;
}

template <int MQ, int NS, bool MASK, int PWC, int S, int NTQ, bool F8 = false, bool UP4 = false, bool RAW = false>
static int launch_q(const Conv16KArgs& k, size_t lds, hipStream_t s) {
  auto kern = conv_f16_q_kernel<MQ, NS, MASK, PWC, S, NTQ, F8, UP4, RAW>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(k.nblocks), dim3(C16Q_THREADS), lds, s, static_cast<const void*>(k.a.src0),
                     static_cast<const void*>(k.a.src1), reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// the fp32-source (RAW) forms exist for the full split with three or four cout tiles per N half
template <int MQ, int NS, bool MASK, int PWC, int S, int NTQ, bool UP4>
static int launch_q_src(bool raw, const Conv16KArgs& k, size_t lds, hipStream_t s) {
  if constexpr (NS == 2 && NTQ >= 3) {
    if (raw) return launch_q<MQ, NS, MASK, PWC, S, NTQ, false, UP4, true>(k, lds, s);
  }
  if (raw) {
    set_error("conv16q: no fp32-source kernel for ns=%d NT=%d", NS, NTQ);
    return CSD_ERR_INVALID;
  }
  return launch_q<MQ, NS, MASK, PWC, S, NTQ, false, UP4, false>(k, lds, s);
}

int conv16q_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s, bool raw) {
  const bool up4 = p.up == 2;
  CSD_REQUIRE(up4 ? conv16q_up4_supported(p, ns) : conv16q_supported(p, ns), "conv16q: unsupported layer");
  CSD_REQUIRE(!a.out_nchw && a.nscale == nullptr && a.out_stride % 4 == 0 && a.out_coff % 4 == 0,
              "conv16q: NHWC fp32 output with 16-byte aligned rows, pre-normalised fp16 source");
  Conv16KArgs k;
  k.a = a;
  k.a.dbg = nullptr;
#ifdef CSD_C16_TIMING
  {
    extern long long* g_c16_dbg; extern int g_c16_dbg_blocks, g_c16_dbg_max, g_c16_dbg_n;
    const int nb = cdiv(p.OW, p.TW) * cdiv(p.B * p.OH, p.TH) * p.n_groups;
    if (g_c16_dbg && nb == g_c16_dbg_blocks && g_c16_dbg_n < g_c16_dbg_max) {
      k.a.dbg = g_c16_dbg + (size_t)g_c16_dbg_n * 4096 * 8;
      if (hipMemsetAsync(k.a.dbg, 0, 4096 * 8 * 8, s) != hipSuccess) return -1;
      long long hdr[4] = {p.C0, p.Cout, 1, (a.res ? 1 : 0) | (a.temb ? 2 : 0)};
      if (hipMemcpyAsync(k.a.dbg + 4095 * 8, hdr, sizeof(hdr), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
      g_c16_dbg_n++;
    }
  }
#endif
  k.B = p.B; k.IH = p.IH; k.IW = p.IW; k.OH = p.OH; k.OW = p.OW;
  k.C0 = p.C0; k.C1 = 0; k.Cout = p.Cout;
  k.stride = p.stride; k.pad = p.stride == 2 ? 0 : 1; k.up = p.up;
  k.TH = p.TH; k.TW = p.TW; k.PH = p.PH; k.PW = p.PW;
  k.tiles_x = p.tiles_x; k.n_groups = p.n_groups;
  k.nblocks = p.tiles_x * p.tiles_y * p.n_groups * (up4 ? 4 : 1);
  if (up4) { k.OH = p.IH; k.OW = p.IW; k.up = 0; }      // the kernel tiles the source image
  k.nck = p.C0 / C16_KC;
  k.nw = 4;
  k.ntiles_n = p.Cout / 32;
  const bool mask = ((up4 ? p.IH : p.OH) % p.TH) != 0;
#define CSD_QU_CASE(MQ_, NS_, NTQ_)                                                              \
  if (up4 && p.MT == MQ_ && ns == NS_ && p.NT == NTQ_) {                                        \
    if (p.PW <= 24) {                                                                            \
      if (mask) return launch_q_src<MQ_, NS_, true, 24, 1, NTQ_, true>(raw, k, p.lds_bytes, s);    \
      return launch_q_src<MQ_, NS_, false, 24, 1, NTQ_, true>(raw, k, p.lds_bytes, s);             \
    }                                                                                            \
    if (mask) return launch_q_src<MQ_, NS_, true, 34, 1, NTQ_, true>(raw, k, p.lds_bytes, s);      \
    return launch_q_src<MQ_, NS_, false, 34, 1, NTQ_, true>(raw, k, p.lds_bytes, s);               \
  }
  CSD_QU_CASE(4, 2, 3) CSD_QU_CASE(2, 2, 3) CSD_QU_CASE(4, 1, 3) CSD_QU_CASE(2, 1, 3)
  CSD_QU_CASE(4, 2, 4) CSD_QU_CASE(2, 2, 4) CSD_QU_CASE(4, 1, 4) CSD_QU_CASE(2, 1, 4)
#undef CSD_QU_CASE
  CSD_REQUIRE(!up4, "conv16q: no phase kernel for MQ=%d ns=%d NT=%d", p.MT, ns, p.NT);
#define CSD_Q_CASE(MQ_, NS_, NTQ_)                                                  \
  if (p.MT == MQ_ && ns == NS_ && p.NT == NTQ_ && p.stride == 1) {                  \
    if (p.PW <= 24) {                                                               \
      if (mask) return launch_q_src<MQ_, NS_, true, 24, 1, NTQ_, false>(raw, k, p.lds_bytes, s);    \
      return launch_q_src<MQ_, NS_, false, 24, 1, NTQ_, false>(raw, k, p.lds_bytes, s);             \
    }                                                                               \
    if (mask) return launch_q_src<MQ_, NS_, true, 34, 1, NTQ_, false>(raw, k, p.lds_bytes, s);      \
    return launch_q_src<MQ_, NS_, false, 34, 1, NTQ_, false>(raw, k, p.lds_bytes, s);               \
  }                                                                                 \
  if (p.MT == MQ_ && ns == NS_ && p.NT == NTQ_ && p.stride == 2) {                  \
    if (p.PW <= 24) {                                                               \
      if (mask) return launch_q_src<MQ_, NS_, true, 24, 2, NTQ_, false>(raw, k, p.lds_bytes, s);    \
      return launch_q_src<MQ_, NS_, false, 24, 2, NTQ_, false>(raw, k, p.lds_bytes, s);             \
    }                                                                               \
    if (mask) return launch_q_src<MQ_, NS_, true, 34, 2, NTQ_, false>(raw, k, p.lds_bytes, s);      \
    return launch_q_src<MQ_, NS_, false, 34, 2, NTQ_, false>(raw, k, p.lds_bytes, s);               \
  }
#define CSD_Q8_CASE(MQ_, NTQ_)                                                         \
  if (p.MT == MQ_ && ns == 3 && p.NT == NTQ_ && p.stride == 1) {                     \
    if (p.PW <= 24) {                                                                \
      if (mask) return launch_q<MQ_, 2, true, 24, 1, NTQ_, true>(k, p.lds_bytes, s);  \
      return launch_q<MQ_, 2, false, 24, 1, NTQ_, true>(k, p.lds_bytes, s);           \
    }                                                                                \
    if (mask) return launch_q<MQ_, 2, true, 34, 1, NTQ_, true>(k, p.lds_bytes, s);    \
    return launch_q<MQ_, 2, false, 34, 1, NTQ_, true>(k, p.lds_bytes, s);             \
  }                                                                                  \
  if (p.MT == MQ_ && ns == 3 && p.NT == NTQ_ && p.stride == 2) {                     \
    if (p.PW <= 24) {                                                                \
      if (mask) return launch_q<MQ_, 2, true, 24, 2, NTQ_, true>(k, p.lds_bytes, s);  \
      return launch_q<MQ_, 2, false, 24, 2, NTQ_, true>(k, p.lds_bytes, s);           \
    }                                                                                \
    if (mask) return launch_q<MQ_, 2, true, 34, 2, NTQ_, true>(k, p.lds_bytes, s);    \
    return launch_q<MQ_, 2, false, 34, 2, NTQ_, true>(k, p.lds_bytes, s);             \
  }
  CSD_REQUIRE(!(raw && ns == 3), "conv16q: the fp8-correction form reads pre-split planes");
  CSD_Q8_CASE(4, 3) CSD_Q8_CASE(2, 3) CSD_Q8_CASE(2, 1)      // (Cout % 96 == 0 only: with four cout tiles the 128-pixel form spills; those nets run the full split)
#undef CSD_Q8_CASE
  CSD_Q_CASE(4, 2, 3) CSD_Q_CASE(2, 2, 3) CSD_Q_CASE(4, 1, 3) CSD_Q_CASE(2, 1, 3) CSD_Q_CASE(2, 2, 1)
  CSD_Q_CASE(4, 2, 4) CSD_Q_CASE(2, 2, 4) CSD_Q_CASE(4, 1, 4) CSD_Q_CASE(2, 1, 4)
#undef CSD_Q_CASE
  set_error("conv16q: no kernel for MQ=%d ns=%d NT=%d", p.MT, ns, p.NT);
  return CSD_ERR_INVALID;
}

}  // namespace csd
