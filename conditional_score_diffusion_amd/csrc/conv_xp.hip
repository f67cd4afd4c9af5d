// conv_xp.hip - the fp16x3 (fp32-class) fused-prologue 3x3 convolution as ONE software-pipelined instruction stream per SIMD
// (reference models/layers.py:632-675: h = Conv(act(GroupNorm(x))) [+ Dense(temb)] / x + Conv(...); models/layerspp.py:212-274).
//
// Why a third schedule (round-4 measurements, profiles/NOTEBOOK.md): conv_ff (two 4-wave workgroups per CU, a barrier per tap) and
// conv_fx (matrix waves + producer waves) both leave the matrix pipe ~40 % busy in the split mode: a partner wave's vector work
// overlaps a dense MFMA stream by 15-20 % only, and the per-tile prologue / epilogue is exposed.  What does hide behind a matrix
// instruction is the SAME wave's next few instructions (MI355X_MICROARCH.md: ~5 single-issue instructions per 32-cycle MFMA gap with
// one wave per SIMD).  So: one 4-wave workgroup per CU (512 registers per lane), persistent over its tiles, and every wave runs
//   MFMA  |  2-5 "filler" instructions  |  MFMA  |  ...      (per tap: the products hi*lo, lo*hi, hi*hi, each round robin over the
//   2 x NT accumulators - consecutive MFMAs are independent, so the wave is free to issue the fillers while the matrix pipe works)
// where the fillers are, in program order and pinned by scheduling fences:
//   * the ds_read_b128 of the NEXT tap's fragments (register double buffer);
//   * this wave's quarter of the NEXT stage's operand patch: GroupNorm affine + exp2-domain SiLU + fp16 hi | lo split of an fp32
//     float4 that was requested one whole stage (>= 5184 matrix cycles) earlier, then the request for the stage after;
//   * this wave's quarter of the NEXT stage's weights: global -> registers at the top of the stage, registers -> LDS at its end
//     (no LDS-DMA: with a DMA in flight hipcc drains vmcnt(0) at the next use of any ordinary load);
//   * in a tile's last two stages: the residual / bias / temb requests of the epilogue.
// ONE barrier per 16-channel stage (whole-stage double buffers: patch 2 x 21 KB, weights 2 x 54 KB), placed in front of the stage's
// LAST tap: that tap's fragments are in registers, so the first fragments of the next stage are read under its 6 NT MFMAs.
// NT = 32-cout tiles per workgroup: 3 (96-cout groups) or 2 (64-cout groups, the nf = 128 networks); NORM = the GroupNorm + SiLU
// prologue (without it the fp32 source is only split: the training graph's convolutions and data gradients); RES = a residual.
// Measured (profiles/r04_power_trace.txt): back-to-back launches hold the package at its 1400 W power limit, 70 % MFMA-busy at 1.6 GHz.
// The pipeline (patch two stages ahead, weights one) runs across tile boundaries; a tile costs its K loop + a short epilogue
// (the accumulators start at zero; out = acc * k + (residual + bias + temb) * out_scale is one or two fused multiply-adds per
// element on data that is already in registers).
// Tile geometry, LDS patch layout, packed-weight layout (conv_ff.hip's NS = 2 pack) and the per-tile GroupNorm partials are conv_ff's.
#include "conv_ff.h"

// Product library: every layer this kernel covered runs on conv_xk.hip (same operator, same or - for the 64-cout groups - the Winograd
// arithmetic); the kernel is compiled in the TUNING build only, where it is conv_xk's A/B partner (CSD_XK=0, CSD_XW=0).
#ifndef CSD_TUNE
namespace csd {
bool convxp_supported(const ConvFFArgs&, int) { return false; }
int convxp_launch(const ConvFFArgs&, int, hipStream_t) {
  set_error("conv_xp: a tuning-build kernel (superseded by conv_xk.hip in the product library)");
  return CSD_ERR_INVALID;
}
}  // namespace csd
#else

namespace csd {

#define XP_THREADS 256

template <int NT_>
struct XPCfg {
  static constexpr int NT = NT_;                             // 32-cout tiles per workgroup (3: 96 couts, 2: 64 couts)
  static constexpr int TAPB = NT * 2 * 1024;                 // weight bytes per tap: NT cout tiles x (hi | lo) fragments
  static constexpr int STB = 9 * TAPB;                       // per stage (55296 | 36864)
  static constexpr int PIECES = STB / 1024;                  // 1 KiB pieces per stage (54 | 36)
  static constexpr int WPW = (PIECES + 3) / 4;               // pieces per wave (14, the last two waves repeat piece 53 | 9)
  static constexpr int WH0 = (WPW + 1) / 2, WH1 = WPW - WH0; // ... fetched in two halves (7 + 7 | 5 + 4)
  static constexpr int GP = 6 * NT;                          // MFMAs (= filler gaps) per tap: 3 products x 2 M tiles x NT
  static constexpr int NF = 4 + 2 * NT;                      // fragment reads per tap
  static constexpr int NSLOT = 6;                            // float4 conversion slots per thread and stage (1296 real slots of 1536)
  static constexpr int OFF_W = 2 * FF_PATCH_BYTES;
  static constexpr int OFF_RED = OFF_W + 2 * STB;
  static constexpr size_t LDS = (size_t)OFF_RED + 4 * NT * 32 * 2 * sizeof(float);
};

typedef unsigned int xp_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xp_pack_f16(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(int, __builtin_convertvector(f2{a, b}, h2));
}
template <bool HIGH>
__device__ __forceinline__ float xp_lo(int hp, float v) {      // v - (float)half: one v_fma_mix_f32 (exact)
  float r;
  if constexpr (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}

#define XP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define XP_SADD(x, y) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(y) : "scc")
// tuning aids (never in the product library): XP_ABL bits remove parts of the stream at compile time (results are then garbage):
// 1 conversion, 2 weight staging, 4 fragment reads, 8 epilogue stores, 16 patch requests, 32 residual requests, 64 barriers
#ifndef XP_ABL
#define XP_ABL 0
#endif
#ifdef CSD_FF_TUNE
#define XP_TS(i) do { if (a_dbg && ts_on && tid == 0) a_dbg[blockIdx.x * 32 + (i)] = clock64(); } while (0)
#define XP_WALL(i) do { if (a_dbg && tid == 0) a_dbg[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define XP_TS(i) do { } while (0)
#define XP_WALL(i) do { } while (0)
#endif

template <int NT_, bool NORM, bool RES>
__global__ __launch_bounds__(XP_THREADS, 1) void conv_xp_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = XPCfg<NT_>;
  constexpr int NT = C::NT, TAPB = C::TAPB, STB = C::STB, PIECES = C::PIECES, WPW = C::WPW, NSLOT = C::NSLOT;
  constexpr int OFF_W = C::OFF_W, OFF_RED = C::OFF_RED;
  constexpr int WH0 = C::WH0, WH1 = C::WH1, GP = C::GP, NF = C::NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const float* const a_src0 = k.a.src0;
  const float* const a_src1 = k.a.src1;
  const float* const a_bias = k.a.bias;
  const float* const a_temb = k.a.temb;
  const float* const a_res = k.a.res;
  float* const a_out = k.a.out;
  double* const a_stats = k.a.stats;
  const int a_temb_stride = k.a.temb_stride, a_out_stride = k.a.out_stride, a_out_coff = k.a.out_coff;
  const float a_out_scale = k.a.out_scale;
  const int kH = k.H, kW = k.W, kC0 = k.C0, kC1 = k.C1, kCout = k.Cout, k_tiles_x = k.tiles_x, k_tpi = k.tpi,
            k_n_groups = k.n_groups, k_nblocks = k.nblocks, NS = k.nstage;
  const int Cin = kC0 + kC1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, p32 = lane & 31, lg = tid & 3;
#ifdef CSD_FF_TUNE
  long long* const a_dbg = k.a.dbg;
  bool ts_on = false;
#endif
  XP_WALL(30);

  // ---- this workgroup's tiles: workgroup p (one per CU, on XCD p % 8) walks tiles wj, wj + P/8, ... of its XCD's contiguous share ----
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, wstride = gridDim.x >> 3;
  const int xq = k_nblocks >> 3, xr = k_nblocks & 7;
  const int x_start = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, x_len = xq + (xcd < xr ? 1 : 0);
  const int n_my = x_len > wj ? (x_len - wj + wstride - 1) / wstride : 0;
  if (n_my == 0) return;
  struct Tile { int ng, b, ty0, tx0, tile; };
  auto tile_at = [&](int it) __attribute__((always_inline)) {
    Tile t;
    it = it < n_my ? it : n_my - 1;                  // (past the end: the last tile again - harmless requests, nobody reads the result)
    const int w = x_start + wj + it * wstride;
    t.ng = w % k_n_groups;
    t.tile = w / k_n_groups;
    t.b = t.tile / k_tpi;
    const int tin = t.tile - t.b * k_tpi;
    t.ty0 = (tin / k_tiles_x) * FF_TILE;
    t.tx0 = (tin - (tin / k_tiles_x) * k_tiles_x) * FF_TILE;
    return t;
  };

  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr float NLOG2E = -1.4426950408889634f;
  // out = acc * ka + (residual + bias + temb) * out_scale; the staged operand is u / (1 + 2^u), u = -log2(e) (x s + t): SiLU = -ln2 * that
  const float ka = (NORM ? -0.6931471805599453f / C16_WSCALE : 1.0f / C16_WSCALE) * a_out_scale;

  // ---- per-lane constants ----
  // conversion slot j of thread t: 4-channel group lg = t & 3 of patch pixel min(j * 64 + (t >> 2), 323)
  int s_dst[NSLOT];                                  // LDS byte offset of the slot's hi half (lo: + 32) inside patch buffer 0
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    int pix = j * 64 + (tid >> 2);
    pix = pix < FF_NPATCH ? pix : FF_NPATCH - 1;
    const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
    s_dst[j] = pr * FF_RS + pc * FF_PSB + lg * 8;
  }
  struct Geom { int pidx[NSLOT], msk[NSLOT]; };
  auto geom_of = [&](const Tile& t) __attribute__((always_inline)) {
    Geom g;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      int pix = j * 64 + (tid >> 2);
      pix = pix < FF_NPATCH ? pix : FF_NPATCH - 1;
      const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
      const int y = t.ty0 - 1 + pr, x = t.tx0 - 1 + pc;
      const bool in = (unsigned)y < (unsigned)kH && (unsigned)x < (unsigned)kW;      // zero padding outside THIS sample
      g.pidx[j] = in ? y * kW + x : 0;               // (outside: pixel 0 of the sample is read and masked away)
      g.msk[j] = in ? -1 : 0;
    }
    return g;
  };
  // fragment bases: pixels (M operand): rows 4 wave + (p32 >> 3), columns 8 mt + (p32 & 7), K half kh; weights: lane * 16
  int xbase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) xbase[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + kh * 16;
  const int wbase = OFF_W + lane * 16;

  // ---- requests ----
  xp_u4 pf[NSLOT];                                   // the raw float4 of each slot (stage X while it is converted, then stage L)
  xp_u4 scn, shn;                                    // GroupNorm scale / shift of the lane's 4 channels: requested for stage L, used by stage X a stage later
  float msc[4], msh[4];
  // the patch requests of one stage: descriptor of (source, sample), row stride, channel offset - set once per stage (src_of), then
  // slot j = 16 bytes of pixel pidx[j], channels [st * 16 + 4 lg, + 4) of the virtual concat
  __amdgpu_buffer_rsrc_t srcL;
  int strideL = 0, soffL = 0;
  auto src_of = [&](const Tile& t, int st) __attribute__((always_inline)) {
    const int cb = st * 16;
    const bool s1 = cb >= kC0;
    const int Cs = s1 ? kC1 : kC0;
    const float* base = (s1 ? a_src1 : a_src0) + (size_t)t.b * kH * kW * Cs;
    srcL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, OOB, RSRC_FLAGS);
    strideL = Cs * 4;
    soffL = (s1 ? cb - kC0 : cb) * 4;
  };
  auto req_slot = [&](int j, const Geom& g) __attribute__((always_inline)) {
    if (!(XP_ABL & 16)) pf[j] = __builtin_amdgcn_raw_buffer_load_b128(srcL, __umul24(g.pidx[j], strideL) + lg * 16, soffL, 0);
  };
  const __amdgpu_buffer_rsrc_t nsc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(NORM ? k.a.nscale : a_src0), 0, OOB, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t nsh_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(NORM ? k.a.nshift : a_src0), 0, OOB, RSRC_FLAGS);
  auto req_norm = [&](const Tile& t, int st) __attribute__((always_inline)) {
    if constexpr (NORM) {
      const int so = (t.b * Cin + st * 16) * 4;
      scn = __builtin_amdgcn_raw_buffer_load_b128(nsc_r, (unsigned)(lg * 16), so, 0);
      shn = __builtin_amdgcn_raw_buffer_load_b128(nsh_r, (unsigned)(lg * 16), so, 0);
    }
  };
  auto cvt_prep_half = [&](int h) __attribute__((always_inline)) {       // once per stage: the affine in the exp2 domain
    if constexpr (NORM) {
#pragma unroll
      for (int q = 2 * h; q < 2 * h + 2; ++q) {
        msc[q] = __uint_as_float(scn[q]) * NLOG2E;
        msh[q] = __uint_as_float(shn[q]) * NLOG2E;
      }
      asm volatile("" : "+v"(msc[2 * h]), "+v"(msc[2 * h + 1]), "+v"(msh[2 * h]), "+v"(msh[2 * h + 1]));
    }
  };
  auto cvt_prep = [&]() __attribute__((always_inline)) { cvt_prep_half(0); cvt_prep_half(1); };
  // weights of a stage: piece i = q * 4 + wave (1 KiB, fragment order = linear; 54 pieces: the last two waves' piece 13 repeats piece
  // 53), in two halves of 7; the scalar offset runs (asm add: hipcc would otherwise precompute 14 offsets per stage ahead of the stream)
  const __amdgpu_buffer_rsrc_t w_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g_wpack), 0, OOB, RSRC_FLAGS);
  const int wvoff = lane * 16 + wave * 1024;
  // offset of the wave's last piece relative to the stage (past the end: the stage's last piece again)
  const int w_last = (((WPW - 1) * 4 + wave < PIECES ? (WPW - 1) * 4 + wave : PIECES - 1) - wave) * 1024;
  xp_u4 wr[WH0];
  int w_run = 0, w_base = 0;
  const int c4096 = 4096;
  auto w_begin = [&](int wso) __attribute__((always_inline)) { w_base = wso; w_run = wso; };
  auto req_w = [&](int q) __attribute__((always_inline)) {      // q: 0 .. WPW - 1, in order
    if (XP_ABL & 2) return;
    if (q < WPW - 1) {
      wr[q < WH0 ? q : q - WH0] = __builtin_amdgcn_raw_buffer_load_b128(w_r, (unsigned)wvoff, w_run, 0);
      XP_SADD(w_run, c4096);
    } else {
      wr[q - WH0] = __builtin_amdgcn_raw_buffer_load_b128(w_r, (unsigned)wvoff, w_base + w_last, 0);
    }
  };
  auto put_w = [&](int q, int par) __attribute__((always_inline)) {
    if (XP_ABL & 2) return;
    *reinterpret_cast<xp_u4*>(smem + OFF_W + par * STB + (q < WPW - 1 ? q * 4096 : w_last) + wvoff) = wr[q < WH0 ? q : q - WH0];
  };

  // ---- conversion of slot j into patch buffer `par`, in five pieces that sit between MFMA chains ----
  float cu[4], cv[4];
  int chp0, chp1, clp0, clp1;
  // The conversion of a slot is cut into 17 micro-steps of 2-3 instructions, one per gap between two MFMAs of a tap.  Every step
  // opens with an empty asm statement on its inputs: hipcc's instruction selection otherwise sinks the whole dependency chain to its
  // one use (the LDS store) and everything lands in ONE gap; the statement on pf[j] is also where the wait for the request lands.
#define XP_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
  auto cvt_step = [&](int j, int g, int par, int m) __attribute__((always_inline)) {
    if (XP_ABL & 1) return;
    const int h = (g & 1) * 2;                       // steps 0 .. 9 work on channels (h, h + 1)
    if (g < 2) {
      cu[h] = __uint_as_float(pf[j][h]); cu[h + 1] = __uint_as_float(pf[j][h + 1]);
      XP_PIN2(cu[h], cu[h + 1]);
      if constexpr (NORM) { cu[h] = fmaf(cu[h], msc[h], msh[h]); cu[h + 1] = fmaf(cu[h + 1], msc[h + 1], msh[h + 1]); }
    } else if (g < 4) {
      if constexpr (NORM) { XP_PIN2(cu[h], cu[h + 1]); cv[h] = __builtin_amdgcn_exp2f(cu[h]); cv[h + 1] = __builtin_amdgcn_exp2f(cu[h + 1]); }
    } else if (g < 6) {
      if constexpr (NORM) { XP_PIN2(cv[h], cv[h + 1]); cv[h] = 1.0f + cv[h]; cv[h + 1] = 1.0f + cv[h + 1]; }
    } else if (g < 8) {
      if constexpr (NORM) { XP_PIN2(cv[h], cv[h + 1]); cv[h] = __builtin_amdgcn_rcpf(cv[h]); cv[h + 1] = __builtin_amdgcn_rcpf(cv[h + 1]); }
    } else if (g < 10) {
      if constexpr (NORM) { XP_PIN2(cv[h], cv[h + 1]); cu[h] = cu[h] * cv[h]; cu[h + 1] = cu[h + 1] * cv[h + 1]; }
    } else if (g == 10) {
      XP_PIN2(cu[0], cu[1]); XP_PIN2(cu[2], cu[3]);
      chp0 = xp_pack_f16(cu[0], cu[1]);
      chp1 = xp_pack_f16(cu[2], cu[3]);
    } else if (g == 11) {
      XP_PIN2(chp0, chp1);
      cv[0] = xp_lo<false>(chp0, cu[0]); cv[1] = xp_lo<true>(chp0, cu[1]);
    } else if (g == 12) {
      XP_PIN2(chp0, chp1);
      cv[2] = xp_lo<false>(chp1, cu[2]); cv[3] = xp_lo<true>(chp1, cu[3]);
    } else if (g == 13) {
      XP_PIN2(chp0, chp1);
      char* const rec = smem + par * FF_PATCH_BYTES + s_dst[j];
      *reinterpret_cast<int2*>(rec) = make_int2(chp0 & m, chp1 & m);      // padding applies to the ACTIVATED tensor: exactly 0
    } else if (g == 14) {
      XP_PIN2(cv[0], cv[1]); XP_PIN2(cv[2], cv[3]);
      clp0 = xp_pack_f16(cv[0], cv[1]);
      clp1 = xp_pack_f16(cv[2], cv[3]);
    } else if (g == 15) {
      XP_PIN2(clp0, clp1);
      char* const rec = smem + par * FF_PATCH_BYTES + s_dst[j];
      *reinterpret_cast<int2*>(rec + 32) = make_int2(clp0 & m, clp1 & m);
    }
  };
  auto cvt_all = [&](int j, int par, int m) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 16; ++g) cvt_step(j, g, par, m);
  };

  // ---- fragments ----
  half8 xh[2][2], xl[2][2], wh[2][NT], wl[2][NT];    // [register buffer][M tile | cout tile]
  // one fragment per call (NF per tap): xh0 wl0 .. wl(NT-1) xh1 | xl0 wh0 .. wh(NT-1) xl1
  auto rd_frag = [&](int buf, int par, int tap, int which) __attribute__((always_inline)) {
    if (XP_ABL & 4) return;
    const int r = tap / 3, sx = tap - r * 3;
    const char* const pb = smem + par * FF_PATCH_BYTES + r * FF_RS + sx * FF_PSB;
    const char* const wb = smem + wbase + par * STB + tap * TAPB;
    if (which == 0) xh[buf][0] = *reinterpret_cast<const half8*>(pb + xbase[0]);
    else if (which <= NT) wl[buf][which - 1] = *reinterpret_cast<const half8*>(wb + (which - 1) * 2048 + 1024);
    else if (which == NT + 1) xh[buf][1] = *reinterpret_cast<const half8*>(pb + xbase[1]);
    else if (which == NT + 2) xl[buf][0] = *reinterpret_cast<const half8*>(pb + xbase[0] + 32);
    else if (which <= 2 * NT + 2) wh[buf][which - NT - 3] = *reinterpret_cast<const half8*>(wb + (which - NT - 3) * 2048);
    else if (which == 2 * NT + 3) xl[buf][1] = *reinterpret_cast<const half8*>(pb + xbase[1] + 32);
  };

  // The accumulators live in the accumulator half of the register file for the whole kernel ("+a"); the matrix instructions are asm
  // statements so that hipcc neither moves them between register classes at control-flow joins nor reorders them.
  floatx16 acc[2][NT];
  // zeroing is a matrix instruction too (0 * 0 + 0 on a zero fragment): an assignment in C++ makes hipcc keep 96 zero constants in
  // registers and copy them around at every control-flow join
  auto zero_acc = [&]() __attribute__((always_inline)) {
    half8 z;
#pragma unroll
    for (int q = 0; q < 8; ++q) z[q] = (_Float16)0.f;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[mt][nt]) : "v"(z));
  };
  // One MFMA per statement, round robin over the six accumulators: product p of a tap (0: hi * lo, 1: lo * hi, 2: hi * hi - small terms
  // first) for all six, then the next product.  Consecutive MFMAs are independent, so the wave is free to issue the gap's fillers while
  // the matrix pipe works (tools/xp_order_probe.hip: ~80 fillers per tap ride for +5 % here; with three dependent MFMAs back to back
  // the wave sits in the dependency stall and 46 fillers per tap already cost +22 %).  Per accumulator the order of the additions is
  // unchanged.
  auto mm = [&](int buf, int p, int i) __attribute__((always_inline)) {
    const int mt = i / NT, nt = i - mt * NT;
    if (p == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(xh[buf][mt]), "v"(wl[buf][nt]));
    else if (p == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(xl[buf][mt]), "v"(wh[buf][nt]));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[mt][nt]) : "v"(xh[buf][mt]), "v"(wh[buf][nt]));
  };

  // hipcc does not know that the asm statements are matrix instructions, and the hardware does not interlock a matrix write against a
  // vector read of the same register (11 wait states for the 8-pass 32x32x16).  Two consequences:
  //  * the epilogue's reads of the accumulators sit behind tie_acc_done(), a wait tied to every accumulator so that no read can move
  //    above it;
  //  * hipcc must never move an accumulator itself.  It did: with the accumulators live across the loop's back edge (and across a
  //    conditional head of unit 0) the two sides of the join held them in different registers, and the 16 v_accvgpr_mov per
  //    accumulator landed directly behind / in front of matrix instructions - reading an accumulator before its last update had
  //    landed (seen as wrong first elements of one accumulator with 64-cout groups) and costing ~200 vector instructions per tile.
  //    Hence the loop structure below: no accumulator is live across any control-flow edge except the stage-pair loop's, where both
  //    sides are the same code.  tools/check_xp_isa.py (run by the build) fails if anything but a matrix instruction or a read behind
  //    tie_acc_done() touches an accumulator register.
  auto tie_acc_done = [&]() __attribute__((always_inline)) {
    if constexpr (NT == 3)
      asm volatile("s_nop 15\n\ts_nop 7"
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]));
    else
      asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));
  };

  // ---- epilogue operands of a tile (requested while its last stage is multiplied, used after that stage's last tap) ----
  // Element (mt, r) of a lane: pixel row 4 wave + r / 4, column 8 mt + 4 kh + r % 4 of the tile, cout nt * 32 + p32 of the group: the
  // lane part (kh, cout, the wave's first row) is one voffset per tensor, nt * 128 an immediate, and the (mt, r) part a RUNNING
  // scalar offset advanced by one of three strides (next column, next row, next M tile) - 96 precomputed scalar offsets per tensor
  // do not fit the scalar registers and came back from spill lanes with a v_readlane + wait states in front of every access.
  float rs[2][NT][16];
  float bv[NT], tv[NT];
  __amdgpu_buffer_rsrc_t res_r, out_r;
  unsigned res_voff = 0, out_voff = 0;
  int e_tile = 0, e_ng = 0;
  struct Steps { int col, row, mtile; };
  auto steps_of = [&](int stride) __attribute__((always_inline)) {
    Steps t;
    t.col = stride * 4;
    t.row = (kW - 3) * stride * 4;                   // from column 3 of a row to column 0 of the next
    t.mtile = (8 - 3 * kW - 3) * stride * 4;         // from (row 3, column 3) of M tile 0 to (row 0, column 0) of M tile 1
    return t;
  };
  const Steps st_res = steps_of(kCout), st_out = steps_of(a_out_stride);
  auto e_advance = [&](int& run, int idx, const Steps& st) __attribute__((always_inline)) {      // from element idx = mt * 16 + r to idx + 1
    const int r = idx & 15;
    if ((r & 3) != 3) XP_SADD(run, st.col);
    else if (r != 15) XP_SADD(run, st.row);
    else XP_SADD(run, st.mtile);
  };
  int res_run = 0;
  auto epi_setup = [&](const Tile& t) __attribute__((always_inline)) {      // in the tile's last unit
    const size_t tile_pix = (size_t)t.b * kH * kW + (size_t)t.ty0 * kW + t.tx0;
    const int c_lane = t.ng * NT * 32 + p32;
    out_r = __builtin_amdgcn_make_buffer_rsrc(a_out + tile_pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
    out_voff = (unsigned)((4 * wave * kW + 4 * kh) * a_out_stride + c_lane) * 4u;
    e_tile = t.tile; e_ng = t.ng;
    if constexpr (RES) {
      res_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_res + tile_pix * kCout), 0, OOB, RSRC_FLAGS);
      res_voff = (unsigned)((4 * wave * kW + 4 * kh) * kCout + c_lane) * 4u;
      res_run = 0;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bv[nt] = a_bias ? a_bias[c_lane + nt * 32] : 0.f;      // (summed in the epilogue: an add here would wait for both requests at once)
      tv[nt] = a_temb ? a_temb[(size_t)t.b * a_temb_stride + c_lane + nt * 32] : 0.f;
    }
  };
  auto req_res = [&](int e) __attribute__((always_inline)) {      // e = (mt * 16 + r) * NT + nt
    if constexpr (RES) {
      if (XP_ABL & 32) return;
      const int idx = e / NT, nt = e - idx * NT;
      rs[idx / 16][nt][idx % 16] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, res_voff + nt * 128, res_run, 0));
      if (nt == NT - 1 && idx < 31) e_advance(res_run, idx, st_res);
    }
  };
  float* const red = reinterpret_cast<float*>(smem + OFF_RED);      // [4 waves][NT*32 couts][2]: statistics hand-over
  auto epilogue = [&]() __attribute__((always_inline)) {
    tie_acc_done();
    XP_TS(22);
    float vs[NT], vq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) vs[nt] = vq[nt] = 0.f;
    int out_run = 0;
#pragma unroll
    for (int idx = 0; idx < 32; ++idx) {
      const int mt = idx / 16, r = idx % 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bs = (bv[nt] + tv[nt]) * a_out_scale;
        float v;
        if constexpr (RES) v = fmaf(acc[mt][nt][r], ka, fmaf(rs[mt][nt][r], a_out_scale, bs));
        else v = fmaf(acc[mt][nt][r], ka, bs);
        if (!(XP_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), out_r, out_voff + nt * 128, out_run, 0);
        vs[nt] += v;
        vq[nt] = fmaf(v, v, vq[nt]);
      }
      if (idx < 31) e_advance(out_run, idx, st_out);
    }
    XP_TS(23);
    if (a_stats) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        vs[nt] += __shfl_xor(vs[nt], 32);
        vq[nt] += __shfl_xor(vq[nt], 32);
        if (kh == 0) {
          red[(wave * NT * 32 + nt * 32 + p32) * 2 + 0] = vs[nt];
          red[(wave * NT * 32 + nt * 32 + p32) * 2 + 1] = vq[nt];
        }
      }
      ff_barrier();
      if (tid < NT * 32) {
        double sm = 0.0, sq = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          sm += (double)red[(wv * NT * 32 + tid) * 2 + 0];
          sq += (double)red[(wv * NT * 32 + tid) * 2 + 1];
        }
        double* dst = a_stats + ((size_t)e_tile * kCout + e_ng * NT * 32 + tid) * 2;
        dst[0] = sm;
        dst[1] = sq;
      }
    }
    XP_TS(24);
  };

  Tile tc = tile_at(0), tn = tile_at(1);
  Geom gc = geom_of(tc), gn = gc;

  // =========================================================================================================================
  // prologue: stage 0 converted + its weights stored, stage 1 requested, the first fragments read
  // =========================================================================================================================
  src_of(tc, 0);
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) req_slot(j, gc);
  req_norm(tc, 0);
  {
    w_begin((tc.ng * (Cin / 16) + 0) * STB);
    cvt_prep();
    req_norm(tc, 1);
    src_of(tc, 1);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      cvt_all(j, 0, gc.msk[j]);
      req_slot(j, gc);
    }
#pragma unroll
    for (int q = 0; q < WH0; ++q) req_w(q);
#pragma unroll
    for (int q = 0; q < WH0; ++q) put_w(q, 0);
#pragma unroll
    for (int q = WH0; q < WPW; ++q) req_w(q);
#pragma unroll
    for (int q = WH0; q < WPW; ++q) put_w(q, 0);
  }
  ff_barrier();
#pragma unroll
  for (int wch = 0; wch < NF; ++wch) rd_frag(0, 0, 0, wch);

  // =========================================================================================================================
  // One unit = [barrier | tap 8 of the PREVIOUS stage | (first unit of a tile: the previous tile's epilogue) | taps 0 .. 7 of stage s].
  // A stage has 9 taps x 18 MFMAs; while stage s of tile tc is multiplied (buffers `par`), stage X = s + 1 is converted into the other
  // buffers from the registers requested a stage ago, its weights fetched and stored, and stage L = s + 2 is requested.  The unit is
  // cut in front of tap 8 because that is where the barrier sits (every wave holds tap 8's fragments in registers: the previous
  // stage's buffers are dead, this stage's complete), and tap 8's gaps - free apart from the first fragment reads of stage s - take
  // the stage's scalar set-up (descriptors, offsets, the GroupNorm affine in the exp2 domain) instead of a preamble in front of the
  // first MFMA.
  // POS 0: s + 2 < NS (X and L in this tile); POS 1: s = NS - 2 (L = stage 0 of the next tile); POS 2: s = NS - 1, the tile's last
  // stage (X = stage 0, L = stage 1 of the next tile; the epilogue's requests ride along).  FLIP: nine taps per stage - the
  // fragment register buffer of tap 0 alternates from stage to stage.  FIRST: stage 0 of a tile.
  // =========================================================================================================================
  int par = 0;                                       // buffer parity of the stage whose taps 0 .. 7 run (or ran last)
  auto unit = [&](auto flip_tag, auto pos_tag, auto first_tag, auto part_tag, int s) __attribute__((always_inline)) {
    constexpr int FLIP = decltype(flip_tag)::value, POS = decltype(pos_tag)::value, PART = decltype(part_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr bool HEAD = PART == 0 || PART == 2, BODY = PART == 0 || PART == 1;      // (PART 3: the set-up pieces alone)
    const Tile& tx = POS == 2 ? tn : tc;             // tile of stage X
    const Tile& tl = POS >= 1 ? tn : tc;             // tile of stage L
    const Geom& gx = POS == 2 ? gn : gc;
    const Geom& gl = POS >= 1 ? gn : gc;
    const int sx = POS == 2 ? 0 : s + 1, sl = POS == 0 ? s + 2 : POS - 1;
    auto setup = [&](int g) __attribute__((always_inline)) {      // the stage's set-up in pieces g = 0 .. 7
      if (g == 0) cvt_prep_half(0);                  // stage X's scale / shift (requested a stage ago)
      else if (g == 1) cvt_prep_half(1);
      else if (g == 2) req_norm(tl, sl);
      else if (g == 3) src_of(tl, sl);
      else if (g == 4) w_begin((tx.ng * (Cin / 16) + sx) * STB);
      else if (g == 5) { if constexpr (POS == 1) epi_setup(tc); }
    };
    if constexpr (PART == 3) {
#pragma unroll
      for (int g = 0; g < 8; ++g) setup(g);
    }
    if constexpr (HEAD) {
      par ^= 1;
      XP_FENCE();
      if (s == 2) XP_TS(20);
      if (!(XP_ABL & 64)) ff_barrier();
      if (s == 2) XP_TS(21);
      XP_FENCE();
      constexpr int buf = 1 - FLIP;                  // (= (8 + the previous stage's FLIP) & 1)
#pragma unroll
      for (int g = 0; g < GP; ++g) {
        XP_FENCE();
        mm(buf, g / (2 * NT), g % (2 * NT));
        XP_FENCE();
        if (g < NF) rd_frag(buf ^ 1, par, 0, g);
        else {                                       // the six set-up pieces over the remaining gaps (at most two per gap)
          constexpr int R = GP - NF;
          const int q0 = (6 * (g - NF) + R - 1) / R, q1 = (6 * (g - NF + 1) + R - 1) / R;
          if (q0 < q1) setup(q0);
          if (q0 + 1 < q1) setup(q0 + 1);
        }
      }
      XP_FENCE();
      if constexpr (FIRST) epilogue();
    }
    if constexpr (!BODY) return;
    XP_TS(0 + s);
    const int npar = par ^ 1;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int buf = (tap + FLIP) & 1;
#pragma unroll
      for (int g = 0; g < GP; ++g) {
        XP_FENCE();
        mm(buf, g / (2 * NT), g % (2 * NT));
        XP_FENCE();
        // ---- fillers of gap (tap, g) ----
        if (g < NF) rd_frag(buf ^ 1, par, tap + 1, g);
        if (tap >= 1 && tap <= 6) {
          // conversion of slot tap - 1 (stage X) in 16 micro-steps, then the request of the same slot for stage L as step 16.
          // NT = 3: one step per gap.  NT = 2: step q in gap q * GP / 17 (at most two per gap).  (Two spellings on purpose: with the
          // general one hipcc leaves the NT = 3 residual registers in scratch memory.)
          if constexpr (NT == 3) {
            if (g < 16) cvt_step(tap - 1, g, npar, gx.msk[tap - 1]);
            else if (g == 16) req_slot(tap - 1, gl);
          } else {
            const int q0 = (17 * g + GP - 1) / GP, q1 = (17 * (g + 1) + GP - 1) / GP;
            if (q0 < q1 && q0 < 16) cvt_step(tap - 1, q0, npar, gx.msk[tap - 1]);
            if (q0 + 1 < q1 && q0 + 1 < 16) cvt_step(tap - 1, q0 + 1, npar, gx.msk[tap - 1]);
            if (q0 <= 16 && 16 < q1) req_slot(tap - 1, gl);
          }
        }
        // stage X's weights: first half requested in tap 0, stored in tap 3; second half requested in tap 4, stored in tap 7
        if (tap == 0 && g >= GP - 1 - WH0 && g < GP - 1) req_w(g - (GP - 1 - WH0));
        if (tap == 4 && (g & 1) == 0 && g / 2 < WH1) req_w(WH0 + g / 2);
        if (tap == 3 && (g & 1) == 1 && g / 2 < WH0) put_w(g / 2, npar);
        if (tap == 7 && g < WH1) put_w(WH0 + g, npar);
        // 32 NT residual requests over the tile's last two units, in taps 1-2 and 5-6 only (two of every three gaps).  The memory counter
        // retires in order: a weight piece (an L2 hit, stored to LDS three taps after its request) cannot retire before an OLDER request
        // that went to HBM - so no slow request may be issued in the ~3 taps in front of a weight request (taps 0 and 4).
        if constexpr (POS >= 1) {
          if ((tap == 1 || tap == 2 || tap == 5 || tap == 6) && g % 3 != 0) {
            const int w = (tap == 1 ? 0 : tap == 2 ? 1 : tap == 5 ? 2 : 3);
            req_res((POS - 1) * 16 * NT + w * 4 * NT + (g / 3) * 2 + (g % 3 - 1));
          }
        }
      }
    }
    XP_FENCE();
  };
  using F0 = std::integral_constant<int, 0>;
  using F1 = std::integral_constant<int, 1>;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using Whole = std::integral_constant<int, 0>;     // parts of a unit: head + taps, the taps alone, the head alone, the set-up alone
  using Taps = std::integral_constant<int, 1>;
  using Head = std::integral_constant<int, 2>;
  using SetUp = std::integral_constant<int, 3>;
  using Yes = std::true_type;
  using No = std::false_type;

  // =========================================================================================================================
  // persistent loop (NS even, >= 4): stage pairs, the first and the last pair of a tile peeled.  Unit 0 of a tile is cut in two: its
  // taps open the iteration, and the iteration closes with the head of the NEXT tile's unit 0 (= this tile's last tap, under which the
  // next tile's first fragments are read, and this tile's epilogue).  The back edge therefore sits where no accumulator is live -
  // they are zeroed at the top - and the last iteration needs no special tail (its "next tile" is the last tile again: harmless reads).
  // =========================================================================================================================
  unit(F0{}, P0{}, Yes{}, SetUp{}, 0);
  for (int it = 0; it < n_my; ++it) {
#ifdef CSD_FF_TUNE
    ts_on = it == 1;
    if (it == 1) XP_WALL(28);
    if (it == 2) XP_WALL(29);
    if (it == 1 && a_dbg && tid == 0) a_dbg[blockIdx.x * 32 + 26] = clock64();
    if (it == 2 && a_dbg && tid == 0) a_dbg[blockIdx.x * 32 + 27] = clock64();
#endif
    zero_acc();
    unit(F0{}, P0{}, Yes{}, Taps{}, 0);
    unit(F1{}, P0{}, No{}, Whole{}, 1);
    for (int s = 2; s + 2 < NS; s += 2) {
      unit(F0{}, P0{}, No{}, Whole{}, s);
      unit(F1{}, P0{}, No{}, Whole{}, s + 1);
    }
    gn = geom_of(tn);
    unit(F0{}, P1{}, No{}, Whole{}, NS - 2);
    unit(F1{}, P2{}, No{}, Whole{}, NS - 1);
    XP_TS(0 + NS);
    tc = tn;
    gc = gn;
    tn = tile_at(it + 2);
    unit(F0{}, P0{}, Yes{}, Head{}, 0);
  }
  XP_WALL(31);
}

template <int NT, bool NORM, bool RES>
static int launch_xp(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_xp_kernel<NT, NORM, RES>;
  CSD_SET_MAX_LDS_ONCE(kern);
  const int n_cu = device_cu_count8();               // persistent: one workgroup per CU, a multiple of the 8 XCDs
  const int grid = k.nblocks < n_cu ? (k.nblocks + 7) / 8 * 8 : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(XP_THREADS), XPCfg<NT>::LDS, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// the fp16x3 (ns = 2) layers conv_ff covers (96- or 64-cout groups) with an even number >= 4 of stages; same packed weights, same arguments
bool convxp_supported(const ConvFFArgs& k, int nt) { return (nt == 3 || nt == 2) && k.nstage >= 4 && k.nstage % 2 == 0; }

template <int NT>
static int launch_xp_nt(const ConvFFArgs& k, hipStream_t s) {
  const bool norm = k.a.nscale != nullptr, res = k.a.res != nullptr;
  if (norm) return res ? launch_xp<NT, true, true>(k, s) : launch_xp<NT, true, false>(k, s);
  return res ? launch_xp<NT, false, true>(k, s) : launch_xp<NT, false, false>(k, s);
}

int convxp_launch(const ConvFFArgs& k, int nt, hipStream_t s) { return nt == 3 ? launch_xp_nt<3>(k, s) : launch_xp_nt<2>(k, s); }

}  // namespace csd
#endif  // CSD_TUNE
