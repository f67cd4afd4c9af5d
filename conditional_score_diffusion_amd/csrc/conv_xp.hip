// conv_xp.hip - the fp16x3 (fp32-class) fused-prologue 3x3 convolution as ONE software-pipelined instruction stream per SIMD
// (reference models/layers.py:632-675: h = Conv(act(GroupNorm(x))) [+ Dense(temb)] / x + Conv(...); models/layerspp.py:212-274).
//
// Why a third schedule (round-4 measurements, profiles/NOTEBOOK.md): conv_ff (two 4-wave workgroups per CU, a barrier per tap) and
// conv_fx (matrix waves + producer waves) both leave the matrix pipe ~40 % busy in the split mode: a partner wave's vector work
// overlaps a dense MFMA stream by 15-20 % only, and the per-tile prologue / epilogue is exposed.  What does hide behind a matrix
// instruction is the SAME wave's next few instructions (MI355X_MICROARCH.md: ~5 single-issue instructions per 32-cycle MFMA gap with
// one wave per SIMD).  So: one 4-wave workgroup per CU (512 registers per lane), persistent over its tiles, and every wave runs
//   chain of 3 MFMAs (lo*hi, hi*lo, hi*hi on one accumulator, back to back)  |  a handful of "filler" instructions  |  next chain ...
// where the fillers are, in program order and pinned by scheduling fences:
//   * the ds_read_b128 of the NEXT tap's fragments (register double buffer);
//   * this wave's quarter of the NEXT stage's operand patch: GroupNorm affine + exp2-domain SiLU + fp16 hi | lo split of an fp32
//     float4 that was requested one whole stage (>= 5184 matrix cycles) earlier, then the request for the stage after;
//   * this wave's quarter of the NEXT stage's weights: global -> registers at the top of the stage, registers -> LDS at its end
//     (no LDS-DMA: with a DMA in flight hipcc drains vmcnt(0) at the next use of any ordinary load);
//   * in a tile's last stage: the residual / bias / temb requests of the epilogue.
// ONE barrier per 16-channel stage (whole-stage double buffers: patch 2 x 21 KB, weights 2 x 54 KB), placed in front of the stage's
// LAST tap: that tap's fragments are in registers, so the first fragments of the next stage are read under its 18 MFMAs.
// The pipeline (patch two stages ahead, weights one) runs across tile boundaries; a tile costs its K loop + a short epilogue
// (the accumulators start at zero; out = acc * k + (residual + bias + temb) * out_scale is one or two fused multiply-adds per
// element on data that is already in registers).
// Tile geometry, LDS patch layout, packed-weight layout (conv_ff.hip's NS = 2 pack) and the per-tile GroupNorm partials are conv_ff's.
#include "conv_ff.h"

namespace csd {

#define XP_THREADS 256

struct XPCfg {
  static constexpr int NT = 3;                               // 32-cout tiles per workgroup (96 couts)
  static constexpr int TAPB = NT * 2 * 1024;                 // weight bytes per tap: NT cout tiles x (hi | lo) fragments
  static constexpr int STB = 9 * TAPB;                       // per stage (55296)
  static constexpr int PIECES = STB / 1024;                  // 1 KiB pieces per stage (54)
  static constexpr int WPW = (PIECES + 3) / 4;               // pieces per wave (14; the last two waves repeat piece 53)
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

template <bool NORM, bool RES>
__global__ __launch_bounds__(XP_THREADS, 1) void conv_xp_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = XPCfg;
  constexpr int NT = C::NT, TAPB = C::TAPB, STB = C::STB, PIECES = C::PIECES, WPW = C::WPW, NSLOT = C::NSLOT;
  constexpr int OFF_W = C::OFF_W, OFF_RED = C::OFF_RED;
  constexpr int WH = WPW / 2;                        // weight pieces per half stage and wave (7)
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
  auto cvt_prep = [&]() __attribute__((always_inline)) {       // once per stage: the affine in the exp2 domain
    if constexpr (NORM) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        msc[q] = __uint_as_float(scn[q]) * NLOG2E;
        msh[q] = __uint_as_float(shn[q]) * NLOG2E;
      }
    }
  };
  // weights of a stage: piece i (1 KiB, fragment order = linear); this wave takes pieces wave, wave + 4, ... in two halves of 7
  const __amdgpu_buffer_rsrc_t w_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g_wpack), 0, OOB, RSRC_FLAGS);
  xp_u4 wr[WH];
  auto req_w = [&](int q, int wso) __attribute__((always_inline)) {      // q: 0 .. 13
    if (XP_ABL & 2) return;
    int i = q * 4 + wave;
    i = i < PIECES ? i : PIECES - 1;
    wr[q % WH] = __builtin_amdgcn_raw_buffer_load_b128(w_r, (unsigned)(lane * 16), wso + i * 1024, 0);
  };
  auto put_w = [&](int q, int par) __attribute__((always_inline)) {
    if (XP_ABL & 2) return;
    int i = q * 4 + wave;
    i = i < PIECES ? i : PIECES - 1;
    *reinterpret_cast<xp_u4*>(smem + OFF_W + par * STB + i * 1024 + lane * 16) = wr[q % WH];
  };

  // ---- conversion of slot j into patch buffer `par`, in five pieces that sit between MFMA chains ----
  float cu[4], cv[4];
  int chp0, chp1;
  // (every piece opens with an empty asm statement on its inputs: hipcc's instruction selection otherwise sinks the whole conversion
  // to its one use, the LDS store, and the five pieces land in ONE gap; the statement on pf[j] is also where the wait for the
  // request lands)
#define XP_PIN4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
  auto cvt_piece = [&](int j, int piece, int par, int m) __attribute__((always_inline)) {
    if (XP_ABL & 1) return;
    if (piece == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) cu[q] = __uint_as_float(pf[j][q]);
      XP_PIN4(cu);
      if constexpr (NORM) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cu[q] = fmaf(cu[q], msc[q], msh[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) cv[q] = __builtin_amdgcn_exp2f(cu[q]);
      }
    } else if (piece == 1) {
      if constexpr (NORM) {
        XP_PIN4(cv);
#pragma unroll
        for (int q = 0; q < 4; ++q) cv[q] = __builtin_amdgcn_rcpf(1.0f + cv[q]);
      }
    } else if (piece == 2) {
      if constexpr (NORM) {
        XP_PIN4(cv);
#pragma unroll
        for (int q = 0; q < 4; ++q) cu[q] = cu[q] * cv[q];
      } else {
        XP_PIN4(cu);
      }
      chp0 = xp_pack_f16(cu[0], cu[1]);
      chp1 = xp_pack_f16(cu[2], cu[3]);
    } else if (piece == 3) {
      asm volatile("" : "+v"(chp0), "+v"(chp1));
      cv[0] = xp_lo<false>(chp0, cu[0]); cv[1] = xp_lo<true>(chp0, cu[1]);
      cv[2] = xp_lo<false>(chp1, cu[2]); cv[3] = xp_lo<true>(chp1, cu[3]);
      char* const rec = smem + par * FF_PATCH_BYTES + s_dst[j];
      *reinterpret_cast<int2*>(rec) = make_int2(chp0 & m, chp1 & m);      // padding applies to the ACTIVATED tensor: exactly 0
    } else {
      XP_PIN4(cv);
      char* const rec = smem + par * FF_PATCH_BYTES + s_dst[j];
      *reinterpret_cast<int2*>(rec + 32) = make_int2(xp_pack_f16(cv[0], cv[1]) & m, xp_pack_f16(cv[2], cv[3]) & m);
    }
  };

  // ---- fragments ----
  half8 xh[2][2], xl[2][2], wh[2][NT], wl[2][NT];    // [register buffer][M tile | cout tile]
  auto rd_frag = [&](int buf, int par, int tap, int which) __attribute__((always_inline)) {
    if (XP_ABL & 4) return;
    const int r = tap / 3, sx = tap - r * 3;
    const char* const pb = smem + par * FF_PATCH_BYTES + r * FF_RS + sx * FF_PSB;
    const char* const wb = smem + wbase + par * STB + tap * TAPB;
    switch (which) {
      case 0: xh[buf][0] = *reinterpret_cast<const half8*>(pb + xbase[0]); xl[buf][0] = *reinterpret_cast<const half8*>(pb + xbase[0] + 32); break;
      case 1: wh[buf][0] = *reinterpret_cast<const half8*>(wb); wl[buf][0] = *reinterpret_cast<const half8*>(wb + 1024); break;
      case 2: xh[buf][1] = *reinterpret_cast<const half8*>(pb + xbase[1]); xl[buf][1] = *reinterpret_cast<const half8*>(pb + xbase[1] + 32); break;
      case 3: wh[buf][1] = *reinterpret_cast<const half8*>(wb + 2048); wl[buf][1] = *reinterpret_cast<const half8*>(wb + 3072); break;
      case 4: wh[buf][2] = *reinterpret_cast<const half8*>(wb + 4096); break;
      default: wl[buf][2] = *reinterpret_cast<const half8*>(wb + 5120); break;
    }
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
  zero_acc();
  auto chain = [&](int buf, int i) __attribute__((always_inline)) {
    const int mt = i / NT, nt = i - mt * NT;
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"       // small terms first: hi * lo, lo * hi, then hi * hi
                 "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"
                 "v_mfma_f32_32x32x16_f16 %0, %1, %4, %0"
                 : "+a"(acc[mt][nt])
                 : "v"(xh[buf][mt]), "v"(wl[buf][nt]), "v"(xl[buf][mt]), "v"(wh[buf][nt]));
  };

  // epilogue operands of the tile being multiplied (requested in its last stage)
  float rs[2][NT][16];
  float bv[NT], tv[NT];
  // element (mt, r) of a lane: pixel row 4 wave + r / 4, column 8 mt + 4 kh + r % 4 (kh rides in the lane's voffset)
  auto e_off = [&](int mt, int r, int stride) __attribute__((always_inline)) { return ((4 * wave + (r >> 2)) * kW + 8 * mt + (r & 3)) * stride * 4; };

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
    const int wso = (tc.ng * (Cin / 16) + 0) * STB;
    cvt_prep();
    req_norm(tc, 1);
    src_of(tc, 1);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
#pragma unroll
      for (int piece = 0; piece < 5; ++piece) cvt_piece(j, piece, 0, gc.msk[j]);
      req_slot(j, gc);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int q = 0; q < WH; ++q) req_w(h * WH + q, wso);
#pragma unroll
      for (int q = 0; q < WH; ++q) put_w(h * WH + q, 0);
    }
  }
  ff_barrier();
#pragma unroll
  for (int wch = 0; wch < 6; ++wch) rd_frag(0, 0, 0, wch);

  // =========================================================================================================================
  // one stage = 9 taps x 6 chains.  While stage s of tile tc is multiplied (buffers `par`), stage X = s + 1 is converted into the
  // other buffers from the registers requested a stage ago, its weights fetched and stored, and stage L = s + 2 is requested.
  // POS 0: s + 2 < NS (X and L in this tile); POS 1: s = NS - 2 (L = stage 0 of the next tile); POS 2: s = NS - 1, the tile's last
  // stage (X = stage 0, L = stage 1 of the next tile; the epilogue's requests ride along).  FLIP: nine taps per stage - the
  // fragment register buffer of tap 0 alternates from stage to stage.
  // =========================================================================================================================
  int par = 0;                                       // buffer parity of the stage being multiplied
  auto stage = [&](auto flip_tag, auto pos_tag, int s) __attribute__((always_inline)) {
    constexpr int FLIP = decltype(flip_tag)::value, POS = decltype(pos_tag)::value;
    const int npar = par ^ 1;
    const Tile& tx = POS == 2 ? tn : tc;             // tile of stage X
    const Tile& tl = POS >= 1 ? tn : tc;             // tile of stage L
    const Geom& gx = POS == 2 ? gn : gc;
    const Geom& gl = POS >= 1 ? gn : gc;
    const int sx = POS == 2 ? 0 : s + 1, sl = POS == 0 ? s + 2 : POS - 1;
    cvt_prep();                                      // stage X's scale / shift (requested a stage ago)
    req_norm(tl, sl);
    src_of(tl, sl);
    const int wso = (tx.ng * (Cin / 16) + sx) * STB;
    unsigned res_voff = 0;
    __amdgpu_buffer_rsrc_t res_r = w_r;
    if constexpr (POS == 2) {
      const size_t tile_pix = (size_t)tc.b * kH * kW + (size_t)tc.ty0 * kW + tc.tx0;
      const int c_lane = tc.ng * NT * 32 + p32;
      if constexpr (RES) {
        res_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_res + tile_pix * kCout), 0, OOB, RSRC_FLAGS);
        res_voff = (unsigned)(4 * kh * kCout + c_lane) * 4u;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bv[nt] = a_bias ? a_bias[c_lane + nt * 32] : 0.f;      // (summed in the epilogue: an add here would wait for both requests at once)
        tv[nt] = a_temb ? a_temb[(size_t)tc.b * a_temb_stride + c_lane + nt * 32] : 0.f;
      }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int buf = (tap + FLIP) & 1;
      if (tap == 8) {                                // every wave holds tap 8's fragments: the stage's buffers are dead, the next stage's complete
        XP_FENCE();
        if (s == 2) XP_TS(20);
        if (!(XP_ABL & 64)) ff_barrier();
        if (s == 2) XP_TS(21);
        XP_FENCE();
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        XP_FENCE();
        chain(buf, i);
        XP_FENCE();
        // ---- fillers of gap (tap, i) ----
        if (tap < 8) rd_frag(buf ^ 1, par, tap + 1, i);
        else rd_frag(buf ^ 1, npar, 0, i);
        if (tap >= 1 && tap <= 6) {                  // conversion of slot tap - 1 (stage X), then the request of the same slot for stage L
          if (i < 5) cvt_piece(tap - 1, i, npar, gx.msk[tap - 1]);
          else req_slot(tap - 1, gl);
        }
        // stage X's weights: first half requested in tap 0, stored in tap 3; second half requested in tap 4, stored in tap 7
        if (tap == 0 || tap == 4) {
          constexpr int n0[7] = {0, 2, 4, 5, 6, 7, 7};
#pragma unroll
          for (int q = n0[i]; q < n0[i + 1]; ++q) req_w((tap / 4) * WH + q, wso);
        } else if (tap == 3 || tap == 7) {
          constexpr int n3[7] = {0, 2, 4, 5, 6, 7, 7};
#pragma unroll
          for (int q = n3[i]; q < n3[i + 1]; ++q) put_w((tap / 4) * WH + q, npar);
        }
        if constexpr (POS == 2 && RES) {
          if (tap < 8 && !(XP_ABL & 32)) {           // 96 residual requests, 2 per gap
            const int g0 = (tap * 6 + i) * 2;
#pragma unroll
            for (int e = g0; e < g0 + 2; ++e) {
              const int mt = e / 48, nt = (e / 16) % 3, r = e % 16;
              rs[mt][nt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, res_voff + nt * 128, e_off(mt, r, kCout), 0));
            }
          }
        }
      }
    }
    XP_FENCE();
    XP_TS(1 + s);
    par = npar;
  };
  using F0 = std::integral_constant<int, 0>;
  using F1 = std::integral_constant<int, 1>;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;

  // =========================================================================================================================
  // persistent loop (NS even): stage pairs, the last pair peeled
  // =========================================================================================================================
  float* const red = reinterpret_cast<float*>(smem + OFF_RED);      // [4 waves][NT*32 couts][2]: statistics hand-over
  for (int it = 0; it < n_my; ++it) {
#ifdef CSD_FF_TUNE
    ts_on = it == 1;
#endif
    XP_TS(0);
    for (int s = 0; s + 2 < NS; s += 2) {
      stage(F0{}, P0{}, s);
      stage(F1{}, P0{}, s + 1);
    }
    gn = geom_of(tn);
    stage(F0{}, P1{}, NS - 2);
    stage(F1{}, P2{}, NS - 1);
    // ---- epilogue of tile tc ----
    {
      // (hipcc does not know the asm statements are MFMAs: without this their results would be read a few cycles after issue; the
      // operands tie every accumulator to the statement so that no read can move above it)
      asm volatile("s_nop 15\n\ts_nop 7"
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]));
      XP_TS(22);
      const size_t tile_pix = (size_t)tc.b * kH * kW + (size_t)tc.ty0 * kW + tc.tx0;
      const int c_lane = tc.ng * NT * 32 + p32;
      const __amdgpu_buffer_rsrc_t out_r =
          __builtin_amdgcn_make_buffer_rsrc(a_out + tile_pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
      const unsigned out_voff = (unsigned)(4 * kh * a_out_stride + c_lane) * 4u;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bs = (bv[nt] + tv[nt]) * a_out_scale;
        float vs = 0.f, vq = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v;
            if constexpr (RES) v = fmaf(acc[mt][nt][r], ka, fmaf(rs[mt][nt][r], a_out_scale, bs));
            else v = fmaf(acc[mt][nt][r], ka, bs);
            if (!(XP_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), out_r, out_voff + nt * 128, e_off(mt, r, a_out_stride), 0);
            vs += v;
            vq = fmaf(v, v, vq);
          }
        if (a_stats) {
          vs += __shfl_xor(vs, 32);
          vq += __shfl_xor(vq, 32);
          if (kh == 0) {
            red[(wave * NT * 32 + nt * 32 + p32) * 2 + 0] = vs;
            red[(wave * NT * 32 + nt * 32 + p32) * 2 + 1] = vq;
          }
        }
      }
      zero_acc();
      XP_TS(23);
      if (a_stats) {
        ff_barrier();
        if (tid < NT * 32) {
          double sm = 0.0, sq = 0.0;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) {
            sm += (double)red[(wv * NT * 32 + tid) * 2 + 0];
            sq += (double)red[(wv * NT * 32 + tid) * 2 + 1];
          }
          double* dst = a_stats + ((size_t)tc.tile * kCout + tc.ng * NT * 32 + tid) * 2;
          dst[0] = sm;
          dst[1] = sq;
        }
      }
    }
    XP_TS(24);
    tc = tn;
    gc = gn;
    tn = tile_at(it + 2);
  }
  XP_WALL(31);
}

template <bool NORM, bool RES>
static int launch_xp(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_xp_kernel<NORM, RES>;
  int dev = 0;
  CSD_CHECK_HIP(hipGetDevice(&dev));
  static int n_cu[64] = {0};
  if (dev < 0 || dev >= 64) dev = 0;
  if (!n_cu[dev]) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipDeviceProp_t prop;
    CSD_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    int n = prop.multiProcessorCount / 8 * 8;        // persistent: one workgroup per CU, a multiple of the 8 XCDs
    n_cu[dev] = n < 8 ? 8 : n;
  }
  const int grid = k.nblocks < n_cu[dev] ? (k.nblocks + 7) / 8 * 8 : n_cu[dev];
  hipLaunchKernelGGL(kern, dim3(grid), dim3(XP_THREADS), XPCfg::LDS, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// the fp16x3 (ns = 2) layers conv_ff covers with 96-cout groups and >= 2 stages; same packed weights, same arguments
bool convxp_supported(const ConvFFArgs& k, int nt) { return nt == 3 && k.nstage >= 2 && k.nstage % 2 == 0; }

int convxp_launch(const ConvFFArgs& k, hipStream_t s) {
  const bool norm = k.a.nscale != nullptr, res = k.a.res != nullptr;
  if (norm) return res ? launch_xp<true, true>(k, s) : launch_xp<true, false>(k, s);
  return res ? launch_xp<false, true>(k, s) : launch_xp<false, false>(k, s);
}

}  // namespace csd
