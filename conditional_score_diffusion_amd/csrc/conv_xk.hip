// conv_xk.hip - the fp16x3 (fp32-class) fused-prologue 3x3 convolution of the 160^2 / 80^2 / 40^2 levels
// (reference models/layers.py:119-132,632-675: h = Conv(act(GroupNorm(x))) [+ Dense(temb)] / x + Conv(...); models/layerspp.py:212-274)
// in the 1-D Winograd F(2,3) form along the image row, one transform component per wave.
//
// Arithmetic.  For an output pair (y0, y1) of a row, inputs d0..d3 and a filter row (g0, g1, g2):
//     D0 = d0 - d2   D1 = d1 + d2   D2 = d2 - d1   D3 = d1 - d3                       (input transform, fp32, BEFORE the hi | lo split)
//     G0 = g0        G1 = (g0 + g1 + g2) / 2       G2 = (g0 - g1 + g2) / 2   G3 = g2  (weights, transformed at pack time)
//     Mk = sum over input channels and the 3 filter ROWS of Dk * Gk                   (4 contractions instead of 6 per output pair)
//     y0 = M0 + M1 + M2      y1 = M1 - M2 - M3                                        (output transform in the epilogue)
// Every operand is carried as hi + lo fp16 (22 significand bits), products lo * hi, hi * hi, hi * lo on v_mfma_f32_32x32x16_f16 with fp32
// accumulation: 2 MFMAs per algorithmic product (the direct form: 3).  A 16-channel stage of a 16 x 16-pixel x 32 NT-cout tile is 3 row
// taps x 12 NT MFMAs.  Packed weights: [cout group][cin / 16][filter row][component][cout tile][hi | lo][64 lanes x 8 halves], x 2^8.
//
// Structure (the lineage conv_xp -> conv_xw -> conv_xk is in profiles/NOTEBOOK.md, rounds 4 - 5; the two predecessors were removed from
// the tree in round 6): ONE persistent 4-wave workgroup per CU (512 registers per lane), every wave ONE instruction stream
//     MFMA | 2 - 5 "filler" instructions | MFMA | ...
// with the fillers pinned in program order by scheduling fences - what hides behind a matrix instruction is the SAME wave's next few
// instructions (MI355X_MICROARCH.md: ~5 single-issue instructions per 32-cycle MFMA gap with one wave per SIMD).  The fillers are the
// conversion of the NEXT stage's operand patch (GroupNorm affine + exp2-domain SiLU + input transform + hi | lo split, on UNITS of two
// adjacent patch pixels x 4 channels with the right-hand neighbour's values by ds_bpermute; the float4s were requested two stages
// earlier), the ds_read_b128 of the next row tap's fragments, the weight requests and, in a tile's last stage, the epilogue's requests.
// The matrix instructions are asm statements (hipcc kept accumulators in VGPRs across joins and copied them through v_accvgpr_*; an asm
// MFMA is opaque to its hazard tracker, so the epilogue's reads sit behind a tied wait and tools/check_xp_isa.py checks every build).
//
// Ownership: WAVE k OWNS TRANSFORM COMPONENT k for all 128 pixel pairs of the tile (4 M tiles x NT cout tiles = 4 NT accumulators).  A wave
// then needs only its own component's weights: 2 NT KiB per row tap, straight from L2 into registers (three register sets, one per row
// tap, each refilled two row taps before its next use) - no weight staging through LDS (in the predecessor, where every wave multiplied
// all four components: 72 ds_write_b128 per stage into a three-slot ring and a workgroup barrier per row tap, 1.0 k of a stage's 6.05 k
// cycles for the stores alone), ONE barrier per stage (the patch double buffer; in front of the stage's last row tap, whose fragments are
// already in registers), 8 instead of 32 fragment reads per row tap.  The price is paid once per tile: the output transform needs the four
// components of a pixel pair in ONE lane, so the epilogue turns them through LDS - per row of a wave's 4 x 16-pixel block every wave
// stores its component of all four blocks (12 ds_write_b128 straight from the accumulator registers), one barrier, 12 ds_read_b128 of
// the four components of its own block; the rounds alternate between two LDS regions (the dead patch buffer + the space between the
// patch buffers | the top of LDS).
// Product order per row tap: lo * hi, hi * hi, hi * lo (the lo pixel fragments are free after the first product, the hi ones M tile by
// M tile under the last: both are re-read for the NEXT row tap inside the current one, so no fragment of a stage's patch is read after
// the barrier in front of its last row tap).
#include "conv_ff.h"

#include <utility>

namespace csd {

#define XW_THREADS 256
#define XW_RS (32 * 64 + 16)                         // LDS pitch of a transformed patch row: (component, pair) records of 64 B (16 ch hi | 16 ch lo)
#define XW_PATCH_BYTES (FF_PW * XW_RS)               // 37152
#define XW_NU 162                                    // conversion units per stage: 18 patch rows x 9 column pairs

template <int NT_>
struct XKCfg {
  static constexpr int NT = NT_;
  static constexpr int NA = 4 * NT;                          // accumulators per wave = MFMAs per product
  static constexpr int GP = 3 * NA;                          // MFMAs (= filler gaps) per row tap
  static constexpr int TAPB = 4 * NT * 2 * 1024;             // weight bytes per row tap: 4 components x NT cout tiles x (hi | lo)
  static constexpr int STB = 3 * TAPB;                       // per stage
  static constexpr int WPT = 2 * NT;                         // 1 KiB pieces per wave and row tap
  // LDS map (XTOG = the distance of the two patch buffers: a toggle is one xor): [0, 37152) patch 0 | exchange region X, blocks 2 and 3 | dummy | [65536, 102688) patch 1 |
  // exchange region Y, slots 0 .. 7 | the dummy's partner | [131072, ..) region Y, slots 8 .. 15 | red.  Region X's blocks 0 and 1 take
  // the patch buffer that is dead during the epilogue.  A slot = (destination block, source component): NT KiB.
  static constexpr int XTOG = 65536;
  static constexpr int SLOTB = NT * 1024;
  static constexpr int OFF_X23 = XW_PATCH_BYTES, OFF_DUMMY = OFF_X23 + 8 * SLOTB, OFF_Y0 = XTOG + XW_PATCH_BYTES, OFF_Y8 = 2 * XTOG;
  static constexpr int OFF_RED = OFF_Y8 + 8 * SLOTB;
  static constexpr size_t LDS = (size_t)OFF_RED + 4 * NT * 32 * 2 * sizeof(float);
  static_assert(OFF_DUMMY + 3072 <= XTOG && OFF_Y0 + 8 * SLOTB <= OFF_DUMMY + XTOG && OFF_DUMMY + XTOG + 3072 <= OFF_Y8 && 8 * SLOTB <= XW_PATCH_BYTES &&
                    LDS <= 160 * 1024,
                "conv_xk: LDS map");
};

template <class F, int... I>
__device__ __forceinline__ void xk_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void xk_static_for(F&& f) {
  xk_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef unsigned int xk_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xk_pack_f16(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(int, __builtin_convertvector(f2{a, b}, h2));
}
template <bool HIGH>
__device__ __forceinline__ float xk_lo(int hp, float v) {      // v - (float)half: one v_fma_mix_f32 (exact)
  float r;
  if constexpr (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}

#define XW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define XW_SADD(x, y) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(y) : "scc")
#define XW_PIN(a) asm volatile("" : "+v"(a))
// tuning aids (never in the product library): XK_ABL bits remove parts of the stream at compile time (results are then garbage):
// 1 conversion, 2 weight requests, 4 fragment reads, 8 output stores, 16 patch requests, 32 residual requests, 64 the stage barriers,
// 128 the epilogue's exchange stores, 65536 its barriers, 131072 its reads (own accumulators instead), 262144 its statistics,
// 524288 the whole row loop of the epilogue (what remains is the tile switch),
// 256 patch stores, 512 neighbour exchange (own value instead), 1024 transcendentals (plain multiplies instead), 2048 hi | lo split,
// 4096 patch requests confined to the first 256 pixels of the sample (cache hits), 8192 the same bytes as whole 1 KiB pieces
// (tools/xk_abl_build.sh builds the libraries, tools/xk_timing.py reads the per-tile stamps)
#ifndef XK_ABL
#define XK_ABL 0
#endif
// Tuning-build stamps sit at TILE boundaries only (the loop top: no accumulator is live there).  A stamp is a branch; between a tile's
// first MFMA and its epilogue's last accumulator read a control-flow edge lets hipcc move accumulators - unprotected reads of matrix
// results: per-unit stamps made the tuning build return NaN, and tools/check_xp_isa.py now checks the tuning build as well.
#ifdef CSD_FF_TUNE
#define XW_WALL(i) do { if (a_dbg && tid == 0) a_dbg[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define XW_WALL(i) do { } while (0)
#endif

// ---- the filler schedule of a row tap (compile time) ----
// MFMA g of a row tap: product g / PB (0 lo * hi, 1 hi * hi, 2 hi * lo), M tile (g % PB) / NT, cout tile g % NT.  Fixed fillers of gap g:
template <int NT>
struct XKSched {
  static constexpr int PB = 4 * NT, GP = 3 * PB, WPT = 2 * NT;
  static constexpr bool rd_xl(int g) { return g < PB && g % NT == NT - 1; }                          // xl[g / NT] of the NEXT tap (its last use was MFMA g)
  static constexpr bool rd_xh(int g) { return g >= 2 * PB && g % NT == NT - 1; }                     // xh[(g - 2 PB) / NT] of the next tap
  static constexpr bool slot_req(int g) { return g == 0; }                                           // the tap's two patch requests
  static constexpr bool ld_w(int g) { return g >= PB && g < PB + WPT; }                              // weight fragment g - PB of the tap two ahead
  static constexpr int fixed(int g) { return (rd_xl(g) ? 1 : 0) + (rd_xh(g) ? 1 : 0) + (ld_w(g) ? 1 : 0) + (slot_req(g) ? 3 : 0); }
  static constexpr int wgt(int g) { return 14 - 3 * fixed(g) > 2 ? 14 - 3 * fixed(g) : 2; }
  static constexpr int cum(int g) { int s = 0; for (int h = 0; h < g; ++h) s += wgt(h); return s; }
  static constexpr int first_op(int n, int g) { return (int)(((long long)n * cum(g)) / cum(GP)); }    // ops [first_op(n, g), first_op(n, g + 1)) in gap g
};

template <int NT_, bool NORM, bool RES>
__global__ __launch_bounds__(XW_THREADS, 1) void conv_xk_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = XKCfg<NT_>;
  using S = XKSched<NT_>;
  constexpr int NT = C::NT, NA = C::NA, GP = C::GP, TAPB = C::TAPB, STB = C::STB, WPT = C::WPT, PB = S::PB;
  constexpr int OFF_RED = C::OFF_RED, OFF_DUMMY = C::OFF_DUMMY, XTOG = C::XTOG, SLOTB = C::SLOTB;
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
  const int kh = lane >> 5, p32 = lane & 31, lg = lane & 3, ul = lane >> 2;
#ifdef CSD_FF_TUNE
  long long* const a_dbg = k.a.dbg;
#endif
  XW_WALL(30);

  // ---- this workgroup's tiles: workgroup p (one per CU, on XCD p % 8) takes tiles wj, wj + P/8, ... of its XCD's share ----
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

  // Per-lane values that only the once-per-tile code needs are recomputed there from a FRESH lane id (two instructions): kept in
  // registers across the stream they would be spilled - and a reload from scratch memory waits for every outstanding request.
  // (an asm volatile statement on the lane id makes what is derived from it opaque: hipcc can neither hoist it out of the tile loop
  // nor merge it with the prologue's copy of the same arithmetic)
  auto fresh_lane = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  constexpr float NLOG2E = -1.4426950408889634f;
  // out = y * ka + (residual + bias + temb) * out_scale; the staged operand is u / (1 + 2^u), u = -log2(e) (x s + t): SiLU = -ln2 * that
  const float ka = (NORM ? -0.6931471805599453f / C16_WSCALE : 1.0f / C16_WSCALE) * a_out_scale;

  // ---- conversion units ----
  // slot j of wave w: units [15 q, 15 q + 16), q = 4 j + w, one per lane quad (ul), 4-channel group lg; unit U = patch row U / 9, patch
  // columns 2 (U % 9), + 1.  A unit with U % 9 < 8 produces the pair U % 9 of its row from its own values and those of unit U + 1 (the
  // lanes + 4); the last quad of a slot only provides (the next slot repeats it), units >= 162 do not exist.
  int s_dst[3];                                      // LDS byte offset of the unit's record (component 0, hi half) in the buffer being WRITTEN
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int U = 15 * (4 * j + wave) + ul;
    const int Uc = U < XW_NU ? U : XW_NU - 1;
    const int pr = Uc / 9, u = Uc - pr * 9;
    const bool writer = U < XW_NU && ul < 15 && u < 8;
    const int rec = pr * XW_RS + u * 64 + lg * 8;
    const int dummy = OFF_DUMMY + ul * 64 + lg * 8;  // (non-writers store their garbage to a scratch area: no exec juggling in the stream)
    s_dst[j] = writer ? rec : dummy;                 // (the prologue writes buffer 0; ^= XTOG at the end of every unit)
  }
  // the two pixels (a, b = a + 1 in the row) of a unit in ONE register: bits 0-23 the index inside the sample of a (of b when only b
  // exists, 0 when neither does - what is read there is masked away: zero padding), bit 31: a exists, bit 30: b exists, bit 29: both
  // (then b's index is a's + 1).  The 24-bit multiply of the request ignores the flags.
  struct Geom { int p[3]; };
  auto geom_of = [&](const Tile& t) __attribute__((always_inline)) {
    Geom g;
    const int ulf = fresh_lane() >> 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int U = 15 * (4 * j + wave) + ulf;
      const int Uc = U < XW_NU ? U : XW_NU - 1;
      const int pr = Uc / 9, u = Uc - pr * 9;
      const int y = t.ty0 - 1 + pr, xa = t.tx0 - 1 + 2 * u;
      const bool iny = (unsigned)y < (unsigned)kH && U < XW_NU;
      const bool ina = iny && (unsigned)xa < (unsigned)kW, inb = iny && (unsigned)(xa + 1) < (unsigned)kW;      // zero padding outside THIS sample
      g.p[j] = (ina ? y * kW + xa : inb ? y * kW + xa + 1 : 0) | (ina ? (int)0x80000000 : 0) | (inb ? 0x40000000 : 0) |
               (ina && inb ? 0x20000000 : 0);
    }
    return g;
  };
  const int nb_addr = ((lane + 4) & 63) * 4;         // ds_bpermute address of the neighbour unit's lane

  // fragment base: pixel pairs (M operand) of M tile m: rows 4 m + (p32 >> 3), pair p32 & 7, K half kh, THIS wave's component
  const int xbase = (p32 >> 3) * XW_RS + (p32 & 7) * 64 + kh * 16 + wave * 512;
  int xb_cur = xbase, xb_nxt = xbase + XTOG;         // ... in the buffer of the stage being multiplied | of the next stage

  // ---- requests ----
  xk_u4 pfa[3], pfb[3];                              // the raw float4 of each slot's two pixels
  xk_u4 scn, shn;                                    // GroupNorm scale / shift of the lane's 4 channels
  float msc[4], msh[4];
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
  auto req_a = [&](int j, const Geom& g) __attribute__((always_inline)) {
    if (!(XK_ABL & 16)) pfa[j] = __builtin_amdgcn_raw_buffer_load_b128(srcL, (XK_ABL & 8192) ? (unsigned)(tid * 16 + j * 8192) : __umul24((XK_ABL & 4096) ? (g.p[j] & 255) : g.p[j], strideL) + lg * 16, soffL, 0);
  };
  auto req_b = [&](int j, const Geom& g) __attribute__((always_inline)) {
    if (!(XK_ABL & 16)) pfb[j] = __builtin_amdgcn_raw_buffer_load_b128(srcL, (XK_ABL & 8192) ? (unsigned)(tid * 16 + j * 8192 + 4096) : __umul24(((XK_ABL & 4096) ? (g.p[j] & 255) : g.p[j]) + ((g.p[j] >> 29) & 1), strideL) + lg * 16, soffL, 0);
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
  auto cvt_prep_half = [&](int h) __attribute__((always_inline)) {       // the affine in the exp2 domain
    if constexpr (NORM) {
#pragma unroll
      for (int q = 2 * h; q < 2 * h + 2; ++q) {
        msc[q] = __uint_as_float(scn[q]) * NLOG2E;
        msh[q] = __uint_as_float(shn[q]) * NLOG2E;
      }
      asm volatile("" : "+v"(msc[2 * h]), "+v"(msc[2 * h + 1]), "+v"(msh[2 * h]), "+v"(msh[2 * h + 1]));
    }
  };
  // weights: the packed tensor is one linear stream per (cout group, tile): stage after stage, row tap after row tap, and inside a row
  // tap [component][cout tile][hi | lo][64 lanes x 16 B]: this wave reads its component's WPT KiB.  The scalar offset runs (asm add) and is
  // re-based when the stream moves to the next tile.
  const __amdgpu_buffer_rsrc_t w_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g_wpack), 0, OOB, RSRC_FLAGS);
  const int wvoff = lane * 16 + wave * (WPT * 1024);
  int w_run = 0;
  const int c_tapb = TAPB;
  auto w_begin = [&](int wso) __attribute__((always_inline)) { w_run = wso; };
  auto w_next_tap = [&]() __attribute__((always_inline)) { XW_SADD(w_run, c_tapb); };

  // ---- conversion of a slot, as a sequence of single operations (placed one by one between the MFMAs) ----
  // "pre": per value (8 = pixels a, b x 4 channels) the chain affine > exp2 > 1 + > rcp > u * > zero padding > neighbour value; emitted
  // along the diagonals of the (value, phase) table, so that a dependent operation sits 7 operations behind its producer and the
  // transcendentals (12.8 cycles of issue each, two per MFMA gap ride free - tools/probes/filler_cost_probe.hip) are spread out.
  // "post", per half of the channels (2 of the lane's 4): the four components (8) | hi pack (4) | lo (8) | lo pack (4).
  // Without the GroupNorm prologue pre = padding + neighbour values only.
  constexpr int NPH = NORM ? 7 : 2, NPRE = 8 * NPH, NPOST = 24, NMATH = NPRE + 2 * NPOST;
  struct PreOrder {
    int val[8 * 7], ph[8 * 7];
    constexpr PreOrder(int nph) : val{}, ph{} {
      int n = 0;
      for (int d = 0; d < nph + 7; ++d)
        for (int i = 0; i < 8; ++i)
          if (d - i >= 0 && d - i < nph) { val[n] = i; ph[n] = d - i; ++n; }
    }
  };
  constexpr PreOrder PRE(NPH);
  float cv[8], ce[8], cn[8], cd[4][2], cl[4][2];     // values: a0..a3 b0..b3 (index = 4 * (b) + channel); [component][channel of the half]
  int chp[4][2], clp[4][2];                          // packed hi | lo of [component][half]
  auto math_op = [&](auto j_tag, auto o_tag, const Geom& g) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, ol = decltype(o_tag)::value;
    if (XK_ABL & 1) return;
    if constexpr (ol < NPRE) {
      constexpr int i = PRE.val[ol], c = i & 3, o = PRE.ph[ol] + (NORM ? 0 : 5);      // value i (a: i < 4), channel c, phase o
      if constexpr (o == 0) {
        const float x = __uint_as_float(i < 4 ? pfa[j][c] : pfb[j][c]);
        cv[i] = fmaf(x, msc[c], msh[c]);
        XW_PIN(cv[i]);
      } else if constexpr (o == 1) {
        ce[i] = (XK_ABL & 1024) ? cv[i] * 1.5f : __builtin_amdgcn_exp2f(cv[i]);
        XW_PIN(ce[i]);
      } else if constexpr (o == 2) {
        ce[i] = 1.0f + ce[i];
        XW_PIN(ce[i]);
      } else if constexpr (o == 3) {
        ce[i] = (XK_ABL & 1024) ? ce[i] * 0.7f : __builtin_amdgcn_rcpf(ce[i]);
        XW_PIN(ce[i]);
      } else if constexpr (o == 4) {
        cv[i] = cv[i] * ce[i];
        XW_PIN(cv[i]);
      } else if constexpr (o == 5) {                 // padding applies to the ACTIVATED tensor: exactly 0
        const int m = (i < 4 ? g.p[j] : g.p[j] << 1) >> 31;
        const int raw = NORM ? __float_as_int(cv[i]) : (int)(i < 4 ? pfa[j][c] : pfb[j][c]);
        cv[i] = __int_as_float(raw & m);
        XW_PIN(cv[i]);
      } else {                                       // the neighbour unit's value (not pinned: the wait belongs in front of its use)
        if (XK_ABL & 512) { cn[i] = cv[i] * 0.5f; XW_PIN(cn[i]); }
        else cn[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(nb_addr, __float_as_int(cv[i])));
      }
    } else {
      constexpr int hh = (ol - NPRE) / NPOST, o = (ol - NPRE) % NPOST;      // the half: channels 2 hh, 2 hh + 1
      if constexpr (o < 8) {
        constexpr int kc = o >> 1, c = o & 1, ca = 2 * hh + c, cb = 4 + ca;
        if constexpr (kc == 0) cd[0][c] = cv[ca] - cn[ca];               // d0 - d2
        else if constexpr (kc == 1) cd[1][c] = cv[cb] + cn[ca];          // d1 + d2
        else if constexpr (kc == 2) cd[2][c] = cn[ca] - cv[cb];          // d2 - d1
        else cd[3][c] = cv[cb] - cn[cb];                                 // d1 - d3
        XW_PIN(cd[kc][c]);
      } else if constexpr (o < 12) {
        constexpr int kc = o - 8;
        if (XK_ABL & 2048) { chp[kc][hh] = __float_as_int(cd[kc][0]); clp[kc][hh] = __float_as_int(cd[kc][1]); return; }
        chp[kc][hh] = xk_pack_f16(cd[kc][0], cd[kc][1]);
        XW_PIN(chp[kc][hh]);
      } else if constexpr (o < 20) {
        if (XK_ABL & 2048) return;
        constexpr int kc = (o - 12) >> 1, c = (o - 12) & 1;
        if constexpr (c == 0) cl[kc][0] = xk_lo<false>(chp[kc][hh], cd[kc][0]);
        else cl[kc][1] = xk_lo<true>(chp[kc][hh], cd[kc][1]);
        XW_PIN(cl[kc][c]);
      } else {
        if (XK_ABL & 2048) return;
        constexpr int kc = o - 20;
        clp[kc][hh] = xk_pack_f16(cl[kc][0], cl[kc][1]);
        XW_PIN(clp[kc][hh]);
      }
    }
  };
  // the eight stores of a slot: component kc hi (o = kc), lo (o = 4 + kc), into the patch buffer `par1` (0 | 1)
  auto write_op = [&](auto j_tag, auto o_tag) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, o = decltype(o_tag)::value;
    if (XK_ABL & (1 | 256)) return;
    constexpr int kc = o & 3, pl = o >> 2;
    char* const rec = smem + s_dst[j] + kc * 512 + pl * 32;
    if constexpr (pl == 0) *reinterpret_cast<int2*>(rec) = make_int2(chp[kc][0], chp[kc][1]);
    else *reinterpret_cast<int2*>(rec) = make_int2(clp[kc][0], clp[kc][1]);
  };
  auto reqs_op = [&](auto j_tag, auto o_tag, const Geom& g) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, o = decltype(o_tag)::value;
    if constexpr (o == 0) req_a(j, g);
    else req_b(j, g);
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  // a whole slot: math, stores, the requests of the same slot for the stage after
  constexpr int NFULL = NMATH + 8;
  auto full_op = [&](auto j_tag, auto o_tag, const Geom& gm, const Geom& gr) __attribute__((always_inline)) {
    constexpr int o = decltype(o_tag)::value;
    if constexpr (o < NMATH) math_op(j_tag, o_tag, gm);
    else write_op(j_tag, std::integral_constant<int, o - NMATH>{});
  };

  // ---- fragments ----
  // pixel pairs of M tile m (this wave's component): xh / xl, single-buffered, re-read for the next row tap as soon as their last MFMA of
  // this one has been issued; weights: one register set per row tap index (cout tile nt: hi = 2 nt, lo = 2 nt + 1), set 2 in the
  // accumulator half of the register file next to xh / xl (the vector half keeps the conversion, the requests and sets 0 and 1)
  half8 xh[4], xl[4], w0[WPT], w1[WPT], w2[WPT];
  auto rd_x = [&](half8& dst, int xb, int r, int m, int pl) __attribute__((always_inline)) {
    if (XK_ABL & 4) return;
    dst = *reinterpret_cast<const half8*>(smem + xb + (4 * m + r) * XW_RS + pl * 32);
  };
  auto ld_w = [&](half8& dst, int i) __attribute__((always_inline)) {      // fragment i of the row tap the stream offset points at
    if (XK_ABL & 2) return;
    dst = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_r, (unsigned)(wvoff + i * 1024), w_run, 0));
  };

  // The accumulators live in the accumulator half of the register file for the whole kernel; the matrix instructions are asm
  // statements (header); a tile's first product writes them with C = 0, so nothing is live across the tile loop's
  // back edge, and the epilogue's reads sit behind tie_acc_done().  tools/check_xp_isa.py checks the generated code.
  floatx16 acc[4][NT];
#define XK_MFMA_AV(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "a"(A), "v"(B))
#define XK_MFMA_AA(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "a"(A), "a"(B))
  // product p of a row tap: 0 lo * hi, 1 hi * hi, 2 hi * lo; one lambda per row tap index = weight register set (non-generic lambdas:
  // an asm statement inside a generic one cannot name the enclosing function's arrays)
  auto mm0 = [&](int p, int i) __attribute__((always_inline)) {
    const int m = i / NT, nt = i - m * NT, wi = 2 * nt + (p == 2 ? 1 : 0);
    if (p == 0) XK_MFMA_AV(acc[m][nt], xl[m], w0[wi]); else XK_MFMA_AV(acc[m][nt], xh[m], w0[wi]);
  };
  auto mm1 = [&](int p, int i) __attribute__((always_inline)) {
    const int m = i / NT, nt = i - m * NT, wi = 2 * nt + (p == 2 ? 1 : 0);
    if (p == 0) XK_MFMA_AV(acc[m][nt], xl[m], w1[wi]); else XK_MFMA_AV(acc[m][nt], xh[m], w1[wi]);
  };
  auto mm2 = [&](int p, int i) __attribute__((always_inline)) {
    const int m = i / NT, nt = i - m * NT, wi = 2 * nt + (p == 2 ? 1 : 0);
    if (p == 0) XK_MFMA_AV(acc[m][nt], xl[m], w2[wi]); else XK_MFMA_AV(acc[m][nt], xh[m], w2[wi]);
  };
  auto mm_first = [&](int i) __attribute__((always_inline)) {      // a tile's first product (row tap 0, lo * hi): C = 0
    const int m = i / NT, nt = i - m * NT;
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc[m][nt]) : "a"(xl[m]), "v"(w0[2 * nt]));
  };
  auto tie_acc_done = [&]() __attribute__((always_inline)) {
    static_assert(NT == 3 || NT == 2, "conv_xk: 96- or 64-cout groups");
    if constexpr (NT == 3)
      asm volatile("s_nop 15\n\ts_nop 7"
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[2][0]),
                     "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]));
    else
      asm volatile("s_nop 15\n\ts_nop 7"
                   : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[3][0]),
                     "+a"(acc[3][1]));
  };

  // ---- epilogue operands of a tile ----
  // Element r of accumulator (k, nt): pair (row 4 wave + r / 4, pair 4 kh + r % 4), cout nt * 32 + p32.  The lane part of an address
  // (kh, cout, the wave's first row) is one voffset per tensor, nt * 128 an immediate, and the (row, column) part a RUNNING scalar
  // offset: the lane's 8 columns of a row are consecutive pixels, then one step to the next row.
  float rs[4][8][NT];                                // residual: [row][column][cout tile]
  float bv[NT], tv[NT];
  __amdgpu_buffer_rsrc_t res_r, out_r;
  unsigned res_voff = 0, out_voff = 0;
  bool e_valid = true;
  int e_tile = 0, e_ng = 0;
  const int res_col = kCout * 4, res_row = (kW - 7) * kCout * 4, out_col = a_out_stride * 4, out_row = (kW - 7) * a_out_stride * 4;
  int res_run = 0;
  const __amdgpu_buffer_rsrc_t bias_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_bias ? a_bias : a_src0), 0, a_bias ? OOB : 0u, RSRC_FLAGS);
  auto epi_setup = [&](const Tile& t) __attribute__((always_inline)) {      // scalars only
    const size_t tile_pix = (size_t)t.b * kH * kW + (size_t)t.ty0 * kW + t.tx0;
    out_r = __builtin_amdgcn_make_buffer_rsrc(a_out + tile_pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
    e_tile = t.tile; e_ng = t.ng;
    if constexpr (RES) {
      res_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_res + tile_pix * kCout), 0, OOB, RSRC_FLAGS);
      res_run = 0;
    }
  };
  auto epi_loads = [&](const Tile& t) __attribute__((always_inline)) {      // under the tile's last tap: the per-lane offsets, bias, temb
    const int lf = fresh_lane();
    const int c_lane = t.ng * NT * 32 + (lf & 31);
    // (ragged tiles - maps of 8 k x 8 k pixels that 16 does not divide: a lane's 4 rows x 8 columns lie inside the image together or not
    // at all; outside, its stores and residual loads go out of the descriptor's range and its statistics are dropped)
    e_valid = (t.ty0 + 4 * wave < kH) && (t.tx0 + 8 * (lf >> 5) < kW);
    out_voff = e_valid ? (unsigned)((4 * wave * kW + 8 * (lf >> 5)) * a_out_stride + c_lane) * 4u : OOB;
    if constexpr (RES) res_voff = e_valid ? (unsigned)((4 * wave * kW + 8 * (lf >> 5)) * kCout + c_lane) * 4u : OOB;
    // (a null bias / temb reads as zeros through an empty descriptor: no branch in the stream)
    const __amdgpu_buffer_rsrc_t tb_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_temb ? a_temb + (size_t)t.b * a_temb_stride : a_src0), 0,
                                                                           a_temb ? OOB : 0u, RSRC_FLAGS);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bv[nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bias_r, (unsigned)(c_lane + nt * 32) * 4u, 0, 0));
      tv[nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(tb_r, (unsigned)(c_lane + nt * 32) * 4u, 0, 0));
    }
  };
  auto req_res = [&](int e) __attribute__((always_inline)) {      // e = (row * 8 + column) * NT + nt, in order
    if constexpr (RES) {
      if (XK_ABL & 32) return;
      const int idx = e / NT, nt = e - idx * NT;
      rs[idx >> 3][idx & 7][nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, res_voff + nt * 128, res_run, 0));
      if (nt == NT - 1 && idx < 31) {
        if ((idx & 7) != 7) XW_SADD(res_run, res_col);
        else XW_SADD(res_run, res_row);
      }
    }
  };
  float* const red = reinterpret_cast<float*>(smem + OFF_RED);      // [4 waves][NT*32 couts][2]: statistics hand-over
  // the component exchange: slot (destination block d, source component c) of a region, NT KiB each, lane * 16 inside a cout tile's KiB.
  // Region X: blocks 0, 1 in the patch buffer that is dead during the epilogue (`dead`: 0 | XTOG), blocks 2, 3 at OFF_X23; region Y: slots
  // 0 .. 7 at OFF_Y0, 8 .. 15 at OFF_Y8.
  int ex_dead = XTOG;                                // scalar: base of the patch buffer xb_nxt lives in (toggles with it): the dead one at a tile's end
  auto epilogue = [&]() __attribute__((always_inline)) {
    if constexpr (RES) {                             // rows 1 .. 3 of the residual (row 0 came in under the last tap)
#pragma unroll
      for (int e = 8 * NT; e < 32 * NT; ++e) req_res(e);
    }
    tie_acc_done();
    const int lf = fresh_lane();
    const int l16 = lf * 16;
    // send side: slot (d, wave); receive side: slots (wave, c)
    const int txX01 = ex_dead + wave * SLOTB + l16, txX23 = C::OFF_X23 + wave * SLOTB + l16;
    const int txY0 = C::OFF_Y0 + wave * SLOTB + l16, txY8 = C::OFF_Y8 + wave * SLOTB + l16;
    const int rxX = (wave < 2 ? ex_dead + wave * 4 * SLOTB : C::OFF_X23 + (wave - 2) * 4 * SLOTB) + l16;
    const int rxY = (wave < 2 ? C::OFF_Y0 + wave * 4 * SLOTB : C::OFF_Y8 + (wave - 2) * 4 * SLOTB) + l16;
    // statistics: four independent partial sums per cout tile (even | odd column x even | odd pair) - one wave per SIMD has nothing to
    // cover the latency of a 32-long dependent chain with; folded in a fixed order at the end
    float ps[NT][4], pq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) ps[nt][q] = pq[nt][q] = 0.f;
    int out_run = 0;
    typedef float xk_f4 __attribute__((ext_vector_type(4)));
    auto send_block = [&](int row, int d) __attribute__((always_inline)) {
      const int tx = (row & 1) ? ((d < 2 ? txY0 : txY8) + (d & 1) * 4 * SLOTB) : ((d < 2 ? txX01 : txX23) + (d & 1) * 4 * SLOTB);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (!(XK_ABL & 128))
          *reinterpret_cast<xk_f4*>(smem + tx + nt * 1024) =
              xk_f4{acc[d][nt][4 * row], acc[d][nt][4 * row + 1], acc[d][nt][4 * row + 2], acc[d][nt][4 * row + 3]};
    };
#pragma unroll
    for (int row = 0; row < ((XK_ABL & 524288) ? 0 : 4); ++row) {
      // every wave's component of row `row` of all four blocks -> LDS: row 0 here, rows 1 .. 3 block by block between the pairs of the
      // row before (into the other region): three stores at a time do not fill the wave's LDS queue, and the pipe works on them
      // while the vector pipe does the arithmetic
      if (row == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d) send_block(0, d);
      }
      if (!(XK_ABL & 65536)) ff_barrier();
      xk_f4 mv[4][NT];                               // [component][cout tile]: the four pairs of this row
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (XK_ABL & 131072) mv[c][nt] = xk_f4{acc[c][nt][4 * row], acc[c][nt][4 * row + 1], acc[c][nt][4 * row + 2], acc[c][nt][4 * row + 3]};
          else mv[c][nt] = *reinterpret_cast<const xk_f4*>(smem + ((row & 1) ? rxY : rxX) + c * SLOTB + nt * 1024);
        }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float y1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bs = (bv[nt] + tv[nt]) * a_out_scale;
          const float m0 = mv[0][nt][jj], m1 = mv[1][nt][jj], m2 = mv[2][nt][jj], m3 = mv[3][nt][jj];
          const float y0 = (m0 + m1) + m2;
          y1[nt] = (m1 - m2) - m3;
          float v;
          if constexpr (RES) v = fmaf(y0, ka, fmaf(rs[row][2 * jj][nt], a_out_scale, bs));
          else v = fmaf(y0, ka, bs);
          if (!(XK_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), out_r, out_voff + nt * 128, out_run, 0);
          if (!(XK_ABL & 262144)) { ps[nt][(jj & 1) * 2] += v; pq[nt][(jj & 1) * 2] = fmaf(v, v, pq[nt][(jj & 1) * 2]); }
        }
        XW_SADD(out_run, out_col);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bs = (bv[nt] + tv[nt]) * a_out_scale;
          float v;
          if constexpr (RES) v = fmaf(y1[nt], ka, fmaf(rs[row][2 * jj + 1][nt], a_out_scale, bs));
          else v = fmaf(y1[nt], ka, bs);
          if (!(XK_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), out_r, out_voff + nt * 128, out_run, 0);
          if (!(XK_ABL & 262144)) { ps[nt][(jj & 1) * 2 + 1] += v; pq[nt][(jj & 1) * 2 + 1] = fmaf(v, v, pq[nt][(jj & 1) * 2 + 1]); }
        }
        if (jj != 3) XW_SADD(out_run, out_col);
        else if (row != 3) XW_SADD(out_run, out_row);
        XW_FENCE();
        if (row < 3) send_block(row + 1, jj);
        XW_FENCE();
      }
    }
    if (a_stats) {
      const int tidf = wave * 64 + lf;
      float vs[NT], vq[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        vs[nt] = (ps[nt][0] + ps[nt][1]) + (ps[nt][2] + ps[nt][3]);
        vq[nt] = (pq[nt][0] + pq[nt][1]) + (pq[nt][2] + pq[nt][3]);
        vs[nt] = e_valid ? vs[nt] : 0.f;
        vq[nt] = e_valid ? vq[nt] : 0.f;
        vs[nt] += __shfl_xor(vs[nt], 32);
        vq[nt] += __shfl_xor(vq[nt], 32);
        if (lf < 32) {
          red[(wave * NT * 32 + nt * 32 + lf) * 2 + 0] = vs[nt];
          red[(wave * NT * 32 + nt * 32 + lf) * 2 + 1] = vq[nt];
        }
      }
      ff_barrier();
      if (tidf < NT * 32) {
        double sm = 0.0, sq = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          sm += (double)red[(wv * NT * 32 + tidf) * 2 + 0];
          sq += (double)red[(wv * NT * 32 + tidf) * 2 + 1];
        }
        double* dst = a_stats + ((size_t)e_tile * kCout + e_ng * NT * 32 + tidf) * 2;
        dst[0] = sm;
        dst[1] = sq;
      }
    }
  };

  Tile tc = tile_at(0), tn = tile_at(1);
  Geom gc = geom_of(tc), gn = gc;
  const int NSTB = (Cin / 16) * STB;                 // packed bytes of one cout group

  // =========================================================================================================================
  // prologue: stage 0 converted, slot 0 of stage 1 computed and held, slots 1 / 2 of stage 1 and slot 0 of stage 2 requested per
  // the stream's own rules, the weights of row taps 0 and 1 requested into their register sets, the first fragments read
  // =========================================================================================================================
  src_of(tc, 0);
#pragma unroll
  for (int j = 0; j < 3; ++j) { req_a(j, gc); req_b(j, gc); }
  req_norm(tc, 0);
  w_begin(tc.ng * NSTB);
  {
    cvt_prep_half(0); cvt_prep_half(1);
    req_norm(tc, 1);
    src_of(tc, 1);
    // stage 0, all three slots, into buffer 0; each slot re-requested for stage 1
    xk_static_for<NFULL>([&](auto o) __attribute__((always_inline)) { full_op(J0{}, o, gc, gc); });
    req_a(0, gc); req_b(0, gc);
    xk_static_for<NFULL>([&](auto o) __attribute__((always_inline)) { full_op(J1{}, o, gc, gc); });
    req_a(1, gc); req_b(1, gc);
    xk_static_for<NFULL>([&](auto o) __attribute__((always_inline)) { full_op(J2{}, o, gc, gc); });
    req_a(2, gc); req_b(2, gc);
#pragma unroll
    for (int j = 0; j < 3; ++j) s_dst[j] ^= XTOG;          // from here on the stream writes buffer 1 (stage 1)
    // stage 1, slot 0: computed and held (stored by the first row tap)
    cvt_prep_half(0); cvt_prep_half(1);
    xk_static_for<NMATH>([&](auto o) __attribute__((always_inline)) { math_op(J0{}, o, gc); });
    // the weights of row taps 0 and 1 of stage 0 (row tap 0 of the first unit requests row tap 2)
#pragma unroll
    for (int i = 0; i < WPT; ++i) ld_w(w0[i], i);
    w_next_tap();
#pragma unroll
    for (int i = 0; i < WPT; ++i) ld_w(w1[i], i);
    w_next_tap();
  }
  ff_barrier();
#pragma unroll
  for (int m = 0; m < 4; ++m) { rd_x(xh[m], xb_cur, 0, m, 0); rd_x(xl[m], xb_cur, 0, m, 1); }

  // =========================================================================================================================
  // One unit = the three row taps of stage s of tile tc; meanwhile stage X = s + 1 is converted into the other patch buffer, stage L =
  // s + 2 requested, and row tap r requests the weights of the row tap two ahead into the register set that row tap r - 1 has just left.
  //   tap 0: [store slot 0 of X (held)] [request slot 0 of L] [slot 1 of X: math, stores, request for L]   + the stage's scalar set-up
  //   tap 1: [slot 2 of X: math, stores, request for L]
  //   ---- barrier: X's patch is complete; every fragment of THIS stage's patch is already in registers ----
  //   tap 2: [the affine of L] [slot 0 of L: math, held]                 (reads X's patch for the next stage's first row tap)
  // POS 0: s + 2 < NS; POS 1: s = NS - 2 (L = stage 0 of the next tile); POS 2: s = NS - 1 (X = stage 0, L = stage 1 of the next
  // tile; the residual's first row rides along).  FIRST: stage 0 of a tile (its first product starts the accumulators).
  // =========================================================================================================================
  auto unit = [&](auto pos_tag, auto first_tag, int s) __attribute__((always_inline)) {
    constexpr int POS = decltype(pos_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    const Tile& tx = POS == 2 ? tn : tc;             // tile of stage X
    const Tile& tl = POS >= 1 ? tn : tc;             // tile of stage L
    const Geom& gx = POS == 2 ? gn : gc;
    const Geom& gl = POS >= 1 ? gn : gc;
    const int sl = POS == 0 ? s + 2 : POS - 1;
    xk_static_for<3>([&](auto r_tag) __attribute__((always_inline)) {
      constexpr int R = decltype(r_tag)::value;
      XW_FENCE();
      if constexpr (R == 2) { if (!(XK_ABL & 64)) ff_barrier(); }
      XW_FENCE();
      if constexpr (R == 0) src_of(tl, sl);          // (scalar: the descriptor of stage L's requests, used from this tap's gap 0 on)
      if constexpr (R == 1 && POS == 2) w_begin(tn.ng * NSTB);      // (this tap's weight requests are the next tile's first row tap)
      constexpr int NOPS = R == 0 ? 8 + NFULL : R == 1 ? NFULL : 2 + NMATH;
      xk_static_for<GP>([&](auto g_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value;
        XW_FENCE();
        if constexpr (FIRST && R == 0 && g < NA) mm_first(g);
        else if constexpr (R == 0) mm0(g / NA, g % NA);
        else if constexpr (R == 1) mm1(g / NA, g % NA);
        else mm2(g / NA, g % NA);
        XW_FENCE();
        // ---- fixed fillers of gap (R, g) ----
        if constexpr (S::rd_xl(g)) rd_x(xl[g / NT], R == 2 ? xb_nxt : xb_cur, (R + 1) % 3, g / NT, 1);
        if constexpr (S::rd_xh(g)) rd_x(xh[(g - 2 * PB) / NT], R == 2 ? xb_nxt : xb_cur, (R + 1) % 3, (g - 2 * PB) / NT, 0);
        if constexpr (S::slot_req(g)) { req_a(R, gl); req_b(R, gl); }      // slot R for stage L (consumed two taps from now)
        if constexpr (S::ld_w(g)) {                  // the row tap two ahead, into the set of the row tap before this one
          if constexpr (R == 0) ld_w(w2[g - PB], g - PB);
          else if constexpr (R == 1) ld_w(w0[g - PB], g - PB);
          else ld_w(w1[g - PB], g - PB);
          if constexpr (g == PB + WPT - 1) w_next_tap();
        }
        // ---- scalar set-up of the stage, in the first gaps of tap 0 ----
        if constexpr (R == 0 && g == 2 && POS == 1) epi_setup(tc);
        if constexpr (R == 1 && g == PB + PB / 2) req_norm(tl, sl);                     // (used by tap 2's first operations)
        if constexpr (R == 2 && POS == 2 && g == PB) epi_loads(tc);
        // ---- the conversion's operations of this gap ----
        constexpr int o0 = S::first_op(NOPS, g), o1 = S::first_op(NOPS, g + 1);
        xk_static_for<o1 - o0>([&](auto d_tag) __attribute__((always_inline)) {
          constexpr int o = o0 + decltype(d_tag)::value;
          if constexpr (R == 0) {
            if constexpr (o < 8) write_op(J0{}, std::integral_constant<int, o>{});
            else full_op(J1{}, std::integral_constant<int, o - 8>{}, gx, gl);
          } else if constexpr (R == 1) {
            full_op(J2{}, std::integral_constant<int, o>{}, gx, gl);
          } else {
            if constexpr (o < 2) cvt_prep_half(o);
            else math_op(J0{}, std::integral_constant<int, o - 2>{}, gl);
          }
        });
        // the residual's first row under the tile's last tap (two requests per gap of its last product)
        if constexpr (RES && POS == 2 && R == 2 && g >= 2 * PB) {
          constexpr int per = (8 * NT + PB - 1) / PB;
#pragma unroll
          for (int e = (g - 2 * PB) * per; e < (g - 2 * PB + 1) * per && e < 8 * NT; ++e) req_res(e);
        }
      });
    });
    XW_FENCE();
    // the patch buffers change roles: per-lane address toggles (once per stage, no select per use)
#pragma unroll
    for (int j = 0; j < 3; ++j) s_dst[j] ^= XTOG;
    xb_cur ^= XTOG;
    xb_nxt ^= XTOG;
    ex_dead ^= XTOG;
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using Yes = std::true_type;
  using No = std::false_type;

  // =========================================================================================================================
  // persistent loop (NS >= 3): the accumulators are written first by the tile's first product (C = 0) and read last by its epilogue
  // =========================================================================================================================
  for (int it = 0; it < n_my; ++it) {
#ifdef CSD_FF_TUNE
    if (it == 1) XW_WALL(28);
    if (it == 2) XW_WALL(29);
    if (it == 1 && a_dbg && tid == 0) a_dbg[blockIdx.x * 32 + 26] = clock64();
    if (it == 2 && a_dbg && tid == 0) a_dbg[blockIdx.x * 32 + 27] = clock64();
#endif
    unit(P0{}, Yes{}, 0);
    for (int s = 1; s + 2 < NS; ++s) unit(P0{}, No{}, s);
    gn = geom_of(tn);
    unit(P1{}, No{}, NS - 2);
    unit(P2{}, No{}, NS - 1);
    epilogue();
    tc = tn;
    gc = gn;
    tn = tile_at(it + 2);
  }
  XW_WALL(31);
}

template <int NT, bool NORM, bool RES>
static int launch_xk(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_xk_kernel<NT, NORM, RES>;
  CSD_SET_MAX_LDS_ONCE(kern);
  const int n_cu = device_cu_count8();               // persistent: one workgroup per CU, a multiple of the 8 XCDs
  const int grid = k.nblocks < n_cu ? (k.nblocks + 7) / 8 * 8 : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(XW_THREADS), XKCfg<NT>::LDS, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// 96-cout groups and the 64-cout groups of the nf = 128 nets; at least three 16-channel stages; the packed weights are the Winograd
// layout of conv_ff.hip's convxw_pack_kernel
bool convxk_supported(const ConvFFArgs& k, int nt) { return (nt == 3 || nt == 2) && k.nstage >= 3; }

int convxk_launch(const ConvFFArgs& k, int nt, hipStream_t s) {
  const bool norm = k.a.nscale != nullptr, res = k.a.res != nullptr;
  if (nt == 2) {
    if (norm) return res ? launch_xk<2, true, true>(k, s) : launch_xk<2, true, false>(k, s);
    return res ? launch_xk<2, false, true>(k, s) : launch_xk<2, false, false>(k, s);
  }
  if (norm) return res ? launch_xk<3, true, true>(k, s) : launch_xk<3, true, false>(k, s);
  return res ? launch_xk<3, false, true>(k, s) : launch_xk<3, false, false>(k, s);
}

}  // namespace csd
