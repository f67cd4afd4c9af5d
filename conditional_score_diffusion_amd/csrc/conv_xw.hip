// conv_xw.hip - conv_xp's operator (the fp16x3 fused-prologue 3x3 convolution, reference models/layers.py:119-132,632-675) with
// ONE THIRD FEWER matrix instructions: 1-D Winograd F(2,3) along the image row.
//
// Why (profiles/NOTEBOOK.md, round 4 / 5): conv_xp holds the package at its power limit with three fp16 MFMAs per algorithmic
// product - the only lever left is fewer MFMAs per product.  For an output pair (y0, y1) of a row, inputs d0..d3 and a filter row
// (g0, g1, g2):
//     D0 = d0 - d2   D1 = d1 + d2   D2 = d2 - d1   D3 = d1 - d3                       (input transform, fp32, BEFORE the hi | lo split)
//     G0 = g0        G1 = (g0 + g1 + g2) / 2       G2 = (g0 - g1 + g2) / 2   G3 = g2  (weights, transformed at pack time)
//     Mk = sum over input channels and the 3 filter ROWS of Dk * Gk                   (4 contractions instead of 6 per output pair)
//     y0 = M0 + M1 + M2      y1 = M1 - M2 - M3                                        (output transform in the epilogue)
// A 16-channel stage of a 16 x 16-pixel x 32 NT-cout tile is 3 row taps x 12 NT MFMAs (conv_xp: 9 taps x 6 NT).
//
// Structure = conv_xp's: one persistent 4-wave workgroup per CU (512 registers per lane), every wave ONE instruction stream
//   MFMA | fillers | MFMA | ...  with the fillers pinned in program order by scheduling fences.  What differs:
//   * a wave owns 32 pixel PAIRS (4 rows x 8 pairs) x all four components x NT cout tiles = 4 NT accumulators (192 registers at NT = 3);
//     per row tap the products hi*lo, hi*hi, lo*hi, each round robin over the accumulators (consecutive MFMAs independent);
//   * LDS (160 KB) does not hold two whole stages of transformed weights (2 x 72 KB) next to the transformed patch double buffer
//     (2 x 36 KB): the weights live in a ring of THREE row-tap slots (slot = filter row) and the workgroup meets at ONE barrier per
//     row tap.  Fragment registers are single-buffered and re-loaded in rotation: this tap's wh / xl under its first product, the
//     next tap's wl under the second, the next tap's xh under the third;
//   * the conversion works on UNITS of two adjacent patch pixels x 4 channels: GroupNorm affine + exp2-domain SiLU on the unit's 8
//     values, the right-hand neighbour unit's values by ds_bpermute (lane + 4), the four components, hi | lo split, eight
//     ds_write_b64.  A wave's 16 units per slot overlap the next slot's by one unit (the neighbour provider), 3 slots per stage;
//     the patch of stage X = s + 1 may only be written while taps 0 and 1 of stage s run (tap 2 already reads it), so slot 0 is
//     computed one tap early (tap 2 of the stage before, results held in 16 registers) and stored first thing in tap 0;
//   * the residual is requested in four row chunks (the first under the tile's last tap, the others at the top of the epilogue).
// Packed weights: [cout group][cin / 16][filter row][component][cout tile][hi | lo][64 lanes x 8 halves], x 2^8 like conv_ff's.
#include "conv_ff.h"

// Product library: every layer this kernel covered runs on conv_xk.hip (same operator, same or - for the 64-cout groups - the Winograd
// arithmetic); the kernel is compiled in the TUNING build only, where it is conv_xk's A/B partner (CSD_XK=0).
#ifndef CSD_TUNE
namespace csd {
bool convxw_supported(const ConvFFArgs&, int) { return false; }
int convxw_launch(const ConvFFArgs&, int, hipStream_t) {
  set_error("conv_xw: a tuning-build kernel (superseded by conv_xk.hip in the product library)");
  return CSD_ERR_INVALID;
}
}  // namespace csd
#else

#include <utility>

namespace csd {

#define XW_THREADS 256
#define XW_RS (32 * 64 + 16)                         // LDS pitch of a transformed patch row: (component, pair) records of 64 B (16 ch hi | 16 ch lo)
#define XW_PATCH_BYTES (FF_PW * XW_RS)               // 37152
#define XW_NU 162                                    // conversion units per stage: 18 patch rows x 9 column pairs

template <int NT_>
struct XWCfg {
  static constexpr int NT = NT_;
  static constexpr int NA = 4 * NT;                          // accumulators per wave = MFMAs per product
  static constexpr int GP = 3 * NA;                          // MFMAs (= filler gaps) per row tap
  static constexpr int TAPB = 4 * NT * 2 * 1024;             // weight bytes per row tap: 4 components x NT cout tiles x (hi | lo)
  static constexpr int STB = 3 * TAPB;                       // per stage
  static constexpr int WPT = 2 * NT;                         // 1 KiB pieces per wave and row tap
  // LDS map: the two patch buffers sit 64 KiB apart, so that changing buffers is ONE xor with a constant for every per-lane address
  // (a select or a per-lane toggle per use costs registers and vector instructions in a stream that has neither to spare):
  //   [0, 37152) patch 0 | ring slot 0 | dummy | [65536, 102688) patch 1 | ring slot 1 | the dummy's partner | [131072, ..) ring slot 2 | red
  static constexpr int XTOG = 65536;
  static constexpr int SLOT0 = XW_PATCH_BYTES, OFF_DUMMY = SLOT0 + TAPB, SLOT1 = XTOG + XW_PATCH_BYTES, SLOT2 = 2 * XTOG;
  static constexpr int OFF_RED = SLOT2 + TAPB;
  static constexpr size_t LDS = (size_t)OFF_RED + 4 * NT * 32 * 2 * sizeof(float);
  static constexpr int slot_off(int r) { return r == 0 ? SLOT0 : r == 1 ? SLOT1 : SLOT2; }
  static_assert(OFF_DUMMY + 3072 <= XTOG && SLOT1 + TAPB + 3072 <= SLOT2 + 0 * TAPB + 4096 && LDS <= 160 * 1024, "conv_xw: LDS map");
};

template <class F, int... I>
__device__ __forceinline__ void xw_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void xw_static_for(F&& f) {
  xw_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

typedef unsigned int xw_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xw_pack_f16(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(int, __builtin_convertvector(f2{a, b}, h2));
}
template <bool HIGH>
__device__ __forceinline__ float xw_lo(int hp, float v) {      // v - (float)half: one v_fma_mix_f32 (exact)
  float r;
  if constexpr (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}

#define XW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define XW_SADD(x, y) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(y) : "scc")
#define XW_PIN(a) asm volatile("" : "+v"(a))
// tuning aids (never in the product library): XW_ABL bits remove parts of the stream at compile time (results are then garbage):
// 1 conversion, 2 weight staging, 4 fragment reads, 8 epilogue stores, 16 patch requests, 32 residual requests, 64 barriers,
// 16384 row-tap barriers without the LDS wait, 32768 the weight ring's LDS stores (the requests stay), 4096 patch requests confined to the first 256 pixels of the sample (cache hits), 256 patch stores, 512 neighbour exchange (own value instead), 1024 transcendentals (plain multiplies instead), 2048 hi | lo split
#ifndef XW_ABL
#define XW_ABL 0
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
// fixed fillers of gap g (PB = 4 NT gaps per product): the fragment reads and the weight ring traffic
template <int NT>
struct XWSched {
  static constexpr int PB = 4 * NT, GP = 3 * PB, WPT = 2 * NT;
  static constexpr bool rd_wh(int g) { return g < PB; }                                              // wh[g] of THIS tap
  static constexpr bool rd_xl(int g) { return g < PB && g % NT == (NT > 1 ? 1 : 0); }                // xl[g / NT] of this tap
  static constexpr int RW = 4;                                                                       // the wl ring (registers)
  static constexpr bool rd_wlB(int g) { return g < PB - RW; }                                        // wl[RW + g] of THIS tap (into the register MFMA g just read)
  static constexpr bool rd_wlA(int g) { return g >= GP - RW; }                                       // wl[g - (GP - RW)] of the NEXT tap
  static constexpr bool rd_xh(int g) { return g >= 2 * PB && (g - 2 * PB) % 2 == 0 && (g - 2 * PB) / 2 < 4; }      // xh[(g - 2 PB) / 2] of the next tap
  // Vector memory: requests retire IN ORDER (one counter for loads and stores), so a wait for an L2-hit weight piece also waits for
  // every OLDER request - and the patch requests go to HBM.  Per row tap therefore, in this order: gap 0 the tap's two patch
  // requests (slot = tap index: its registers were consumed under the tap before), gaps 1 .. WPT the ring stores of the WPT pieces
  // requested under the tap before (they are OLDER than this tap's patch requests, and the patch requests of the tap before are a
  // whole tap old), gaps WPT + 1 .. 2 WPT the requests of the next pieces into the same registers.
  static constexpr bool slot_req(int g) { return g == 0; }
  static constexpr bool put(int g) { return g >= 1 && g <= WPT; }                                    // ring store of piece g - 1
  static constexpr bool req(int g) { return g > WPT && g <= 2 * WPT; }                               // request of piece g - WPT - 1
  static constexpr int wq(int g) { return put(g) ? g - 1 : g - WPT - 1; }
  static constexpr int fixed(int g) { return (rd_wh(g) ? 1 : 0) + (rd_xl(g) ? 1 : 0) + (rd_wlA(g) ? 1 : 0) + (rd_wlB(g) ? 1 : 0) + (rd_xh(g) ? 1 : 0) + (put(g) ? 1 : 0) + (req(g) ? 1 : 0) + (slot_req(g) ? 3 : 0); }
  // the conversion's micro-operations go where the fixed fillers leave room: weight of a gap = 14 - 3 fixed (thirds of an issue slot)
  static constexpr int wgt(int g) { return 14 - 3 * fixed(g) > 2 ? 14 - 3 * fixed(g) : 2; }
  static constexpr int cum(int g) { int s = 0; for (int h = 0; h < g; ++h) s += wgt(h); return s; }
  static constexpr int first_op(int n, int g) { return (int)(((long long)n * cum(g)) / cum(GP)); }    // ops [first_op(n, g), first_op(n, g + 1)) in gap g
};

template <int NT_, bool NORM, bool RES>
__global__ __launch_bounds__(XW_THREADS, 1) void conv_xw_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = XWCfg<NT_>;
  using S = XWSched<NT_>;
  constexpr int NT = C::NT, NA = C::NA, GP = C::GP, TAPB = C::TAPB, STB = C::STB, WPT = C::WPT, PB = S::PB;
  constexpr int OFF_RED = C::OFF_RED, OFF_DUMMY = C::OFF_DUMMY, XTOG = C::XTOG;
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

  // ---- this workgroup's tiles (conv_xp's walk): workgroup p (one per CU, on XCD p % 8) takes tiles wj, wj + P/8, ... of its XCD's share ----
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

  // fragment bases: pixel pairs (M operand): rows 4 wave + (p32 >> 3), pair p32 & 7, K half kh; weights: lane * 16
  const int xbase = (4 * wave + (p32 >> 3)) * XW_RS + (p32 & 7) * 64 + kh * 16;
  int xb_cur = xbase, xb_nxt = xbase + XTOG;         // ... in the buffer of the stage being multiplied | of the next stage
  const int wbase = lane * 16;

  // ---- requests ----
  xw_u4 pfa[3], pfb[3];                              // the raw float4 of each slot's two pixels
  xw_u4 scn, shn;                                    // GroupNorm scale / shift of the lane's 4 channels
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
    if (!(XW_ABL & 16)) pfa[j] = __builtin_amdgcn_raw_buffer_load_b128(srcL, (XW_ABL & 8192) ? (unsigned)(tid * 16 + j * 8192) : __umul24((XW_ABL & 4096) ? (g.p[j] & 255) : g.p[j], strideL) + lg * 16, soffL, 0);
  };
  auto req_b = [&](int j, const Geom& g) __attribute__((always_inline)) {
    if (!(XW_ABL & 16)) pfb[j] = __builtin_amdgcn_raw_buffer_load_b128(srcL, (XW_ABL & 8192) ? (unsigned)(tid * 16 + j * 8192 + 4096) : __umul24(((XW_ABL & 4096) ? (g.p[j] & 255) : g.p[j]) + ((g.p[j] >> 29) & 1), strideL) + lg * 16, soffL, 0);
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
  // weights: the packed tensor is one linear stream per (cout group, tile): stage after stage, row tap after row tap, 1 KiB pieces;
  // piece i = q * 4 + wave of a row tap.  The scalar offset runs (asm add) and is re-based when the stream moves to the next tile.
  const __amdgpu_buffer_rsrc_t w_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g_wpack), 0, OOB, RSRC_FLAGS);
  const int wvoff = lane * 16 + wave * 1024;
  xw_u4 wr[WPT];
  int w_run = 0;
  const int c4096 = 4096;
  auto w_begin = [&](int wso) __attribute__((always_inline)) { w_run = wso; };
  auto req_w = [&](int q) __attribute__((always_inline)) {      // the next piece of the stream into register q
    if (XW_ABL & 2) return;
    wr[q] = __builtin_amdgcn_raw_buffer_load_b128(w_r, (unsigned)wvoff, w_run, 0);
    XW_SADD(w_run, c4096);
  };
  auto put_w = [&](int q, int slot) __attribute__((always_inline)) {
    if (XW_ABL & (2 | 32768)) return;
    *reinterpret_cast<xw_u4*>(smem + C::slot_off(slot) + q * 4096 + wvoff) = wr[q];
  };

  // ---- conversion of a slot, as a sequence of single operations (placed one by one between the MFMAs) ----
  // "pre": per value (8 = pixels a, b x 4 channels) the chain affine > exp2 > 1 + > rcp > u * > zero padding > neighbour value; emitted
  // along the diagonals of the (value, phase) table, so that a dependent operation sits 7 operations behind its producer and the
  // transcendentals (12.8 cycles of issue each, two per MFMA gap ride free - tools/filler_cost_probe.hip) are spread out.
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
    if (XW_ABL & 1) return;
    if constexpr (ol < NPRE) {
      constexpr int i = PRE.val[ol], c = i & 3, o = PRE.ph[ol] + (NORM ? 0 : 5);      // value i (a: i < 4), channel c, phase o
      if constexpr (o == 0) {
        const float x = __uint_as_float(i < 4 ? pfa[j][c] : pfb[j][c]);
        cv[i] = fmaf(x, msc[c], msh[c]);
        XW_PIN(cv[i]);
      } else if constexpr (o == 1) {
        ce[i] = (XW_ABL & 1024) ? cv[i] * 1.5f : __builtin_amdgcn_exp2f(cv[i]);
        XW_PIN(ce[i]);
      } else if constexpr (o == 2) {
        ce[i] = 1.0f + ce[i];
        XW_PIN(ce[i]);
      } else if constexpr (o == 3) {
        ce[i] = (XW_ABL & 1024) ? ce[i] * 0.7f : __builtin_amdgcn_rcpf(ce[i]);
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
        if (XW_ABL & 512) { cn[i] = cv[i] * 0.5f; XW_PIN(cn[i]); }
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
        if (XW_ABL & 2048) { chp[kc][hh] = __float_as_int(cd[kc][0]); clp[kc][hh] = __float_as_int(cd[kc][1]); return; }
        chp[kc][hh] = xw_pack_f16(cd[kc][0], cd[kc][1]);
        XW_PIN(chp[kc][hh]);
      } else if constexpr (o < 20) {
        if (XW_ABL & 2048) return;
        constexpr int kc = (o - 12) >> 1, c = (o - 12) & 1;
        if constexpr (c == 0) cl[kc][0] = xw_lo<false>(chp[kc][hh], cd[kc][0]);
        else cl[kc][1] = xw_lo<true>(chp[kc][hh], cd[kc][1]);
        XW_PIN(cl[kc][c]);
      } else {
        if (XW_ABL & 2048) return;
        constexpr int kc = o - 20;
        clp[kc][hh] = xw_pack_f16(cl[kc][0], cl[kc][1]);
        XW_PIN(clp[kc][hh]);
      }
    }
  };
  // the eight stores of a slot: component kc hi (o = kc), lo (o = 4 + kc), into the patch buffer `par1` (0 | 1)
  auto write_op = [&](auto j_tag, auto o_tag) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, o = decltype(o_tag)::value;
    if (XW_ABL & (1 | 256)) return;
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

  // ---- fragments (single-buffered, re-loaded in rotation) ----
  // pixel pairs of component k; weights of (component, cout tile) i = k * NT + nt.  A weight lo fragment feeds exactly one MFMA: they
  // stream through a ring of NA / 2 registers (first half read under the tap before, second half behind the MFMAs that free the ring)
  constexpr int RW = S::RW;
  half8 xh[4], xl[4], wh[NA], wl[RW];
  auto rd_x = [&](half8& dst, int xb, int r, int kc, int pl) __attribute__((always_inline)) {
    if (XW_ABL & 4) return;
    dst = *reinterpret_cast<const half8*>(smem + xb + r * XW_RS + kc * 512 + pl * 32);
  };
  auto rd_w = [&](half8& dst, int r, int i, int pl) __attribute__((always_inline)) {
    if (XW_ABL & 4) return;
    dst = *reinterpret_cast<const half8*>(smem + wbase + C::slot_off(r) + i * 2048 + pl * 1024);
  };

  // The accumulators live in the accumulator half of the register file for the whole kernel; the matrix instructions are asm
  // statements (conv_xp.hip explains why); a tile's first product writes them with C = 0, so nothing is live across the tile loop's
  // back edge, and the epilogue's reads sit behind tie_acc_done().  tools/check_xp_isa.py checks the generated code.
  floatx16 acc[4][NT];
  // product p of a row tap: 0 hi * lo, 1 hi * hi, 2 lo * hi (the pixel-side hi fragments are free for the next tap after product 1)
  auto mm = [&](int p, int i) __attribute__((always_inline)) {
    const int kc = i / NT, nt = i - kc * NT;
    // (wh and xl - 64 registers - live in the accumulator half of the register file: the LDS reads land there directly, and the
    // vector half, 256 registers, keeps the conversion, the requests and the other fragments without spilling)
    if (p == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[kc][nt]) : "v"(xh[kc]), "v"(wl[i % RW]));
    else if (p == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[kc][nt]) : "v"(xh[kc]), "a"(wh[i]));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[kc][nt]) : "a"(xl[kc]), "a"(wh[i]));
  };
  auto mm_first = [&](int i) __attribute__((always_inline)) {      // a tile's first product: C = 0
    const int kc = i / NT, nt = i - kc * NT;
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc[kc][nt]) : "v"(xh[kc]), "v"(wl[i % RW]));
  };
  auto tie_acc_done = [&]() __attribute__((always_inline)) {
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
    out_voff = (unsigned)((4 * wave * kW + 8 * (lf >> 5)) * a_out_stride + c_lane) * 4u;
    if constexpr (RES) res_voff = (unsigned)((4 * wave * kW + 8 * (lf >> 5)) * kCout + c_lane) * 4u;
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
      if (XW_ABL & 32) return;
      const int idx = e / NT, nt = e - idx * NT;
      rs[idx >> 3][idx & 7][nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, res_voff + nt * 128, res_run, 0));
      if (nt == NT - 1 && idx < 31) {
        if ((idx & 7) != 7) XW_SADD(res_run, res_col);
        else XW_SADD(res_run, res_row);
      }
    }
  };
  float* const red = reinterpret_cast<float*>(smem + OFF_RED);      // [4 waves][NT*32 couts][2]: statistics hand-over
  auto epilogue = [&]() __attribute__((always_inline)) {
    if constexpr (RES) {                             // rows 1 .. 3 of the residual (row 0 came in under the last tap)
#pragma unroll
      for (int e = 8 * NT; e < 32 * NT; ++e) req_res(e);
    }
    tie_acc_done();
    float vs[NT], vq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) vs[nt] = vq[nt] = 0.f;
    int out_run = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int r = row * 4 + jj;
        float y1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bs = (bv[nt] + tv[nt]) * a_out_scale;
          const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
          const float y0 = (m0 + m1) + m2;
          y1[nt] = (m1 - m2) - m3;
          float v;
          if constexpr (RES) v = fmaf(y0, ka, fmaf(rs[row][2 * jj][nt], a_out_scale, bs));
          else v = fmaf(y0, ka, bs);
          if (!(XW_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), out_r, out_voff + nt * 128, out_run, 0);
          vs[nt] += v;
          vq[nt] = fmaf(v, v, vq[nt]);
        }
        XW_SADD(out_run, out_col);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bs = (bv[nt] + tv[nt]) * a_out_scale;
          float v;
          if constexpr (RES) v = fmaf(y1[nt], ka, fmaf(rs[row][2 * jj + 1][nt], a_out_scale, bs));
          else v = fmaf(y1[nt], ka, bs);
          if (!(XW_ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), out_r, out_voff + nt * 128, out_run, 0);
          vs[nt] += v;
          vq[nt] = fmaf(v, v, vq[nt]);
        }
        if (jj != 3) XW_SADD(out_run, out_col);
        else if (row != 3) XW_SADD(out_run, out_row);
      }
    }
    if (a_stats) {
      const int lf = fresh_lane(), tidf = wave * 64 + lf;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
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
  // the stream's own rules, the weights of row taps 0 and 1 in the ring and those of row tap 2 requested, the first fragments read
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
    xw_static_for<NFULL>([&](auto o) __attribute__((always_inline)) { full_op(J0{}, o, gc, gc); });
    req_a(0, gc); req_b(0, gc);
    xw_static_for<NFULL>([&](auto o) __attribute__((always_inline)) { full_op(J1{}, o, gc, gc); });
    req_a(1, gc); req_b(1, gc);
    xw_static_for<NFULL>([&](auto o) __attribute__((always_inline)) { full_op(J2{}, o, gc, gc); });
    req_a(2, gc); req_b(2, gc);
#pragma unroll
    for (int j = 0; j < 3; ++j) s_dst[j] ^= XTOG;          // from here on the stream writes buffer 1 (stage 1)
    // stage 1, slot 0: computed and held (stored by the first row tap)
    cvt_prep_half(0); cvt_prep_half(1);
    xw_static_for<NMATH>([&](auto o) __attribute__((always_inline)) { math_op(J0{}, o, gc); });
#pragma unroll
    for (int r = 0; r < 2; ++r) {                    // row taps 0 and 1
#pragma unroll
      for (int q = 0; q < WPT; ++q) req_w(q);
#pragma unroll
      for (int q = 0; q < WPT; ++q) put_w(q, r);
    }
#pragma unroll
    for (int q = 0; q < WPT; ++q) req_w(q);          // row tap 2 (stored under row tap 0)
  }
  ff_barrier();
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) rd_x(xh[kc], xb_cur, 0, kc, 0);
#pragma unroll
  for (int i = 0; i < RW; ++i) rd_w(wl[i], 0, i, 1);

  // =========================================================================================================================
  // One unit = the three row taps of stage s of tile tc (patch buffer `par`); meanwhile stage X = s + 1 is converted into the other
  // buffer, stage L = s + 2 requested, and the weight ring turns: row tap r stores the pieces of the tap two ahead (requested under
  // the tap before) into slot (r + 2) % 3 and requests the tap three ahead.
  //   tap 0: [store slot 0 of X (held)] [request slot 0 of L] [slot 1 of X: math, stores, request for L]   + the stage's scalar set-up
  //   tap 1: [slot 2 of X: math, stores, request for L]
  //   tap 2: [the affine of L] [slot 0 of L: math, held]                 (tap 2 already reads X's patch: nothing may be stored)
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
    xw_static_for<3>([&](auto r_tag) __attribute__((always_inline)) {
      constexpr int R = decltype(r_tag)::value;
      XW_FENCE();
      if (!(XW_ABL & 64)) ff_barrier();
      XW_FENCE();
      if constexpr (R == 0) src_of(tl, sl);          // (scalar: the descriptor of stage L's requests, used from this tap's gap 0 on)
      constexpr int NOPS = R == 0 ? 8 + NFULL : R == 1 ? NFULL : 2 + NMATH;
      xw_static_for<GP>([&](auto g_tag) __attribute__((always_inline)) {
        constexpr int g = decltype(g_tag)::value;
        XW_FENCE();
        if constexpr (FIRST && R == 0 && g < NA) mm_first(g);
        else mm(g / NA, g % NA);
        XW_FENCE();
        // ---- fixed fillers of gap (R, g) ----
        if constexpr (S::rd_wh(g)) rd_w(wh[g], R, g, 0);
        if constexpr (S::rd_xl(g)) rd_x(xl[g / NT], xb_cur, R, g / NT, 1);
        if constexpr (S::rd_wlB(g)) rd_w(wl[g % RW], R, RW + g, 1);
        if constexpr (S::rd_wlA(g)) rd_w(wl[g - (GP - RW)], (R + 1) % 3, g - (GP - RW), 1);
        if constexpr (S::rd_xh(g)) rd_x(xh[(g - 2 * PB) / 2], R == 2 ? xb_nxt : xb_cur, (R + 1) % 3, (g - 2 * PB) / 2, 0);
        if constexpr (S::slot_req(g)) { req_a(R, gl); req_b(R, gl); }      // slot R for stage L (consumed two taps from now)
        if constexpr (S::put(g)) put_w(S::wq(g), (R + 2) % 3);
        if constexpr (S::req(g)) req_w(S::wq(g));
        // ---- scalar set-up of the stage, in the first gaps of tap 0 ----
        if constexpr (R == 0 && g == 2 && POS == 1) epi_setup(tc);
        if constexpr (R == 1 && g == PB + PB / 2) req_norm(tl, sl);                     // (used by tap 2's first operations)
        if constexpr (R == 2 && POS == 2 && g == PB) epi_loads(tc);
        if constexpr (R == 0 && POS == 2 && g == 1) w_begin(tn.ng * NSTB);           // (this unit's requests are the next tile's stage 0)
        // ---- the conversion's operations of this gap ----
        constexpr int o0 = S::first_op(NOPS, g), o1 = S::first_op(NOPS, g + 1);
        xw_static_for<o1 - o0>([&](auto d_tag) __attribute__((always_inline)) {
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
        // the residual's first row under the tile's last tap (two requests per gap of its last product, behind the weight requests)
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
static int launch_xw(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_xw_kernel<NT, NORM, RES>;
  CSD_SET_MAX_LDS_ONCE(kern);
  const int n_cu = device_cu_count8();               // persistent: one workgroup per CU, a multiple of the 8 XCDs
  const int grid = k.nblocks < n_cu ? (k.nblocks + 7) / 8 * 8 : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(XW_THREADS), XWCfg<NT>::LDS, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// the fp16x3 layers of conv_xp with 96-cout groups and at least three 16-channel stages
bool convxw_supported(const ConvFFArgs& k, int nt) { return nt == 3 && k.nstage >= 3; }

int convxw_launch(const ConvFFArgs& k, int nt, hipStream_t s) {
  const bool norm = k.a.nscale != nullptr, res = k.a.res != nullptr;
  if (norm) return res ? launch_xw<3, true, true>(k, s) : launch_xw<3, true, false>(k, s);
  return res ? launch_xw<3, false, true>(k, s) : launch_xw<3, false, false>(k, s);
}

}  // namespace csd
#endif  // CSD_TUNE
