// conv_fx.hip - the fp16f8 fused-prologue 3x3 convolution (conv_ff.hip's F8 form: reference models/layers.py:632-675,
// h = Conv(act(GroupNorm(x))) [+ Dense(temb)] / x + Conv(...)) with ONE workgroup per CU instead of two.
//
// Why (round 3 measurements, DESIGN.md / profiles/NOTEBOOK.md): conv_ff's two co-resident workgroups hardly overlap - one workgroup
// per CU alone reaches 85 % of the pair's throughput - and inside a lone workgroup a 16-channel K stage takes 8.9 k cycles for 3.6 k
// cycles of matrix work: nine barriers per stage (the weight ring holds one tap per group), the fp8 fragments read in the step that
// uses them (an exposed LDS round trip per tap pair), the weight DMA on one wave.  The matrix-instruction mix itself, with its LDS
// fragment traffic and no barriers, runs at 91-100 % of the matrix pipe (tools/probes/mfma_mix_probe.hip).  With the whole CU's resources
// for one workgroup (512 registers per lane, 160 KB LDS) the K loop can be that stream:
//   * the weights of a WHOLE stage (9 taps x 16 cin x 96 cout x (fp16 + two e4m3 planes) = 54 KB) are resident, double-buffered:
//     ONE barrier per stage instead of nine; all four waves issue the LDS-DMA of the next stage's weights (13-14 pieces each),
//     spread over the taps of the current stage;
//   * fragments are register-prefetched two taps (fp16) / one tap pair (fp8) ahead: no LDS wait in front of a matrix instruction;
//   * all four waves prefetch + convert the next stage's patch (5 slots each; wave 0 takes the four left-over pixels).
// Tile geometry, LDS patch layout, packed-weight layout, accumulator / epilogue / statistics code are conv_ff's (conv_ff.h).
#include "conv_ff.h"

namespace csd {

#define FX_THREADS 512

template <int NT>
struct FXCfg {
  static constexpr int KC = 16;
  static constexpr int TAPB = NT * 2 * 1024;                 // weight bytes per tap: NT cout tiles x (fp16 A fragments | e4m3 hi + lo blocks)
  static constexpr int STB = 9 * TAPB;                       // per stage
  static constexpr int PIECES = STB / 1024;                  // LDS-DMA instructions per stage
  static constexpr int PPW = (PIECES + 3) / 4;               // per producer wave
  static constexpr int NSLOT = 5;                            // full conversion slots per producer thread (256 x 5 = 1280 of the 1296); wave 0: + 1
  static constexpr size_t LDS = 2 * (size_t)FF_PATCH_BYTES + 2 * (size_t)STB + 4 * NT * 32 * 2 * sizeof(float) + NT * 32 * sizeof(float) + 32;
};

// two f32 -> one dword of two fp16 (round to nearest even: v_cvt_pk_f16_f32)
__device__ __forceinline__ int fx_pack_f16(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(int, __builtin_convertvector(f2{a, b}, h2));
}
// lo = v - (float)half: ONE v_fma_mix_f32 that reads the fp16 half of the packed dword directly (exact: fp32 fma)
template <bool HIGH>
__device__ __forceinline__ float fx_lo(int hp, float v) {
  float r;
  if constexpr (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}

template <int NT, bool NORM>
__global__ __launch_bounds__(FX_THREADS, 1) void conv_fx_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = FXCfg<NT>;
  constexpr int KC = C::KC, TAPB = C::TAPB, STB = C::STB, PIECES = C::PIECES, PPW = C::PPW, NSLOT = C::NSLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;                                         // 2 buffers
  char* const wbuf = smem + 2 * FF_PATCH_BYTES;                     // 2 stage buffers of weights in fragment order
  float* const red = reinterpret_cast<float*>(wbuf + 2 * STB);      // [4 waves][NT*32 couts][2]: statistics hand-over
  float* const btl = red + 4 * NT * 32 * 2;                         // [NT*32]: (bias + temb row) of the tile about to start
  char* const zero32 = reinterpret_cast<char*>(btl + NT * 32);      // 32 zero bytes: the x operand of the unpaired ninth tap's upper K half

  const float* const a_src0 = k.a.src0;
  const float* const a_src1 = k.a.src1;
  const float* const a_bias = k.a.bias;
  const float* const a_temb = k.a.temb;
  const float* const a_res = k.a.res;
  const float* const a_nscale = k.a.nscale;
  const float* const a_nshift = k.a.nshift;
  float* const a_out = k.a.out;
  double* const a_stats = k.a.stats;
  const int a_temb_stride = k.a.temb_stride, a_out_stride = k.a.out_stride, a_out_coff = k.a.out_coff;
  const float a_out_scale = k.a.out_scale;
  const int kH = k.H, kW = k.W, kC0 = k.C0, kC1 = k.C1, kCout = k.Cout, k_tiles_x = k.tiles_x, k_tpi = k.tpi,
            k_n_groups = k.n_groups, k_nblocks = k.nblocks, k_nstage = k.nstage;

  const int lane = threadIdx.x & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave8 >= 4;                  // waves 0-3: matrix waves (one per SIMD); waves 4-7: their SIMD partners, everything else
  const int wave = wave8 & 3;
  const int tid = threadIdx.x & 255;                 // thread index inside the role
  const int kh = lane >> 5, p32 = lane & 31;
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // MODE.FP16_OVFL = 1: fp8 (and fp16) conversions saturate instead of NaN / inf
#ifdef CSD_FF_TUNE
  long long* const a_dbg = k.a.dbg;
  int ts_n = 0;
  bool ts_on = false;
#define FX_TS() do { if (a_dbg && ts_on && (threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x < 4096 && ts_n < 14) a_dbg[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 16 + ts_n++] = clock64(); } while (0)
#define FX_WALL(i) do { if (a_dbg && (threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x < 4096) a_dbg[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define FX_TS() do { } while (0)
#define FX_WALL(i) do { } while (0)
#endif
  FX_WALL(14);

  // ---- persistent tile loop: workgroup p of gridDim.x (one per CU) runs the tiles j, j + P/8, ... of its XCD's contiguous share of
  // the launch (block p lands on XCD p % 8: the 32 workgroups of an XCD walk neighbouring tiles of the same samples) ----
  struct Tile { int w, ng, tile, b, ty0, tx0; size_t img0, pix; };
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, wstride = gridDim.x >> 3;
  const int xq = k_nblocks >> 3, xr = k_nblocks & 7;
  const int x_start = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, x_len = xq + (xcd < xr ? 1 : 0);
  auto tile_at = [&](int it) __attribute__((always_inline)) {
    Tile t;
    const int local = wj + it * wstride;
    t.w = local < x_len ? x_start + local : -1;
    const int w = t.w < 0 ? 0 : t.w;
    t.ng = w % k_n_groups;
    t.tile = w / k_n_groups;
    t.b = t.tile / k_tpi;
    const int tin = t.tile - t.b * k_tpi;
    t.ty0 = (tin / k_tiles_x) * FF_TILE;
    t.tx0 = (tin - (tin / k_tiles_x) * k_tiles_x) * FF_TILE;
    t.img0 = (size_t)t.b * kH * kW;
    t.pix = t.img0 + (size_t)t.ty0 * kW + t.tx0;
    return t;
  };
  const int Cin = kC0 + kC1;
  if (tile_at(0).w < 0) return;
  constexpr float FX_NLOG2E = -1.4426950408889634f;
  constexpr float acc_in = NORM ? C16_WSCALE * FX_NLOG2E : C16_WSCALE;         // = 2^8 / -ln2 (the exp2-domain SiLU of conv_ff.hip)
  constexpr float acc_out = NORM ? -0.6931471805599453f / C16_WSCALE : 1.0f / C16_WSCALE;

  // Hand-over protocol (every s_barrier is executed by all eight waves): A0 = stage 0 of the first tile is ready; E_g at the end of
  // every global stage g = the matrix waves are done with buffers g & 1 AND the producers have finished stage g + 1 in buffers
  // (g + 1) & 1 (its patch converted, its weights landed); after a tile's last stage one more barrier splits the statistics hand-over.
  if (producer) {
    // =====================================================================================================================
    // producer waves: patch prefetch + GroupNorm affine + SiLU + split (slot j of thread t = 4-channel group (t & 3) of patch
    // pixel j*64 + (t >> 2); producer wave 0's extra slot covers pixels 320..323), the weight DMA, the (bias + temb) row
    // =====================================================================================================================
    const int lg = tid & 3, lp0 = tid >> 2;
    // Two register sets (A: even global stages, B: odd): the loads of stage g + 2 are requested BEFORE stage g + 1 is converted, so an
    // HBM round trip lies under a whole conversion + hand-over instead of in front of it.
    struct Set {
      float4 pf[NSLOT + 1];
      float4 n_sc, n_sh;
      int doff[NSLOT + 1];                           // LDS record offset of each slot's pixel | bit 31: outside the image
      float4 brow;                                   // (bias + temb) row piece of the tile (stage 0, first NT*8 threads)
      int ng, b, stage;
    };
    Set A, B;
    int soff[NSLOT + 1], doff_cur[NSLOT + 1];        // slot geometry of the tile the fetch cursor is in
    auto slots_of = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j <= NSLOT; ++j) {
        const int pix = j < NSLOT ? j * 64 + lp0 : min(NSLOT * 64 + lp0, FF_NPATCH - 1);
        const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
        const int y = t.ty0 - 1 + pr, x = t.tx0 - 1 + pc;
        const bool in = y >= 0 && y < kH && x >= 0 && x < kW;        // zero padding outside THIS sample
        soff[j] = in ? y * kW + x : -1;
        doff_cur[j] = (pr * FF_RS + pc * FF_PSB) | (in ? 0 : (int)0x80000000);
      }
    };
    auto fetch = [&](Set& S, const Tile& t, int stage) __attribute__((always_inline)) {
      const int cb = stage * KC;
      const bool s1 = cb >= kC0;
      const float* src = (s1 ? a_src1 : a_src0) + t.img0 * (s1 ? kC1 : kC0) + (s1 ? cb - kC0 : cb) + lg * 4;
      const int Cs = s1 ? kC1 : kC0;
      // (wave 0's extras first: they are the oldest of the requests the hand-over's counted wait lets stay in flight)
      if (stage == 0 && tid < NT * 8) {
        const int c = t.ng * NT * 32 + tid * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_bias) v = gload4f(a_bias + c);
        if (a_temb) {
          const float4 tv = gload4f(a_temb + (size_t)t.b * a_temb_stride + c);
          v = make_float4(v.x + tv.x, v.y + tv.y, v.z + tv.z, v.w + tv.w);
        }
        S.brow = v;
      }
      if (wave == 0) S.pf[NSLOT] = gload4f(src + (size_t)(soff[NSLOT] >= 0 ? soff[NSLOT] : 0) * Cs);
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) S.pf[j] = gload4f(src + (size_t)(soff[j] >= 0 ? soff[j] : 0) * Cs);
      if constexpr (NORM) {
        S.n_sc = gload4f(a_nscale + (size_t)t.b * Cin + cb + lg * 4);
        S.n_sh = gload4f(a_nshift + (size_t)t.b * Cin + cb + lg * 4);
      }
#pragma unroll
      for (int j = 0; j <= NSLOT; ++j) S.doff[j] = doff_cur[j];
      S.ng = t.ng; S.b = t.b; S.stage = stage;
    };
    auto dma = [&](Set& S, int gbuf) __attribute__((always_inline)) {
      // the stage's weights: piece i (1 KiB, fragment order = linear) by producer wave i & 3
#pragma unroll
      for (int q = 0; q < PPW; ++q) {
        const int i = q * 4 + wave;
        if (i < PIECES)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(g_wpack + ((size_t)S.ng * (Cin / 16) + S.stage) * STB + i * 1024 + lane * 16),
              (__attribute__((address_space(3))) void*)(wbuf + gbuf * STB + i * 1024), 16, 0, 0);
      }
    };
    auto convert = [&](Set& S, int gbuf) __attribute__((always_inline)) {
      if (S.stage == 0 && tid < NT * 8) *reinterpret_cast<float4*>(btl + tid * 4) = S.brow;      // the tile's (bias + temb) row through LDS
      // convert + write the patch
      char* const buf = patch + gbuf * FF_PATCH_BYTES;
      float4 m_sc, m_sh;
      if constexpr (NORM) {
        m_sc = make_float4(S.n_sc.x * FX_NLOG2E, S.n_sc.y * FX_NLOG2E, S.n_sc.z * FX_NLOG2E, S.n_sc.w * FX_NLOG2E);
        m_sh = make_float4(S.n_sh.x * FX_NLOG2E, S.n_sh.y * FX_NLOG2E, S.n_sh.z * FX_NLOG2E, S.n_sh.w * FX_NLOG2E);
      }
      auto one = [&](int j) __attribute__((always_inline)) {
        const int dt = S.doff[j];
        const bool in = dt >= 0;
        float h[4] = {S.pf[j].x, S.pf[j].y, S.pf[j].z, S.pf[j].w};
        if constexpr (NORM) {
          h[0] = h[0] * m_sc.x + m_sh.x; h[1] = h[1] * m_sc.y + m_sh.y;
          h[2] = h[2] * m_sc.z + m_sh.z; h[3] = h[3] * m_sc.w + m_sh.w;
#pragma unroll
          for (int q = 0; q < 4; ++q) h[q] = h[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h[q]));
        }
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = in ? h[q] : 0.f;          // padding is applied to the ACTIVATED tensor: exactly 0
        char* const rec = buf + (dt & 0x7fffffff);
        const int hp0 = fx_pack_f16(v[0], v[1]), hp1 = fx_pack_f16(v[2], v[3]);
        *reinterpret_cast<int2*>(rec + lg * 8) = make_int2(hp0, hp1);
        const float l0 = fx_lo<false>(hp0, v[0]), l1 = fx_lo<true>(hp0, v[1]), l2 = fx_lo<false>(hp1, v[2]), l3 = fx_lo<true>(hp1, v[3]);
        short2v l8 = __builtin_bit_cast(short2v, __float_as_int(h[0]));
        l8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l8, l0, l1, 1.0f / 2048.0f, false);
        l8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l8, l2, l3, 1.0f / 2048.0f, true);
        int h8 = __float_as_int(h[1]);
        h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], h8, false);
        h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], h8, true);
        *reinterpret_cast<int*>(rec + 32 + lg * 4) = __builtin_bit_cast(int, l8);
        *reinterpret_cast<int*>(rec + 48 + lg * 4) = h8;
      };
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) one(j);
      if (wave == 0) one(NSLOT);
    };
    if (tid < 8) reinterpret_cast<int*>(zero32)[tid] = 0;
    // fetch cursor: the (tile, stage) whose loads are requested next; it runs two global stages ahead of the matrix waves
    int f_it = 0, f_s = 0;
    Tile f_t = tile_at(0);
    bool f_ok = true;
    auto f_advance = [&]() __attribute__((always_inline)) {
      if (++f_s == k_nstage) {
        f_s = 0;
        f_t = tile_at(++f_it);
        f_ok = f_t.w >= 0;
        if (f_ok) slots_of(f_t);
      }
    };
    slots_of(f_t);
    fetch(A, f_t, 0);                                // global stage 0
    f_advance();
    if (f_ok) fetch(B, f_t, f_s);                    // global stage 1
    const bool b_ok0 = f_ok;
    if (f_ok) f_advance();
    dma(A, 0);
    convert(A, 0);
    ff_wait_vm<0>();
    ff_barrier();                                    // A0
    // producers' view of the matrix waves' progress: global stage g, stage sc of its tile
    bool nxt_ok = b_ok0;                             // "global stage g + 1 exists"
    for (int g = 0, sc = 0;; ++g) {
      // the set of stage g + 2 is the one stage g was converted from: even g -> A
      auto step = [&](Set& Sf, Set& Sn) __attribute__((always_inline)) {
        const bool f2 = nxt_ok && f_ok;              // stage g + 2 exists
        if (nxt_ok) dma(Sn, (g + 1) & 1);            // stage g + 1's weights first: the counted wait below lets only YOUNGER requests fly
        if (f2) { fetch(Sf, f_t, f_s); f_advance(); }
        if (nxt_ok) convert(Sn, (g + 1) & 1);
        // the DMA has landed; stage g + 2's loads (every wave issues at least NSLOT (+ 2) of them after its DMA pieces) stay in flight
        if (f2) ff_wait_vm<NSLOT + (NORM ? 2 : 0)>(); else ff_wait_vm<0>();
        nxt_ok = f2;
      };
      if (g & 1) step(B, A); else step(A, B);
      ff_barrier();                                  // E_g
      if (++sc == k_nstage) {
        sc = 0;
        if (a_stats) ff_barrier();                   // (the matrix waves' statistics hand-over)
        if (!nxt_ok) break;                          // (no stage g + 1: that was the last tile)
      }
    }
    return;
  }

  // =======================================================================================================================
  // matrix waves: wave w owns pixel rows 4w..4w+3 of the 16 x 16 tile as two 4 x 8 M tiles and ALL NT cout tiles; LDS fragment
  // reads (register-prefetched two taps ahead; the fp8 fragments one tap ahead), matrix instructions, the residual (requested in
  // pieces during the last stage), the epilogue.  Lane = pixel p32 of each M tile; its 16 results of cout tile nt are couts
  // nt*32 + 8q + 4kh + i (q = r >> 2, i = r & 3).
  // =======================================================================================================================
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  int base[2], base8[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    base[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + kh * 16;
    base8[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + 32;
  }
  floatx16 acc[2][NT];
  Tile cur = tile_at(0);
  int g = 0;
  ff_barrier();                                      // A0
  for (int it = 0;; ++it) {
#ifdef CSD_FF_TUNE
    ts_on = it == 1;                                 // (stamps: the second tile = steady state)
#endif
    FX_TS();
    const Tile nxt = tile_at(it + 1);
    const bool has_next = nxt.w >= 0;
    // the PIXELS are the MFMA's M operand (conv_ff.hip): a lane holds ONE cout (nt*32 + p32) of 16 pixels of each M tile - register r =
    // pixel row 4 wave + r / 4, column 8 mt + 4 kh + r % 4 - so residual loads and output stores cover whole 128-byte lines and the
    // GroupNorm partials are in-lane sums
    const int c_lane = cur.ng * NT * 32 + p32;
    // ---- accumulators start at (bias + temb) * acc_in; the residual joins them during the first stages ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float bq = btl[nt * 32 + p32];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bq * acc_in;
    }
    const __amdgpu_buffer_rsrc_t res_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_res ? a_res + cur.pix * kCout : a_out), 0, OOB, RSRC_FLAGS);

    float rr[2][16];                                 // residual chunks in flight (even / odd chunk index)
    const unsigned res_voff = (unsigned)(4 * kh * kCout + c_lane) * 4u;
    auto upix = [&](int mt, int r) __attribute__((always_inline)) { return (4 * wave + (r >> 2)) * kW + 8 * mt + (r & 3); };      // uniform
    auto res_load = [&](auto c_tag) __attribute__((always_inline)) {
      constexpr int CI = decltype(c_tag)::value, mt = CI / NT, nt = CI % NT;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rr[CI & 1][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, res_voff, (unsigned)(upix(mt, r) * kCout + nt * 32) * 4u, 0));
    };
    if (a_res != nullptr) res_load(std::integral_constant<int, 0>{});
    // One stage (9 taps x 16 channels).  RC >= 0: the residual's (M tile, cout tile) chunk RC (16 dwords per lane) is requested at the
    // top of the stage and added after its last tap - an HBM round trip under a whole stage of matrix work.  RC is a compile-time
    // constant (the first 2 NT stages of a tile are unrolled): a run-time accumulator selection would put branches into the K loop.
    auto stage = [&](auto rc_tag) __attribute__((always_inline)) {
      constexpr int RC = decltype(rc_tag)::value;
      const char* const pb = patch + (g & 1) * FF_PATCH_BYTES;
      const char* const wb = wbuf + (g & 1) * STB;
      half8 wa[2][NT], xb[2][2];
      int8v wa8[NT], xb8[2];
      auto ld16 = [&](int buf, int tap) __attribute__((always_inline)) {
        const int r = tap / 3, sx = tap - r * 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wa[buf][nt] = *reinterpret_cast<const half8*>(wb + tap * TAPB + nt * 2048 + lane * 16);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xb[buf][mt] = *reinterpret_cast<const half8*>(pb + base[mt] + r * FF_RS + sx * FF_PSB);
      };
      // operands of the correction MFMA of the tap pair (tap, tap + 1): lanes 0-31 carry tap, lanes 32-63 tap + 1; the unpaired
      // ninth tap multiplies zeros in the upper K half
      auto ld8 = [&](int tap) __attribute__((always_inline)) {
        const bool pair = tap + 1 < 9;
        const int t1 = pair ? tap + 1 : tap;
        const int o0 = (tap / 3) * FF_RS + (tap % 3) * FF_PSB, o1 = (t1 / 3) * FF_RS + (t1 % 3) * FF_PSB;
        const char* const wp = wb + (kh ? t1 : tap) * TAPB + p32 * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const uint4f a = *reinterpret_cast<const uint4f*>(wp + nt * 2048 + 1024);
          const uint4f c = *reinterpret_cast<const uint4f*>(wp + nt * 2048 + 1024 + 512);
          wa8[nt] = int8v{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)c.x, (int)c.y, (int)c.z, (int)c.w};
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const char* const xp = (!pair && kh) ? zero32 : pb + base8[mt] + (kh ? o1 : o0);      // (address select: no data select)
          const uint4f a = *reinterpret_cast<const uint4f*>(xp);
          const uint4f c = *reinterpret_cast<const uint4f*>(xp + 16);
          xb8[mt] = int8v{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)c.x, (int)c.y, (int)c.z, (int)c.w};
        }
      };
      const bool do_res = RC >= 0 && a_res != nullptr;
      if constexpr (RC >= 0 && RC + 1 < 2 * NT) {
        if (do_res) res_load(std::integral_constant<int, RC + 1>{});      // the NEXT stage's chunk: two stages of cover for the round trip
      }
      ld16(0, 0);
      ld8(0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cb3 = tap & 1;
        if (tap + 1 < 9) ld16(cb3 ^ 1, tap + 1);
        const bool corr = (tap & 1) == 0;            // taps 0, 2, 4, 6 bring their right neighbour, tap 8 goes alone
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i)
          acc[i / NT][i % NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xb[cb3][i / NT], wa[cb3][i % NT], acc[i / NT][i % NT], 0, 0, 0);
        if (corr) {
#pragma unroll
          for (int i = 0; i < 2 * NT; ++i)
            acc[i / NT][i % NT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xb8[i / NT], wa8[i % NT], acc[i / NT][i % NT], 0, 0, 0,
                                                                                    127, 0, 116);      // block scale 2^-11 on B (the weights)
        } else if (tap + 1 < 9) {
          ld8(tap + 1);                              // the next pair's fp8 fragments, one tap ahead (single register set)
        }
      }
      if constexpr (RC >= 0) {
        if (do_res) {
          constexpr int mt = RC / NT, nt = RC % NT;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] += rr[RC & 1][r] * acc_in;
        }
      }
      FX_TS();
      ff_barrier();                                  // E_g
      ++g;
    };
    // the first 2 NT stages carry one residual chunk each (the host sends layers with fewer stages to conv_ff)
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    if constexpr (NT == 3) {
      stage(std::integral_constant<int, 4>{});
      stage(std::integral_constant<int, 5>{});
    }
    for (int s = 2 * NT; s < k_nstage; ++s) stage(std::integral_constant<int, -1>{});

    // ---- epilogue ----
    const __amdgpu_buffer_rsrc_t out_r =
        __builtin_amdgcn_make_buffer_rsrc(a_out + cur.pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = acc[mt][nt][r] * acc_out * a_out_scale;
    const unsigned out_voff = (unsigned)(4 * kh * a_out_stride + c_lane) * 4u;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r]), out_r, out_voff, (unsigned)(upix(mt, r) * a_out_stride + nt * 32) * 4u, 0);
    FX_TS();

    // ---- GroupNorm partials of the written tile: (sum, sum of squares) per cout over its 256 pixels (conv_ff's order: in-lane over the
    // lane's 32 pixels, one exchange between the K halves, the four matrix waves through LDS) ----
    if (a_stats) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float vs = 0.f, vq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a0 = acc[0][nt][r], a1 = acc[1][nt][r];
          vs += a0 + a1;
          vq += a0 * a0 + a1 * a1;
        }
        vs += __shfl_xor(vs, 32);
        vq += __shfl_xor(vq, 32);
        if (kh == 0) {
          red[(wave * NT * 32 + nt * 32 + p32) * 2 + 0] = vs;
          red[(wave * NT * 32 + nt * 32 + p32) * 2 + 1] = vq;
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
        double* dst = a_stats + ((size_t)cur.tile * kCout + cur.ng * NT * 32 + tid) * 2;
        dst[0] = sm;
        dst[1] = sq;
      }
    }
    FX_TS();
    if (!has_next) break;
    cur = nxt;
  }
  FX_WALL(15);
}

template <int NT, bool NORM>
static int launch_fx(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_fx_kernel<NT, NORM>;
  CSD_SET_MAX_LDS_ONCE(kern);
  const int n_cu = device_cu_count8();               // persistent: one workgroup per CU, a multiple of the 8 XCDs
  const int grid = k.nblocks < n_cu ? (k.nblocks + 7) / 8 * 8 : n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FX_THREADS), FXCfg<NT>::LDS, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// the fp16f8 (ns = 3) layers conv_ff covers; same packed weights, same arguments
int convfx_launch(const ConvFFArgs& k, int nt, hipStream_t s) {
  const bool norm = k.a.nscale != nullptr;
  if (nt == 2) return norm ? launch_fx<2, true>(k, s) : launch_fx<2, false>(k, s);
  return norm ? launch_fx<3, true>(k, s) : launch_fx<3, false>(k, s);
}

}  // namespace csd
