// conv_fx.hip - the fp16f8 fused-prologue 3x3 convolution (conv_ff.hip's F8 form: reference models/layers.py:632-675,
// h = Conv(act(GroupNorm(x))) [+ Dense(temb)] / x + Conv(...)) with ONE workgroup per CU instead of two.
//
// Why (round 3 measurements, DESIGN.md / profiles/NOTEBOOK.md): conv_ff's two co-resident workgroups hardly overlap - one workgroup
// per CU alone reaches 85 % of the pair's throughput - and inside a lone workgroup a 16-channel K stage takes 8.9 k cycles for 3.6 k
// cycles of matrix work: nine barriers per stage (the weight ring holds one tap per group), the fp8 fragments read in the step that
// uses them (an exposed LDS round trip per tap pair), the weight DMA on one wave.  The matrix-instruction mix itself, with its LDS
// fragment traffic and no barriers, runs at 91-100 % of the matrix pipe (tools/mfma_mix_probe.hip).  With the whole CU's resources
// for one workgroup (512 registers per lane, 160 KB LDS) the K loop can be that stream:
//   * the weights of a WHOLE stage (9 taps x 16 cin x 96 cout x (fp16 + two e4m3 planes) = 54 KB) are resident, double-buffered:
//     ONE barrier per stage instead of nine; all four waves issue the LDS-DMA of the next stage's weights (13-14 pieces each),
//     spread over the taps of the current stage;
//   * fragments are register-prefetched two taps (fp16) / one tap pair (fp8) ahead: no LDS wait in front of a matrix instruction;
//   * all four waves prefetch + convert the next stage's patch (5 slots each; wave 0 takes the four left-over pixels).
// Tile geometry, LDS patch layout, packed-weight layout, accumulator / epilogue / statistics code are conv_ff's (conv_ff.h).
#include "conv_ff.h"

namespace csd {

template <int NT>
struct FXCfg {
  static constexpr int KC = 16;
  static constexpr int TAPB = NT * 2 * 1024;                 // weight bytes per tap: NT cout tiles x (fp16 A fragments | e4m3 hi + lo blocks)
  static constexpr int STB = 9 * TAPB;                       // per stage
  static constexpr int PIECES = STB / 1024;                  // LDS-DMA instructions per stage
  static constexpr int PPW = (PIECES + 3) / 4;               // per wave
  static constexpr int NSLOT = 5;                            // full conversion slots per thread (256 threads x 5 = 1280 of the 1296); wave 0: + 1
  static constexpr size_t LDS = 2 * (size_t)FF_PATCH_BYTES + 2 * (size_t)STB + 2 * FF_NPATCH * sizeof(int);
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
__global__ __launch_bounds__(FF_THREADS, 1) void conv_fx_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = FXCfg<NT>;
  constexpr int KC = C::KC, TAPB = C::TAPB, STB = C::STB, PIECES = C::PIECES, PPW = C::PPW, NSLOT = C::NSLOT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;                                         // 2 buffers
  char* const wbuf = smem + 2 * FF_PATCH_BYTES;                     // 2 stage buffers of weights in fragment order
  int* const stab = reinterpret_cast<int*>(wbuf + 2 * STB);         // [324] source pixel index inside the sample, or -1
  int* const dtab = stab + FF_NPATCH;                               // [324] LDS byte offset of the patch pixel | bit 31: outside the image

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

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, p32 = lane & 31;
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // MODE.FP16_OVFL = 1: fp8 (and fp16) conversions saturate instead of NaN / inf
#ifdef CSD_FF_TUNE
  long long* const a_dbg = k.a.dbg;
  int ts_n = 0;
#define FX_TS() do { if (a_dbg && (tid == 0 || tid == 64) && blockIdx.x < 4096 && ts_n < 14) a_dbg[(blockIdx.x * 2 + (tid >> 6)) * 16 + ts_n++] = clock64(); } while (0)
#define FX_WALL(i) do { if (a_dbg && (tid == 0 || tid == 64) && blockIdx.x < 4096) a_dbg[(blockIdx.x * 2 + (tid >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define FX_TS() do { } while (0)
#define FX_WALL(i) do { } while (0)
#endif
  FX_TS();
  FX_WALL(14);

  int w;
  {
    const int bid = blockIdx.x, nb = k_nblocks;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ng = w % k_n_groups;
  const int tile = w / k_n_groups;
  const int b = tile / k_tpi;
  const int tin = tile - b * k_tpi;
  const int ty0 = (tin / k_tiles_x) * FF_TILE, tx0 = (tin - (tin / k_tiles_x) * k_tiles_x) * FF_TILE;
  const int Cin = kC0 + kC1;

  // ---- patch prefetch + conversion: slot j of thread t = 4-channel group (t & 3) of patch pixel j*64 + (t >> 2); wave 0's extra
  // slot covers pixels 320..323 (its lanes past 15 repeat pixel 323: same data to the same address) ----
  const int lg = tid & 3, lp0 = tid >> 2;
  float4 pf[NSLOT + 1];
  float4 n_sc = make_float4(1.f, 1.f, 1.f, 1.f), n_sh = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t img0 = (size_t)b * kH * kW;
  auto issue_patch = [&](int stage, auto from_table) __attribute__((always_inline)) {
    const int cb = stage * KC;
    const bool s1 = cb >= kC0;
    const float* src = (s1 ? a_src1 : a_src0) + img0 * (s1 ? kC1 : kC0) + (s1 ? cb - kC0 : cb) + lg * 4;
    const int Cs = s1 ? kC1 : kC0;
    auto one = [&](int j) __attribute__((always_inline)) {
      const int pix = j < NSLOT ? j * 64 + lp0 : min(NSLOT * 64 + lp0, FF_NPATCH - 1);
      int sp;
      if constexpr (decltype(from_table)::value) {
        sp = stab[pix];
      } else {                                       // (the first stage is requested before the tables exist)
        const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
        const int y = ty0 - 1 + pr, x = tx0 - 1 + pc;
        sp = (y >= 0 && y < kH && x >= 0 && x < kW) ? y * kW + x : -1;
      }
      pf[j] = gload4f(src + (size_t)(sp >= 0 ? sp : 0) * Cs);      // (out-of-image pixels read pixel 0: replaced by zeros when stored)
    };
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) one(j);
    if (wave == 0) one(NSLOT);
    if constexpr (NORM) {
      n_sc = gload4f(a_nscale + (size_t)b * Cin + cb + lg * 4);
      n_sh = gload4f(a_nshift + (size_t)b * Cin + cb + lg * 4);
    }
  };
  // SiLU in the exp2 domain (conv_ff.hip): the staged operand is a = u / (1 + 2^u), u = -log2(e) (x s + t); the factor -ln2 rides on
  // the accumulators (acc_in / acc_out)
  constexpr float FX_NLOG2E = -1.4426950408889634f;
  constexpr float acc_in = NORM ? C16_WSCALE * FX_NLOG2E : C16_WSCALE;
  constexpr float acc_out = NORM ? -0.6931471805599453f / C16_WSCALE : 1.0f / C16_WSCALE;
  auto store_patch = [&](char* buf) __attribute__((always_inline)) {
    float4 m_sc, m_sh;
    if constexpr (NORM) {
      m_sc = make_float4(n_sc.x * FX_NLOG2E, n_sc.y * FX_NLOG2E, n_sc.z * FX_NLOG2E, n_sc.w * FX_NLOG2E);
      m_sh = make_float4(n_sh.x * FX_NLOG2E, n_sh.y * FX_NLOG2E, n_sh.z * FX_NLOG2E, n_sh.w * FX_NLOG2E);
    }
    auto one = [&](int j) __attribute__((always_inline)) {
      const int pix = j < NSLOT ? j * 64 + lp0 : min(NSLOT * 64 + lp0, FF_NPATCH - 1);
      const int dt = dtab[pix];
      const bool in = dt >= 0;
      float h[4] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w};
      if constexpr (NORM) {
        h[0] = h[0] * m_sc.x + m_sh.x; h[1] = h[1] * m_sc.y + m_sh.y;
        h[2] = h[2] * m_sc.z + m_sh.z; h[3] = h[3] * m_sc.w + m_sh.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = h[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h[q]));
      }
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = in ? h[q] : 0.f;            // padding is applied to the ACTIVATED tensor: exactly 0
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

  // ---- weight stream: piece i of a stage (1 KiB, fragment order = linear) is issued by wave i & 3 ----
  const char* const wsrc = g_wpack + (size_t)ng * ((size_t)(Cin / 16) * STB) + lane * 16;
  auto issue_w = [&](int stage, int q) __attribute__((always_inline)) {       // q: compile-time after unrolling
    const int i = q * 4 + wave;
    if (i < PIECES)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)stage * STB + i * 1024),
                                       (__attribute__((address_space(3))) void*)(wbuf + (stage & 1) * STB + i * 1024), 16, 0, 0);
  };

  // ---- prologue ----
#pragma unroll
  for (int q = 0; q < PPW; ++q) issue_w(0, q);
  issue_patch(0, std::false_type{});
  for (int pix = tid; pix < FF_NPATCH; pix += FF_THREADS) {
    const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
    const int y = ty0 - 1 + pr, x = tx0 - 1 + pc;
    const bool in = y >= 0 && y < kH && x >= 0 && x < kW;          // zero padding outside THIS sample
    stab[pix] = in ? y * kW + x : -1;
    dtab[pix] = (pr * FF_RS + pc * FF_PSB) | (in ? 0 : (int)0x80000000);
  }

  // ---- accumulators start at (bias + temb) * acc_in; lane = pixel p32 of each M tile, couts nt*32 + 8q + 4kh + i ----
  const int c_lane = ng * NT * 32 + kh * 4;
  floatx16 acc[2][NT];
  {
    float4 bv[NT * 4];
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a_bias) {
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) bv[i] = gload4f(a_bias + c_lane + (i >> 2) * 32 + (i & 3) * 8);
    }
    if (a_temb) {
      float4 tv[NT * 4];
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) tv[i] = gload4f(a_temb + (size_t)b * a_temb_stride + c_lane + (i >> 2) * 32 + (i & 3) * 8);
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) bv[i] = make_float4(bv[i].x + tv[i].x, bv[i].y + tv[i].y, bv[i].z + tv[i].z, bv[i].w + tv[i].w);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[mt][nt][q * 4 + 0] = bv[nt * 4 + q].x * acc_in; acc[mt][nt][q * 4 + 1] = bv[nt * 4 + q].y * acc_in;
          acc[mt][nt][q * 4 + 2] = bv[nt * 4 + q].z * acc_in; acc[mt][nt][q * 4 + 3] = bv[nt * 4 + q].w * acc_in;
        }
  }
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  const size_t tile_pix = img0 + (size_t)ty0 * kW + tx0;
  int opix[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) opix[mt] = (4 * wave + (p32 >> 3)) * kW + 8 * mt + (p32 & 7);
  if (a_res != nullptr) {
    const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_res + tile_pix * kCout), 0, OOB, RSRC_FLAGS);
    uint4f rv[2][NT][4];                             // (512 registers per lane: the whole residual tile in flight at once)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rv[mt][nt][q] = __builtin_amdgcn_raw_buffer_load_b128(res_r, (unsigned)(opix[mt] * kCout + c_lane + nt * 32 + q * 8) * 4u, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[mt][nt][q * 4 + 0] += __uint_as_float(rv[mt][nt][q].x) * acc_in;
          acc[mt][nt][q * 4 + 1] += __uint_as_float(rv[mt][nt][q].y) * acc_in;
          acc[mt][nt][q * 4 + 2] += __uint_as_float(rv[mt][nt][q].z) * acc_in;
          acc[mt][nt][q * 4 + 3] += __uint_as_float(rv[mt][nt][q].w) * acc_in;
        }
  }

  int base[2], base8[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    base[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + kh * 16;
    base8[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + 32;
  }

  ff_barrier();                                      // tables visible
  FX_TS();
  store_patch(patch);
  ff_wait_vm<0>();                                   // this wave's share of stage 0's weights has landed
  ff_barrier();
  FX_TS();

  for (int s = 0; s < k_nstage; ++s) {
    const bool more = s + 1 < k_nstage;
    if (more) issue_patch(s + 1, std::true_type{});
    const char* const pb = patch + (s & 1) * FF_PATCH_BYTES;
    const char* const wb = wbuf + (s & 1) * STB;
    half8 wa[3][NT], xb[3][2];
    int8v wa8[2][NT], xb8[2][2];
    auto ld16 = [&](int buf, int tap) __attribute__((always_inline)) {
      const int r = tap / 3, sx = tap - r * 3;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wa[buf][nt] = *reinterpret_cast<const half8*>(wb + tap * TAPB + nt * 2048 + lane * 16);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xb[buf][mt] = *reinterpret_cast<const half8*>(pb + base[mt] + r * FF_RS + sx * FF_PSB);
    };
    // operands of the correction MFMA of the tap pair (tap, tap + 1): lanes 0-31 carry tap, lanes 32-63 tap + 1; the unpaired ninth
    // tap multiplies zeros in the upper K half
    auto ld8 = [&](int buf, int tap) __attribute__((always_inline)) {
      const bool pair = tap + 1 < 9;
      const int t1 = pair ? tap + 1 : tap;
      const int o0 = (tap / 3) * FF_RS + (tap % 3) * FF_PSB, o1 = (t1 / 3) * FF_RS + (t1 % 3) * FF_PSB;
      const char* const wp = wb + (kh ? t1 : tap) * TAPB + p32 * 16;
      const bool z = !pair && kh;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4f a = *reinterpret_cast<const uint4f*>(wp + nt * 2048 + 1024);
        const uint4f c = *reinterpret_cast<const uint4f*>(wp + nt * 2048 + 1024 + 512);
        wa8[buf][nt] = int8v{z ? 0 : (int)a.x, z ? 0 : (int)a.y, z ? 0 : (int)a.z, z ? 0 : (int)a.w,
                             z ? 0 : (int)c.x, z ? 0 : (int)c.y, z ? 0 : (int)c.z, z ? 0 : (int)c.w};
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const char* const xp = pb + base8[mt] + (kh ? o1 : o0);
        const uint4f a = *reinterpret_cast<const uint4f*>(xp);
        const uint4f c = *reinterpret_cast<const uint4f*>(xp + 16);
        xb8[buf][mt] = int8v{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)c.x, (int)c.y, (int)c.z, (int)c.w};
      }
    };
    ld16(0, 0);
    ld16(1, 1);
    ld8(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap % 3;
      if (tap + 2 < 9) ld16((tap + 2) % 3, tap + 2);
      const bool corr = (tap & 1) == 0;              // taps 0, 2, 4, 6 bring their right neighbour, tap 8 goes alone
      if (corr && tap + 2 < 9) ld8(((tap >> 1) + 1) & 1, tap + 2);
#pragma unroll
      for (int i = 0; i < 2 * NT; ++i)
        acc[i / NT][i % NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][i % NT], xb[cur][i / NT], acc[i / NT][i % NT], 0, 0, 0);
      // the next stage's weights: two DMA pieces per tap and wave (taps 0-6), under the matrix instructions above
      if (more) {
        if (2 * tap < PPW) issue_w(s + 1, 2 * tap);
        if (2 * tap + 1 < PPW) issue_w(s + 1, 2 * tap + 1);
      }
      if (corr) {
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i)
          acc[i / NT][i % NT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa8[(tap >> 1) & 1][i % NT], xb8[(tap >> 1) & 1][i / NT],
                                                                                  acc[i / NT][i % NT], 0, 0, 0, 116, 0, 127);      // block scale 2^-11 on A
      }
    }
    FX_TS();
    if (more) store_patch(patch + ((s + 1) & 1) * FF_PATCH_BYTES);
    ff_wait_vm<0>();                                 // this wave's DMA pieces of stage s + 1 have landed
    ff_barrier();
    FX_TS();
  }

  // ---- epilogue (conv_ff.hip) ----
  const __amdgpu_buffer_rsrc_t out_r =
      __builtin_amdgcn_make_buffer_rsrc(a_out + tile_pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = acc[mt][nt][r] * acc_out * a_out_scale;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint4f ov;
        ov.x = __float_as_uint(acc[mt][nt][q * 4 + 0]); ov.y = __float_as_uint(acc[mt][nt][q * 4 + 1]);
        ov.z = __float_as_uint(acc[mt][nt][q * 4 + 2]); ov.w = __float_as_uint(acc[mt][nt][q * 4 + 3]);
        const unsigned off = (unsigned)(opix[mt] * a_out_stride + c_lane + nt * 32 + q * 8) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(ov, out_r, off, 0, 0);
      }

  // ---- GroupNorm partials of the written tile: (sum, sum of squares) per cout over its 256 pixels (the reduction tree of conv_ff:
  // same order, same bits) ----
  if (a_stats) {
    constexpr int NV = NT * 16;
    float vs[NV], vq[NV];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a0 = acc[0][nt][r], a1 = acc[1][nt][r];
        vs[nt * 16 + r] = a0 + a1;
        vq[nt * 16 + r] = a0 * a0 + a1 * a1;
      }
#define FX_HALVE(XCHG, BIT, H)                                                                   \
    {                                                                                            \
      const bool up = (lane >> BIT) & 1;                                                         \
      _Pragma("unroll") for (int i = 0; i < H; ++i) {                                            \
        const float ss = up ? vs[i] : vs[i + H], ks = up ? vs[i + H] : vs[i];                    \
        const float sq = up ? vq[i] : vq[i + H], kq = up ? vq[i + H] : vq[i];                    \
        vs[i] = ks + XCHG(ss, BIT);                                                              \
        vq[i] = kq + XCHG(sq, BIT);                                                              \
      }                                                                                          \
    }
#define FX_X_DPP(v, BIT) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (BIT) == 0 ? 0xB1 : 0x4E, 0xF, 0xF, true))
#define FX_X_SHFL(v, BIT) __shfl_xor(v, 1 << (BIT))
    FX_HALVE(FX_X_DPP, 0, NV / 2)
    FX_HALVE(FX_X_DPP, 1, NV / 4)
    FX_HALVE(FX_X_SHFL, 2, NV / 8)
    FX_HALVE(FX_X_SHFL, 3, NV / 16)
#undef FX_HALVE
#undef FX_X_DPP
#undef FX_X_SHFL
    constexpr int NF = NV / 16;
    float* const red = reinterpret_cast<float*>(smem);       // [4 waves][NT*32 couts][2] (the patch buffers are dead)
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      vs[i] += __shfl_xor(vs[i], 16);
      vq[i] += __shfl_xor(vq[i], 16);
    }
    __syncthreads();
    if ((lane & 16) == 0) {
      const int sel = (lane & 1) * (NV / 2) + ((lane >> 1) & 1) * (NV / 4) + ((lane >> 2) & 1) * (NV / 8) +
                      ((lane >> 3) & 1) * (NV / 16);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int idx = sel + i;                       // = nt*16 + r
        const int nt = idx >> 4, r = idx & 15;
        const int cl = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        red[(wave * NT * 32 + cl) * 2 + 0] = vs[i];
        red[(wave * NT * 32 + cl) * 2 + 1] = vq[i];
      }
    }
    __syncthreads();
    if (tid < NT * 32) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        s += (double)red[(wv * NT * 32 + tid) * 2 + 0];
        q += (double)red[(wv * NT * 32 + tid) * 2 + 1];
      }
      double* dst = a_stats + ((size_t)tile * kCout + ng * NT * 32 + tid) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
  FX_TS();
  FX_WALL(15);
}

template <int NT, bool NORM>
static int launch_fx(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_fx_kernel<NT, NORM>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(k.nblocks), dim3(FF_THREADS), FXCfg<NT>::LDS, s, reinterpret_cast<const char*>(k.a.wpack), k);
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
