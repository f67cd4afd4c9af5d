// conv_ff.hip - "fused-prologue" 3x3 stride-1 convolution of the ResnetBlock convs (reference models/layers.py:632-675:
// h = Conv(act(GroupNorm(x))) [+ Dense(temb)] / x + Conv(...); models/layerspp.py:212-274) at the resolutions where a
// 16 x 16 pixel tile lies inside one sample (H % 16 == 0 and W % 16 == 0: the 160^2 / 80^2 levels of SR3-160, every level
// >= 16^2 of the 128^2 / 64^2 / 256^2 nets).
//
// What it fuses: the GroupNorm affine + activation + fp16 (hi|lo) split of the conv OPERAND.  The other fp16 schedules
// read fp16 planes that gn_apply16_kernel wrote (fp32 in, 2*NS bytes out per element, then 2*NS bytes back in) - a pure
// HBM pass that cost 17-22 % of a PC step.  Here the fp32 residual stream (two-source virtual concat allowed) is the
// kernel's input: every element is converted ONCE per workgroup while it is staged, by three of the four waves.
//
// Schedule (one workgroup = 4 waves = 256 output pixels (16 x 16) x NT*32 couts, 2 workgroups per CU):
//   * MFMA: v_mfma_f32_32x32x16_f16 with A = pixels, B = weights: a lane's 16 results are ONE cout of 16 pixels, so a residual load /
//     output store of a wave is two whole 128-byte lines per instruction and the GroupNorm partials are in-lane sums (the
//     weights-as-M order gave each lane 4 x 4 consecutive couts of one pixel - 16-byte stores whose 32-byte pieces reach L2 as four
//     partial-line writes: 3.3 TB/s against 5.4 TB/s in tools/probes/store_probe.hip, and -6..8 % on the 160^2 layers).  Wave w owns pixel
//     rows 4w..4w+3 as two 4 x 8 M tiles (that lane->pixel map + a 1168-byte LDS row pitch makes every ds_read_b128 of a tap
//     conflict-free) and ALL NT cout tiles: 2*NT accumulators, 2 + NT fragment reads per K step.
//   * weights: through LDS, shared by the four waves (each wave pulling its own fragments from L2 saturates the
//     64 B/clk L1 path - the binding resource of the quad / loader-consumer schedules).  Wave 0 streams them with
//     global_load_lds_dwordx4 (no registers, fragment order = linear) into a ring of R groups, a few hundred cycles
//     ahead; it issues no other VMEM load inside the loop, so its in-order vmcnt queue holds L2-latency DMAs only.
//   * activations: waves 1-3 fetch the NEXT stage's patch (18 x 18 pixels x KC channels, fp32) into registers at the
//     top of a stage, convert + write it into the other LDS patch buffer at the end of the stage: a full stage
//     (>= 1.4 us of MFMAs) of latency cover, and their vmcnt queue holds nothing but that burst.
//   * one LDS-only barrier per ring group (18 MFMAs per wave between barriers in both arithmetic modes).
//   NS = 1 (fp16): KC = 32 channels per stage (2 K steps per tap), ring group = 3 steps.
//   NS = 2 (fp16x3: hi|lo operands, 3 MFMAs per product): KC = 16, ring group = 1 step.
// Epilogue: acc starts at (bias + temb) * 2^8, residual added, out_scale, full-line dword stores, and the GroupNorm partials
// (sum, sum of squares per (tile, cout)) of the written tensor for the NEXT GroupNorm - fp32 sums over the tile's 256
// pixels in a fixed order, folded in fp64 by gn_finalize_tiles_kernel.
#include "conv_ff.h"

namespace csd {

// F8 (NS = 2 only): "fp16 + fp8 corrections" - the operands are still split x = hi + lo, hi*hi runs on v_mfma_f32_32x32x16_f16, but
// the two correction products hi_w*lo_x + lo_w*hi_x (2^-11 of the result: ~4 significant bits are enough) run K-concatenated
// on the fp8 matrix cores: ONE v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 operands, block scale 2^-11 undoing the scaling
// of the lo parts) per TWO taps x 16 channels, its K = 64 being [hi_w | lo_w] . [lo_x | hi_x] of tap t (lanes 0-31) and of
// tap t+1 (lanes 32-63).  MFMA cycles per 16-channel stage: 54 x 32 + 30 x 64 = 3648 against 162 x 32 = 5184; measured
// network error 2e-5 norm-wise / 8e-5 element-wise (oracle/fp8_correction_study.py) against 1.2e-6 / 4.5e-6 of the full split.
// LDS formats: pixel record [hi fp16 x16 | lo*2^11 e4m3 x16 | hi e4m3 x16] (64 B, as NS = 2); weight step [cout tile][plane]
// with plane 0 = fp16 A fragments, plane 1 = [hi e4m3 | lo*2^11 e4m3][cout row][16 channels] (16 B per row and half).
// two f32 -> one dword of two fp16 (round to nearest even: v_cvt_pk_f16_f32)
__device__ __forceinline__ int ff_pack_f16(float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(int, __builtin_convertvector(f2{a, b}, h2));
}
// lo = v - (float)half: ONE v_fma_mix_f32 that reads the fp16 half of the packed dword directly (exact: fp32 fma)
template <bool HIGH>
__device__ __forceinline__ float ff_lo(int hp, float v) {
  float r;
  if constexpr (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(v));
  return r;
}

// NORM: the operand is act(GroupNorm(x)) (per-(sample, channel) scale / shift + SiLU); false: the raw tensor
template <int NS, int NT, bool F8, bool NORM>
__global__ __launch_bounds__(FF_THREADS, 2) void conv_ff_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  static_assert(!F8 || NS == 2, "the fp8-correction form shares the two-plane layouts");
  using C = FFCfg<NS, NT>;
  constexpr int KC = C::KC, STEPS = C::STEPS, TG = C::TG, GPS = C::GPS, SB = C::SB, GB = C::GB, GL = C::GL, R = C::R;
  constexpr int G4 = C::G4, NSLOT = C::NSLOT, PPJ = C::PPJ, NDMA = C::NDMA, GLW = C::GLW;
  static_assert(GL % NDMA == 0, "weight DMA split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;                                         // 2 buffers
  char* const ring = smem + 2 * FF_PATCH_BYTES;                     // R groups of weights in fragment order
  int* const stab = reinterpret_cast<int*>(ring + R * GB);          // [324] source pixel index inside the sample, or -1
  int* const dtab = stab + FF_NPATCH;                               // [324] LDS byte offset of the patch pixel

  // (local copies: a lambda that captures the by-value kernel argument struct by reference forces it into scratch)
  const float* const a_src0 = k.a.src0;
  const float* const a_src1 = k.a.src1;
  const float* const a_bias = k.a.bias;
  const float* const a_temb = k.a.temb;
  const float* const a_res = k.a.res;
  const float* const a_nscale = k.a.nscale;
  const float* const a_nshift = k.a.nshift;
  float* const a_out = k.a.out;
  double* const a_stats = k.a.stats;
  long long* const a_dbg = k.a.dbg;
  const int a_temb_stride = k.a.temb_stride, a_out_stride = k.a.out_stride, a_out_coff = k.a.out_coff, a_act = k.a.act;
  const float a_out_scale = k.a.out_scale;
  const int kH = k.H, kW = k.W, kC0 = k.C0, kC1 = k.C1, kCout = k.Cout, k_tiles_x = k.tiles_x, k_tpi = k.tpi,
            k_n_groups = k.n_groups, k_nblocks = k.nblocks, k_nstage = k.nstage, k_abl = k.abl;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, p32 = lane & 31;
  if constexpr (F8) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);      // MODE.FP16_OVFL = 1: fp8 (and fp16) conversions saturate instead of NaN / inf
#ifdef CSD_FF_TUNE
#ifdef CSD_FF_CONST_ABL
#define FF_ABL(bit) ((CSD_FF_CONST_ABL) & (bit))      // compile-time ablation: the tested code is really gone
#else
#define FF_ABL(bit) (k_abl & (bit))
#endif
  int ts_n = 0;
#define FF_TS() do { if (a_dbg && (tid == 0 || tid == 64) && blockIdx.x < 4096 && ts_n < 16) a_dbg[(blockIdx.x * 2 + (tid >> 6)) * 16 + ts_n++] = clock64(); } while (0)
#define FF_FS(i) do { if (a_dbg && (k_abl & 256) && s == 2 && (tid == 0 || tid == 64) && blockIdx.x < 1024) a_dbg[4096 * 2 * 16 + (blockIdx.x * 2 + (tid >> 6)) * 64 + (i)] = clock64(); } while (0)
#define FF_WALL(i) do { if (a_dbg && (tid == 0 || tid == 64) && blockIdx.x < 4096) a_dbg[(blockIdx.x * 2 + (tid >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define FF_WALL(i) do { } while (0)
#define FF_FS(i) do { } while (0)
#define FF_ABL(bit) false
#define FF_TS() do { } while (0)
#endif
  FF_TS();
  FF_WALL(14);

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

  // ---- loader state (waves 1-3): slot j of thread t = 4-channel group (t % G4) of patch pixel j*PPJ + t/G4 ----
  const int lt = tid - 64 * NDMA;
  const int lg = (lt >= 0 ? lt : 0) % G4, lp0 = (lt >= 0 ? lt : 0) / G4;
  float4 pf[NSLOT];
  float4 n_sc = make_float4(1.f, 1.f, 1.f, 1.f), n_sh = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t img0 = (size_t)b * kH * kW;
  auto issue_patch = [&](int stage, auto from_table) __attribute__((always_inline)) {      // (waves 1-3 only)
    const int cb = stage * KC;
    const bool s1 = cb >= kC0;
    const float* src = (s1 ? a_src1 : a_src0) + img0 * (s1 ? kC1 : kC0) + (s1 ? cb - kC0 : cb) + lg * 4;
    const int Cs = s1 ? kC1 : kC0;
    // straight-line: slots past the patch and out-of-image pixels read pixel 0 of the sample (their values are replaced
    // by zeros / dumped when stored) - a per-slot branch would put every load into its own basic block behind a vmcnt(0)
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int pix = min(j * PPJ + lp0, FF_NPATCH - 1);
      int sp;
      if constexpr (decltype(from_table)::value) {
        sp = stab[pix];
      } else {                                       // (the first stage is requested before the tables exist)
        const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
        const int y = ty0 - 1 + pr, x = tx0 - 1 + pc;
        sp = (y >= 0 && y < kH && x >= 0 && x < kW) ? y * kW + x : -1;
      }
      pf[j] = gload4f(src + (size_t)(sp >= 0 ? sp : 0) * Cs);
    }
    if constexpr (NORM) {
      n_sc = gload4f(a_nscale + (size_t)b * Cin + cb + lg * 4);
      n_sh = gload4f(a_nshift + (size_t)b * Cin + cb + lg * 4);
    }
  };
  // SiLU in the exp2 domain: with u = -log2(e) (x s + t) the activated value is act = -ln2 * u / (1 + 2^u).  The staged operand is
  // a = u / (1 + 2^u) (one v_exp_f32 + one v_rcp_f32, no argument scaling); the factor -ln2 is linear and rides on the accumulators:
  // they start at (bias + temb + residual) * 2^8 / -ln2 and the epilogue's one multiply restores it (acc_in / acc_out).
  constexpr float FF_NLOG2E = -1.4426950408889634f;
  constexpr float acc_in = NORM ? C16_WSCALE * FF_NLOG2E : C16_WSCALE;         // = 2^8 / -ln2
  constexpr float acc_out = NORM ? -0.6931471805599453f / C16_WSCALE : 1.0f / C16_WSCALE;
  // convert + write the prefetched stage: GroupNorm affine + SiLU + split, ~40 vector instructions per 4 channels (every one of
  // them is matrix-pipe time on this chip: tools/probes/mfma_valu_overlap.hip, tools/probes/mfma_valu_prio.hip - a partner wave's VALU stream
  // overlaps a dense MFMA stream by 15-20 % whatever the age / s_setprio of the two waves).  Per element: fma, v_exp, add, v_rcp,
  // mul (SiLU), select (zero padding of the ACTIVATED tensor), half a packed f32 -> f16 conversion, ONE mixed-precision fma for
  // lo = v - hi (reads the fp16 half directly), and in the F8 form half an e4m3 conversion each for lo * 2^11 and for v.
  // (the host routes activations other than SiLU - none of the reference's configs uses one - to the older schedules)
  auto store_patch = [&](char* buf) __attribute__((always_inline)) {
    float4 m_sc, m_sh;
    if constexpr (NORM) {
      m_sc = make_float4(n_sc.x * FF_NLOG2E, n_sc.y * FF_NLOG2E, n_sc.z * FF_NLOG2E, n_sc.w * FF_NLOG2E);
      m_sh = make_float4(n_sh.x * FF_NLOG2E, n_sh.y * FF_NLOG2E, n_sh.z * FF_NLOG2E, n_sh.w * FF_NLOG2E);
    }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      // (slots past the patch - possible in the last j only - repeat patch pixel 323: same data to the same address)
      const int pix = (j * PPJ + PPJ - 1 < FF_NPATCH) ? j * PPJ + lp0 : min(j * PPJ + lp0, FF_NPATCH - 1);
      const int dt = dtab[pix];                      // LDS byte offset of the pixel record; bit 31: the pixel lies outside the image
      const bool in = dt >= 0;
      float h[4] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w};
      if constexpr (NORM) {
        h[0] = h[0] * m_sc.x + m_sh.x; h[1] = h[1] * m_sc.y + m_sh.y;
        h[2] = h[2] * m_sc.z + m_sh.z; h[3] = h[3] * m_sc.w + m_sh.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = h[q] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h[q]));      // (v_exp_f32, v_rcp_f32: 1 ulp)
      }
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = in ? h[q] : 0.f;            // padding is applied to the ACTIVATED tensor: exactly 0
      char* const rec = buf + (dt & 0x7fffffff);
      const int hp0 = ff_pack_f16(v[0], v[1]), hp1 = ff_pack_f16(v[2], v[3]);
      *reinterpret_cast<int2*>(rec + lg * 8) = make_int2(hp0, hp1);
      if constexpr (NS == 2) {
        const float l0 = ff_lo<false>(hp0, v[0]), l1 = ff_lo<true>(hp0, v[1]), l2 = ff_lo<false>(hp1, v[2]), l3 = ff_lo<true>(hp1, v[3]);
        if constexpr (F8) {
          // e4m3 has no infinity: beyond 448 a conversion produces NaN - unless MODE.FP16_OVFL is set (kernel entry), which makes
          // every fp8 conversion saturate at +-448 (tools/probes/cvt_probe.hip): no clamp instructions.  The 2^11 scaling of the lo part
          // is the scale operand of v_cvt_scalef32_pk_fp8_f32 (it divides by the scale); the e4m3 "hi" operand is taken from v
          // itself (3 mantissa bits either way).  The conversions' pass-through operand is dead data (both halves are written).
          short2v l8 = __builtin_bit_cast(short2v, __float_as_int(h[0]));
          l8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l8, l0, l1, 1.0f / 2048.0f, false);
          l8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l8, l2, l3, 1.0f / 2048.0f, true);
          int h8 = __float_as_int(h[1]);
          h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], h8, false);
          h8 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], h8, true);
          *reinterpret_cast<int*>(rec + 32 + lg * 4) = __builtin_bit_cast(int, l8);
          *reinterpret_cast<int*>(rec + 48 + lg * 4) = h8;
        } else {
          *reinterpret_cast<int2*>(rec + 32 + lg * 8) = make_int2(ff_pack_f16(l0, l1), ff_pack_f16(l2, l3));
        }
      }
    }
  };

  // ---- weight stream (wave 0): group G -> ring slot G % R, GL LDS-DMA instructions of 1 KiB ----
  const int total_groups = k_nstage * GPS;
  const char* const wsrc = g_wpack + (size_t)ng * ((size_t)(Cin / 16) * 9 * SB) + lane * 16;
  // DMA wave w owns pieces w*GLW .. w*GLW+GLW-1 of every ring group
  auto issue_w1 = [&](int G, int slot, int piece) __attribute__((always_inline)) {
    const int i = wave * GLW + piece;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)G * GB + i * 1024),
                                     (__attribute__((address_space(3))) void*)(ring + slot * GB + i * 1024), 16, 0, 0);
  };
  auto issue_w = [&](int G, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < GLW; ++i) issue_w1(G, slot, i);
  };

  // ---- accumulators: (bias + temb + residual) * 2^8 ----
  // The PIXELS are the MFMA's M operand, the couts its N operand: a lane holds ONE cout (nt*32 + p32) of 16 pixels of each M tile -
  // register r = pixel 8 (r / 4) + 4 kh + r % 4 of the tile's 32 (4 rows x 8 columns) - so every residual load and output store of a
  // wave covers whole 128-byte lines (tools/probes/store_probe.hip: 5.4 TB/s against 3.3 TB/s for the weights-as-M layout, whose lanes own
  // 16-byte pieces that four different instructions assemble into a line), and the GroupNorm partials are in-lane sums.
  // The residual (the block's shortcut) is requested FIRST, straight into the accumulator registers, before the weight DMAs and the
  // first patch: ONE HBM round trip for all three (it used to be loaded one M tile at a time behind the patch: two more round trips,
  // +50-70 us per launch at 160^2).  Being the oldest entries of the in-order vmcnt queue, the residual has landed wherever the
  // patch / the first weight groups have; the scaling to the accumulator domain follows the first patch conversion.
  const int c_lane = ng * NT * 32 + p32;
  floatx16 acc[2][NT];
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = a_bias ? a_bias[c_lane + nt * 32] : 0.f;
  if (a_temb) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] += a_temb[(size_t)b * a_temb_stride + c_lane + nt * 32];
  }
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  const size_t tile_pix = img0 + (size_t)ty0 * kW + tx0;
  // output / residual element of register r of M tile mt: pixel row 4 wave + r / 4, column 8 mt + 4 kh + r % 4, cout c_lane + 32 nt
  // (buffer addressing: the lane part - K half and cout - is ONE voffset register per tensor, the register's pixel a scalar soffset)
  auto upix = [&](int mt, int r) __attribute__((always_inline)) { return (4 * wave + (r >> 2)) * kW + 8 * mt + (r & 3); };      // uniform
  const bool has_res = a_res != nullptr && !FF_ABL(8);
  if (has_res) {
    const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_res + tile_pix * kCout), 0, OOB, RSRC_FLAGS);
    const unsigned res_voff = (unsigned)(4 * kh * kCout + c_lane) * 4u;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[mt][nt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, res_voff, (unsigned)(upix(mt, r) * kCout + nt * 32) * 4u, 0));
  } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  }

  // ---- prologue: the first stage's patch (waves 1-3) and the first R-1 weight groups (wave 0) are requested before anything
  // else, the tables / bias / time-embedding work runs under their latency ----
  if (wave < NDMA) {
#pragma unroll
    for (int G = 0; G < R - 1; ++G)
      if (G < total_groups) issue_w(G, G);
  } else {
    issue_patch(0, std::false_type{});
  }
  // staging tables (the pixel -> source / destination map is the same for every stage)
  for (int pix = tid; pix < FF_NPATCH; pix += FF_THREADS) {
    const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
    const int y = ty0 - 1 + pr, x = tx0 - 1 + pc;
    stab[pix] = (y >= 0 && y < kH && x >= 0 && x < kW) ? y * kW + x : -1;      // zero padding outside THIS sample
    dtab[pix] = (pr * FF_RS + pc * FF_PSB) | (stab[pix] < 0 ? (int)0x80000000 : 0);
  }


  // per-lane LDS offset of tap (0,0) of its pixel in M tile mt (rows 4*wave + (p32 >> 3), cols 8*mt + (p32 & 7)) + K half
  int base[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) base[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + kh * 16;

  int base8[2];                                      // F8: the pixel record itself (the lane's K half selects a TAP there, not a channel half)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) base8[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + 32;

  ff_barrier();                                      // tables visible
  FF_TS();
  if (wave < NDMA) ff_wait_vm<(R - 3) * GLW>();      // groups 0 and 1 have landed (and the older residual loads)
  else store_patch(patch);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float b_in = bv[nt] * acc_in;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], acc_in, b_in);      // (residual or 0) * 2^8 [/ -ln2] + (bias + temb) * ...
  }
  ff_barrier();
  FF_TS();

  // Invariant at the top of group G (after its barrier): groups G and G+1 are in LDS, so every wave may fetch the NEXT K step's
  // fragments - also across a ring-group boundary - before it issues the current step's MFMAs (register double buffer).
  int Gc = 0, slot = 0;                              // current group and its ring slot
  for (int s = 0; s < k_nstage; ++s) {
    const bool W0 = wave < NDMA;                     // (a weight-DMA wave)
    const bool more = s + 1 < k_nstage;
    if (!W0 && more && !FF_ABL(4)) issue_patch(s + 1, std::true_type{});
    const char* const pb = patch + (s & 1) * FF_PATCH_BYTES;
    constexpr int NP = F8 ? 1 : NS;                  // fp16 planes read per fragment
    half8 wa[2][NT][NP], xb[2][2][NP];
    auto load_frags = [&](int buf, int step, int sl) __attribute__((always_inline)) {     // step: compile-time after unrolling
      const int ksub = step / 9, tap = step - ksub * 9;
      const int r = tap / 3, sx = tap - r * 3;
      const char* const wb = ring + sl * GB + (step % TG) * SB + lane * 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) wa[buf][nt][pl] = *reinterpret_cast<const half8*>(wb + (nt * NS + pl) * 1024);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          xb[buf][mt][pl] = *reinterpret_cast<const half8*>(pb + base[mt] + r * FF_RS + sx * FF_PSB + (NS == 1 ? ksub * 32 : pl * 32));
    };
    // F8: operands of the correction MFMA of the tap pair (tap, tap + 1): lanes 0-31 carry tap, lanes 32-63 tap + 1 (whose ring
    // group has landed: the invariant below); the unpaired last tap multiplies zeros in the upper K half
    int8v wa8[NT], xb8[2];
    auto load_frags8 = [&](int tap, int sl) __attribute__((always_inline)) {
      const bool pair = tap + 1 < 9;
      const int t1 = pair ? tap + 1 : tap;
      const int o0 = (tap / 3) * FF_RS + (tap % 3) * FF_PSB, o1 = (t1 / 3) * FF_RS + (t1 % 3) * FF_PSB;
      const int sl1 = pair ? (sl + 1 == R ? 0 : sl + 1) : sl;
      const char* const wb = ring + (kh ? sl1 : sl) * GB + p32 * 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4f a = *reinterpret_cast<const uint4f*>(wb + (nt * 2 + 1) * 1024);
        const uint4f c = *reinterpret_cast<const uint4f*>(wb + (nt * 2 + 1) * 1024 + 512);
        const bool z = !pair && kh;
        wa8[nt] = int8v{z ? 0 : (int)a.x, z ? 0 : (int)a.y, z ? 0 : (int)a.z, z ? 0 : (int)a.w,
                        z ? 0 : (int)c.x, z ? 0 : (int)c.y, z ? 0 : (int)c.z, z ? 0 : (int)c.w};
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const char* const xp = pb + base8[mt] + (kh ? o1 : o0);
        const uint4f a = *reinterpret_cast<const uint4f*>(xp);
        const uint4f c = *reinterpret_cast<const uint4f*>(xp + 16);
        xb8[mt] = int8v{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)c.x, (int)c.y, (int)c.z, (int)c.w};
      }
    };
    load_frags(0, 0, slot);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
      const int cur = step & 1;
      const bool first = step % TG == 0, last = step % TG == TG - 1;
      if (first && W0) {                             // refill the slot group Gc-1 has just left
        // (measured: the pieces issued one by one between the MFMAs of the group, or split over two DMA waves - no faster)
        const int Gn = Gc + R - 1;
        int sn = slot + R - 1;
        sn = sn >= R ? sn - R : sn;
        if (Gn < total_groups && !FF_ABL(1)) issue_w(Gn, sn);
      }
      // the first accumulator's MFMAs, THEN the next step's fragment reads (they issue and complete under the remaining
      // MFMAs of this step), then the rest: a wave's instruction stream stalls on the matrix pipe, so reads placed after
      // the MFMA block would start ~160 cycles late, and reads placed before it would be waited for at once (lgkmcnt(0)).
      // MFMAs on ONE accumulator stay back to back (tools/probes/mfma_chain.hip: 2465 TF/s against 2218 round-robin).
      auto mma = [&](int mt, int nt) __attribute__((always_inline)) {
        if constexpr (NS == 2 && !F8) {              // small terms first: lo*hi, hi*lo, then hi*hi
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xb[cur][mt][0], wa[cur][nt][1], acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xb[cur][mt][1], wa[cur][nt][0], acc[mt][nt], 0, 0, 0);
        }
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xb[cur][mt][0], wa[cur][nt][0], acc[mt][nt], 0, 0, 0);
      };
      __builtin_amdgcn_sched_barrier(0);
      FF_FS(step * 5 + 0);
      mma(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      FF_FS(step * 5 + 1);
      if (step + 1 < STEPS && !FF_ABL(64)) {
        int sl = slot;
        if (last) sl = slot + 1 == R ? 0 : slot + 1;
        load_frags(cur ^ 1, step + 1, sl);
      }
      const bool corr = F8 && (step % 2 == 0);       // taps 0, 2, 4, 6 bring their right neighbour, tap 8 goes alone
      if constexpr (F8) {
        if (corr) load_frags8(step, slot);           // (read under the remaining fp16 MFMAs of this step)
      }
      __builtin_amdgcn_sched_barrier(0);
      FF_FS(step * 5 + 2);
#pragma unroll
      for (int i = 1; i < 2 * NT; ++i) {
        mma(i / NT, i % NT);
        if (NS == 2) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (F8) {
#pragma unroll
        for (int i = 0; i < (corr ? 2 * NT : 0); ++i) {
          acc[i / NT][i % NT] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xb8[i / NT], wa8[i % NT], acc[i / NT][i % NT], 0, 0, 0,
                                                                                  127, 0, 116);      // block scale 2^-11 on B (the weights)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      FF_FS(step * 5 + 3);
      if (last) {
        if (W0) {                                    // group Gc+2 has landed before anyone passes the barrier into group Gc+1
          if (Gc + R - 1 < total_groups) ff_wait_vm<(R - 3) * GLW>();
          else ff_wait_vm<0>();
        } else if (step == STEPS - 1 && more && !FF_ABL(2)) {
          FF_TS();
          store_patch(patch + ((s + 1) & 1) * FF_PATCH_BYTES);
        }
        FF_FS(step * 5 + 4);
        if (!FF_ABL(32)) ff_barrier();
        if (step == STEPS - 1) FF_TS();
        ++Gc;
        slot = slot + 1 == R ? 0 : slot + 1;
      }
    }
  }

  // ---- epilogue ----
  const float wunscale = acc_out;
  const __amdgpu_buffer_rsrc_t out_r =
      __builtin_amdgcn_make_buffer_rsrc(a_out + tile_pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
  if (FF_ABL(16)) return;
  const unsigned out_voff = (unsigned)(4 * kh * a_out_stride + c_lane) * 4u;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = acc[mt][nt][r] * wunscale * a_out_scale;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][nt][r]), out_r, out_voff, (unsigned)(upix(mt, r) * a_out_stride + nt * 32) * 4u, 0);

  FF_TS();
  // ---- GroupNorm partials of the written tile: (sum, sum of squares) per cout over its 256 pixels ----
  if (a_stats) {
    // in-lane over a lane's 32 pixels (16 registers x 2 M tiles), one exchange between the K halves, the four waves through LDS
    float* const red = reinterpret_cast<float*>(smem);       // [4 waves][NT*32 couts][2] (the patch buffers are dead)
    ff_barrier();                                      // everyone is done with the patch / ring (no fence: __syncthreads() would wait for the
                                                       // output stores above to be acknowledged - vmcnt(0) - before the statistics start)
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
  FF_TS();
  FF_WALL(15);
#undef FF_TS
#undef FF_WALL
#undef FF_FS
#undef FF_ABL
}

long long* g_ff_dbg = nullptr;      // tuning aid: per-workgroup phase stamps (csd_debug_ff_timing + CSD_FF_ABL bit 7)

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// cout tiles per workgroup: 3 (96 couts: the nf = 96 nets) or 2 (64 couts: the nf = 128 nets - four tiles need 256 VGPRs + spills)
static inline int ff_nt(int cout) { return cout % 96 == 0 ? 3 : 2; }

// Maps that 16 does not divide (the 40^2 level of the 160^2 nets) run on RAGGED tiles - in conv_xk.hip alone: its epilogue masks the lanes
// whose 4 x 8-pixel block lies outside the image (8 | H, W), the patch geometry pads with zeros as at any image border.  Worth it while
// the tiles are at least 65 % full (40^2: 9 tiles of which 6.25 are work; the alternative is the quad kernel behind a gn_apply16 pass).
static bool ff_ragged_ok(const ConvPlan& p, int ns) {
  const int nstage = (p.C0 + p.C1) / 16;
  if (CSD_TUNE_ENV("CSD_XP_OPS_OFF") || CSD_TUNE_ENV("CSD_NO_RAGGED")) return false;
  const int th = (p.OH + FF_TILE - 1) / FF_TILE * FF_TILE, tw = (p.OW + FF_TILE - 1) / FF_TILE * FF_TILE;
  return ns == 2 && p.Cout % 96 == 0 && nstage >= 4 && nstage % 2 == 0 && p.OH % 8 == 0 && p.OW % 8 == 0 && p.OH > FF_TILE && p.OW > FF_TILE &&
         (double)p.OH * p.OW >= 0.65 * th * tw;
}
static bool ff_ragged(const ConvPlan& p) { return p.OH % FF_TILE != 0 || p.OW % FF_TILE != 0; }

bool convff_supported(const ConvPlan& p, int ns) {
  if (CSD_TUNE_ENV("CSD_NO_FF")) return false;
  const int kc = ns == 1 ? 32 : 16;
  return (ns >= 1 && ns <= 3) && p.taps == 9 && p.stride == 1 && p.up == 0 && p.pad == 1 && p.C0 > 0 && p.C0 % kc == 0 &&
         p.C1 % kc == 0 && (p.Cout % 96 == 0 || (p.Cout % 64 == 0 && !CSD_TUNE_ENV("CSD_FF_NO_NT2"))) &&
         (!ff_ragged(p) || ff_ragged_ok(p, ns)) && p.IH == p.OH && p.IW == p.OW;
}

bool convff_pipelined(const ConvPlan& p, int ns) {
  const int nstage = (p.C0 + p.C1) / 16;
  return ns == 2 && convff_supported(p, ns) && nstage >= 4 && nstage % 2 == 0 && !CSD_TUNE_ENV("CSD_XP_OPS_OFF");
}

// the pipelined layers run in the Winograd F(2,3) form on conv_xk.hip (96- and 64-cout groups; transformed weights, 4 components x 3
// filter rows instead of 9 taps)
bool convff_winograd(const ConvPlan& p, int ns) { return convff_pipelined(p, ns); }

size_t convff_packed_bytes(const ConvPlan& p, int ns) {
  const int nt = ff_nt(p.Cout);
  if (convff_winograd(p, ns)) return (size_t)(p.Cout / (32 * nt)) * ((p.C0 + p.C1) / 16) * 12 * nt * 2 * 1024 + 4096;
  return (size_t)(p.Cout / (32 * nt)) * ((p.C0 + p.C1) / 16) * 9 * nt * (ns == 3 ? 2 : ns) * 1024 + 4096;
}

// weights in A-fragment order of v_mfma_f32_32x32x16_f16: [cout group][cin / 16][tap][cout tile][plane][lane][8 halves],
// lane = (k half << 5) | cout row, scaled by 2^8 (exact) so the lo plane stays out of the fp16 subnormals
__global__ void convff_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int layout, int cin_src,
                                   int cout_src, int cout_off, int Cin, int Cout, int ns, int nt, uint32_t* __restrict__ slack) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slack && idx < 1024) slack[idx] = 0u;          // the 4096-byte prefetch slack behind the fragments (see convff_pack_weight)
  const size_t total = (size_t)cout_src * Cin * 9;
  if (idx >= total) return;
  const int tap = (int)(idx % 9);
  const int cin = (int)((idx / 9) % Cin);
  const int co = (int)(idx / ((size_t)9 * Cin));
  const int cout = cout_off + co;
  if (cin >= cin_src || cout >= Cout) return;
  const float v = ((layout == 0) ? w[((size_t)co * cin_src + cin) * 9 + tap]
                   : (layout == 1) ? w[(size_t)cin * cout_src + co]
                                   : w[((size_t)cin * cout_src + co) * 9 + (8 - tap)]) * C16_WSCALE;
  const int gc = 32 * nt;
  const int ng = cout / gc, t = (cout % gc) / 32, row = cout % 32;
  const int kb = cin / 16, khalf = (cin % 16) / 8, e = cin % 8;
  const size_t step = ((size_t)ng * (Cin / 16) + kb) * 9 + tap;
  const int planes = ns == 3 ? 2 : ns;                // ns = 3: fp16 hi plane + one plane of e4m3 correction operands
  _Float16* dst = wpack + ((step * nt + t) * (size_t)planes) * 512 + (khalf * 32 + row) * 8 + e;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  if (ns == 2) dst[512] = (_Float16)(v - (float)hi);
  if (ns == 3) {                                      // [hi e4m3 | lo * 2^11 e4m3][cout row][16 channels]: a lane's two 16-byte halves sit 512 bytes
                                                      // apart, rows 16 bytes apart - conflict-free ds_read_b128 (32-byte rows: 2-way conflicts,
                                                      // SQ_LDS_BANK_CONFLICT = 29 % of the kernel's LDS cycles)
    const float hf = fminf(fmaxf((float)hi, -448.f), 448.f), lf = fminf(fmaxf((v - (float)hi) * 2048.f, -448.f), 448.f);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(hf, lf, 0, false);
    unsigned char* c8 = reinterpret_cast<unsigned char*>(wpack + ((step * nt + t) * (size_t)planes + 1) * 512) + row * 16 + (cin % 16);
    c8[0] = (unsigned char)(pk & 255);
    c8[512] = (unsigned char)((pk >> 8) & 255);
  }
}

// conv_xk.hip's weights: G0 = g0, G1 = (g0 + g1 + g2) / 2, G2 = (g0 - g1 + g2) / 2, G3 = g2 of every filter row (g0, g1, g2), in
// [cout group][cin / 16][filter row][component][cout tile][hi | lo][lane][8 halves]; transform in double, one rounding to fp32, x 2^8
__global__ void convxw_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int layout, int cin_src, int cout_src,
                                   int cout_off, int Cin, int Cout, int nt, uint32_t* __restrict__ slack) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slack && idx < 1024) slack[idx] = 0u;
  const size_t total = (size_t)cout_src * Cin * 3;
  if (idx >= total) return;
  const int r = (int)(idx % 3);
  const int cin = (int)((idx / 3) % Cin);
  const int co = (int)(idx / ((size_t)3 * Cin));
  const int cout = cout_off + co;
  if (cin >= cin_src || cout >= Cout) return;
  double g[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int tap = r * 3 + t;
    g[t] = (layout == 0) ? (double)w[((size_t)co * cin_src + cin) * 9 + tap] : (double)w[((size_t)cin * cout_src + co) * 9 + (8 - tap)];
  }
  const double G[4] = {g[0], 0.5 * (g[0] + g[1] + g[2]), 0.5 * (g[0] - g[1] + g[2]), g[2]};
  const int gc = 32 * nt;
  const int ng = cout / gc, t = (cout % gc) / 32, row = cout % 32;
  const int kb = cin / 16, khalf = (cin % 16) / 8, e = cin % 8;
  const size_t step = ((size_t)ng * (Cin / 16) + kb) * 3 + r;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const float v = (float)G[kc] * C16_WSCALE;
    _Float16* dst = wpack + (((step * 4 + kc) * nt + t) * (size_t)2) * 512 + (khalf * 32 + row) * 8 + e;
    const _Float16 hi = (_Float16)v;
    dst[0] = hi;
    dst[512] = (_Float16)(v - (float)hi);
  }
}

__global__ void convff_zero_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

int convff_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                       void* wpack, hipStream_t s) {
  const int Cin = p.C0 + p.C1;
  // (the packed layout has no padding in cin or cout: zero it only when this call does not cover all of it - a weight that arrives
  // in several cout slices or with fewer input channels than the layer)
  if (cout_off == 0 && !(cin_src == Cin && cout_src == p.Cout)) {
    const size_t n32 = convff_packed_bytes(p, ns) / 4;
    hipLaunchKernelGGL(convff_zero_kernel, dim3((unsigned)cdiv64(n32, 256)), dim3(256), 0, s, (uint32_t*)wpack, n32);
    CSD_LAUNCH_CHECK();
  }
  // full coverage: only the 4096-byte slack behind the fragments is zeroed, by the pack kernel's first 1024 threads (the weight
  // streams prefetch past the last step; nothing multiplies it today, but it must never hold NaN patterns a future schedule could
  // consume - round-4 advisor finding; no extra launch: the training graph repacks every weight every step)
  uint32_t* const slack = cout_off == 0 ? (uint32_t*)((char*)wpack + convff_packed_bytes(p, ns) - 4096) : nullptr;
  if (convff_winograd(p, ns)) {
    CSD_REQUIRE(layout == 0 || layout == 2, "convff: the Winograd pack takes 3x3 weights");
    const size_t total3 = (size_t)cout_src * Cin * 3;
    hipLaunchKernelGGL(convxw_pack_kernel, dim3((unsigned)cdiv64(total3, 256)), dim3(256), 0, s, w, (_Float16*)wpack, layout, cin_src,
                       cout_src, cout_off, Cin, p.Cout, ff_nt(p.Cout), slack);
    CSD_LAUNCH_CHECK();
    return CSD_OK;
  }
  const size_t total = (size_t)cout_src * Cin * 9;
  hipLaunchKernelGGL(convff_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, (_Float16*)wpack, layout,
                     cin_src, cout_src, cout_off, Cin, p.Cout, ns, ff_nt(p.Cout), slack);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int convff_plan_tiles(ConvPlan* p, int ns) {
  CSD_REQUIRE(convff_supported(*p, ns), "convff: unsupported shape (Cin=%d+%d Cout=%d %dx%d)", p->C0, p->C1, p->Cout, p->OH, p->OW);
  p->KC = C16_KC;
  p->CoutPad = p->Cout;
  p->NT = ff_nt(p->Cout);
  p->n_groups = p->Cout / (32 * p->NT);
  p->KCS = ns == 1 ? 2 : 1;
  p->LC = 0;
  p->MT = 0;
  p->TH = p->TW = FF_TILE;
  p->PH = p->PW = FF_PW;
  p->tiles_x = cdiv(p->OW, FF_TILE);
  p->tiles_y = p->B * cdiv(p->OH, FF_TILE);
  p->lds_bytes = 0;
  return CSD_OK;
}

template <int NS, int NT, bool F8, bool NORM>
static int launch_ff(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_ff_kernel<NS, NT, F8, NORM>;
  CSD_SET_MAX_LDS_ONCE(kern);
  {
    static bool occ_shown = false;
    if (!occ_shown && CSD_TUNE_ENV("CSD_FF_OCC")) {
      occ_shown = true;
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), FF_THREADS, FFCfg<NS, NT>::LDS);
      fprintf(stderr, "conv_ff<%d,%d>: %d workgroups per CU (LDS %zu B)\n", NS, NT, nb, (size_t)FFCfg<NS, NT>::LDS);
    }
  }
  size_t lds = FFCfg<NS, NT>::LDS;
  if (CSD_TUNE_ENV("CSD_FF_LDS_PAD")) lds += (size_t)atoi(CSD_TUNE_ENV("CSD_FF_LDS_PAD"));      // tuning aid: forces one workgroup per CU
  hipLaunchKernelGGL(kern, dim3(k.nblocks), dim3(FF_THREADS), lds, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int convff_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s) {
  CSD_REQUIRE(convff_supported(p, ns), "convff: unsupported layer");
  CSD_REQUIRE(!a.out_nchw && a.out_stride % 4 == 0 && a.out_coff % 4 == 0 && a.src0 && (p.C1 == 0 || a.src1),
              "convff: NHWC fp32 output with 16-byte aligned rows, fp32 NHWC sources");
  CSD_REQUIRE((a.nscale == nullptr) == (a.nshift == nullptr), "convff: scale and shift come together");
  CSD_REQUIRE(a.nscale == nullptr || a.act == CSD_ACT_SWISH, "convff: the fused prologue implements GroupNorm + SiLU");
  ConvFFArgs k;
  k.a = a;
  k.B = p.B; k.H = p.OH; k.W = p.OW; k.C0 = p.C0; k.C1 = p.C1; k.Cout = p.Cout;
  k.tiles_x = cdiv(p.OW, FF_TILE);
  k.tpi = cdiv(p.OH, FF_TILE) * k.tiles_x;
  k.n_groups = p.Cout / (32 * ff_nt(p.Cout));
  k.nblocks = p.B * k.tpi * k.n_groups;
  k.nstage = (p.C0 + p.C1) / (ns == 1 ? 32 : 16);
  k.abl = CSD_TUNE_ENV("CSD_FF_ABL") ? atoi(CSD_TUNE_ENV("CSD_FF_ABL")) : 0;
  k.a.dbg = (k.abl & 128) ? g_ff_dbg : nullptr;
  const int nt = ff_nt(p.Cout);
  // conv_fx.hip (one persistent 8-wave workgroup per CU: four matrix waves + four producer waves, whole-stage weight buffers): 4-5 %
  // faster than this kernel on the layers with a long K loop (>= 192 input channels: the up path's Conv_0 on the virtual concat), at
  // parity or slower on the 96-channel ones (profiles/NOTEBOOK.md).  It carries one residual chunk in each of its first 2 nt stages.
  // Tuning build: CSD_FX=1 forces it for every fp16f8 layer, CSD_FX=0 disables it.
  {
    const char* fx = CSD_TUNE_ENV("CSD_FX");
    const bool use_fx = fx ? atoi(fx) != 0 : k.nstage >= 12;
    if (ns == 3 && use_fx && k.nstage >= 2 * nt) return convfx_launch(k, nt, s);
  }
  // conv_xk.hip: the fp16x3 form of every pipelined layer (1-D Winograd F(2,3), one persistent 4-wave workgroup per CU; the weights were
  // packed for it: convff_winograd is the one switch)
  if (convff_winograd(p, ns)) {
    CSD_REQUIRE(convxk_supported(k, nt), "convff: Winograd layer outside conv_xk's range");
    return convxk_launch(k, nt, s);
  }
  const bool norm = a.nscale != nullptr;
#define FF_DISPATCH(NT_)                                                                                            \
  if (ns == 1) return norm ? launch_ff<1, NT_, false, true>(k, s) : launch_ff<1, NT_, false, false>(k, s);       \
  if (ns == 3) return norm ? launch_ff<2, NT_, true, true>(k, s) : launch_ff<2, NT_, true, false>(k, s);         \
  return norm ? launch_ff<2, NT_, false, true>(k, s) : launch_ff<2, NT_, false, false>(k, s);
  if (nt == 2) { FF_DISPATCH(2) }
  FF_DISPATCH(3)
#undef FF_DISPATCH
}

}  // namespace csd

// tuning aid: buf = [4096 workgroups][2 waves][16] int64 clock stamps, filled by launches with CSD_FF_ABL bit 7 set
extern "C" int csd_debug_ff_timing(void* buf) {
  csd::g_ff_dbg = static_cast<long long*>(buf);
  return 0;
}
