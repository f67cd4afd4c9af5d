// conv_ffp.hip - persistent producer / consumer schedule of the fused-prologue block convolution (same arithmetic, weight layout, LDS
// patch layout and fragment maps as conv_ff.hip; reference layers: models/layers.py:632-675, models/layerspp.py:212-274).
//
// Why a second schedule.  Phase stamps of conv_ff_kernel (tools/ff_timing.py, split mode, 96 -> 96 at 160^2): a tile costs a
// workgroup 134k cycles of which 31k are MFMA issue - the rest is the prologue (first patch from HBM, 12k), the conversion passes at
// the stage ends (6 x 4k, three waves), the epilogue (residual loads, stores, statistics: 20k) and MFMA phases that run at a
// third of the matrix rate because the two co-resident workgroups take turns on every SIMD; each of those phases is only
// overlapped by whatever the single partner workgroup happens to be doing.
//
// Here ONE workgroup of EIGHT waves owns a CU for the whole launch and the overlap is structural:
//   * waves 0-3 (one per SIMD) are the CONSUMERS: fragment reads + MFMAs only; each has its SIMD's matrix pipe to itself.
//   * waves 4-6 are the PATCH producers: they request stage S+2's pixels from HBM (two register sets: two stages of latency
//     cover), convert stage S+1 (GroupNorm affine + SiLU + fp16 hi|lo split) a few slots per ring group into the other LDS patch
//     buffer - under the consumers' MFMAs, across tile boundaries: the next tile's first stage is converted during this tile's
//     last, so there is no per-tile prologue.
//   * wave 7 is the WEIGHT producer: the six LDS-DMA pieces of a ring group, a few groups ahead; nothing else is in its
//     in-order memory queue.
//   * the residual is added INTO the accumulators (x 2^8) after the first two stages, one M tile each, so the epilogue has no
//     load to wait for: scale, bias + time embedding, 16-byte stores, per-tile GroupNorm partials (DPP butterfly).
//   One workgroup barrier per ring group (18 MFMAs per consumer wave in the split mode, 12 in fp16 mode) + one per tile.
#include "conv_ff.h"

namespace csd {

#define FFP_THREADS 512

template <int NS, int NT>
struct FFPCfg : FFCfg<NS, NT> {
  using B = FFCfg<NS, NT>;
  static constexpr int RP = 6;                                 // ring depth (groups)
  static constexpr int PTHREADS = 192;                         // patch producers (waves 4-6): the conversion is ~450 cycles per slot and wave
  static constexpr int NSLOT = (FF_NPATCH * B::G4 + PTHREADS - 1) / PTHREADS;
  static constexpr int PPJ = PTHREADS / B::G4;                 // patch pixels between a thread's consecutive slots
  static constexpr int GLW = B::GL;                            // LDS-DMA pieces per group of the weight wave (wave 7)
  static constexpr size_t LDS = 2 * (size_t)FF_PATCH_BYTES + (size_t)RP * B::GB + 4 * NT * 32 * 2 * sizeof(float);
};

template <int NS, int NT>
__global__ __launch_bounds__(FFP_THREADS, 1) void conv_ffp_kernel(const char* __restrict__ g_wpack, const ConvFFArgs k) {
  using C = FFPCfg<NS, NT>;
  constexpr int KC = C::KC, STEPS = C::STEPS, TG = C::TG, GPS = C::GPS, SB = C::SB, GB = C::GB, GL = C::GL, R = C::RP;
  constexpr int G4 = C::G4, NSLOT = C::NSLOT, PPJ = C::PPJ, GLW = C::GLW;
  constexpr int PPS = NS == 1 ? 2 : 1;             // residual pieces per stage (the fp16 mode has half as many stages)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;                                         // 2 buffers
  char* const ring = smem + 2 * FF_PATCH_BYTES;                     // R groups of weights in fragment order
  float* const red = reinterpret_cast<float*>(ring + R * GB);       // [4 consumer waves][NT*32 couts][2]

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
            k_n_groups = k.n_groups, k_nstage = k.nstage, k_abl = k.abl;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cin = kC0 + kC1;
  const bool norm = a_nscale != nullptr;

  // ---- this workgroup's items (tile, cout group): XCD-aware - all items of a workgroup lie in the contiguous range of ITS XCD ----
  const int nitems = k.nblocks;
  const int G = gridDim.x;
  const int my_items = (nitems - (int)blockIdx.x + G - 1) / G;
  auto item_of = [&](int i) __attribute__((always_inline)) -> int {
    const int v = blockIdx.x + i * G;
    const int xcd = v & 7, slot = v >> 3;
    const int q = nitems >> 3, r = nitems & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  };
  const int total_stages = my_items * k_nstage;
  const int groups_per_item = k_nstage * GPS;
  const int total_groups = my_items * groups_per_item;
  if (my_items <= 0) return;

  if (wave == 7) {
    // ================================================ WEIGHT PRODUCER (wave 7) =================================================
    const int wv = 0;
    const size_t wstride = (size_t)(Cin / 16) * 9 * SB;              // bytes of one cout group's weight stream
    int it_g = 0, it_i = 0;                                          // (group within item, item) of the NEXT group to request
    const char* wsrc = g_wpack + (size_t)(item_of(0) % k_n_groups) * wstride + (wv * GLW) * 1024 + lane * 16;
    int islot = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
      char* dst = ring + islot * GB + (wv * GLW) * 1024;
#pragma unroll
      for (int i = 0; i < GLW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (size_t)it_g * GB + i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      islot = islot + 1 == R ? 0 : islot + 1;
      if (++it_g == groups_per_item) {
        it_g = 0;
        ++it_i;
        if (it_i < my_items) wsrc = g_wpack + (size_t)(item_of(it_i) % k_n_groups) * wstride + (wv * GLW) * 1024 + lane * 16;
      }
    };
    int issued = 0;
    for (; issued < R - 1 && issued < total_groups; ++issued) issue_next();
    ff_wait_vm<(R - 3) * GLW>();                     // groups 0 and 1 have landed
    ff_barrier();                                    // (prologue barrier)
    for (int Gc = 0; Gc < total_groups; ++Gc) {
      if (issued < total_groups && !(k_abl & 1)) { issue_next(); ++issued; ff_wait_vm<(R - 3) * GLW>(); }
      else ff_wait_vm<0>();
      if (!(k_abl & 32)) ff_barrier();               // group barrier
      if ((Gc + 1) % groups_per_item == 0) ff_barrier();      // tile barrier (statistics hand-over of the consumers)
    }
    return;
  }

  if (wave >= 4) {
    // ============================================ PATCH PRODUCERS (waves 4-6) ==============================================
    const int pt = tid - 256;
    const int lg = pt % G4, lp0 = pt / G4;
    float4 pf[2][NSLOT];
    float4 n_sc[2], n_sh[2];
    unsigned vmask[2] = {0u, 0u};
    n_sc[0] = n_sc[1] = make_float4(1.f, 1.f, 1.f, 1.f);
    n_sh[0] = n_sh[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue_patch = [&](int S, auto set_tag) __attribute__((always_inline)) {      // request stage S into register set SET
      constexpr int SET = decltype(set_tag)::value;
      const int it = S / k_nstage, s = S - it * k_nstage;
      const int tile = item_of(it) / k_n_groups;
      const int b = tile / k_tpi, tin = tile - b * k_tpi;
      const int ty0 = (tin / k_tiles_x) * FF_TILE, tx0 = (tin - (tin / k_tiles_x) * k_tiles_x) * FF_TILE;
      const int cb = s * KC;
      const bool s1 = cb >= kC0;
      const size_t img0 = (size_t)b * kH * kW;
      const float* src = (s1 ? a_src1 : a_src0) + img0 * (s1 ? kC1 : kC0) + (s1 ? cb - kC0 : cb) + lg * 4;
      const int Cs = s1 ? kC1 : kC0;
      unsigned vm = 0;
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        const int pix = min(j * PPJ + lp0, FF_NPATCH - 1);
        const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
        const int y = ty0 - 1 + pr, x = tx0 - 1 + pc;
        const bool in = y >= 0 && y < kH && x >= 0 && x < kW;       // zero padding outside THIS sample
        vm |= (in ? 1u : 0u) << j;
        pf[SET][j] = gload4f(src + (size_t)(in ? y * kW + x : 0) * Cs);
      }
      vmask[SET] = vm;
      if (norm) {
        n_sc[SET] = gload4f(a_nscale + (size_t)b * Cin + cb + lg * 4);
        n_sh[SET] = gload4f(a_nshift + (size_t)b * Cin + cb + lg * 4);
      }
    };
    auto convert_slot = [&](int j, auto set_tag, char* buf) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_tag)::value;
      const int pixr = j * PPJ + lp0;
      const int pix = min(pixr, FF_NPATCH - 1);
      const bool in = (vmask[SET] >> j) & 1u;
      float h[4] = {pf[SET][j].x, pf[SET][j].y, pf[SET][j].z, pf[SET][j].w};
      if (norm) {
        h[0] = h[0] * n_sc[SET].x + n_sh[SET].x; h[1] = h[1] * n_sc[SET].y + n_sh[SET].y;
        h[2] = h[2] * n_sc[SET].z + n_sh[SET].z; h[3] = h[3] * n_sc[SET].w + n_sh[SET].w;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = h[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-h[q]));
      }
      half4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = in ? h[q] : 0.f;             // padding is applied to the ACTIVATED tensor: exactly 0
        hi[q] = (_Float16)v;
        lo[q] = (_Float16)(v - (float)hi[q]);
      }
      const bool real = (j * PPJ + PPJ - 1 < FF_NPATCH) || pixr < FF_NPATCH;      // (slots past the patch: only in the last j)
      const int pr = pix / FF_PW, pc = pix - pr * FF_PW;
      char* dst = buf + (real ? pr * FF_RS + pc * FF_PSB + lg * 8 : FF_PATCH_BYTES - 16);
      *reinterpret_cast<half4*>(dst) = hi;
      if (NS == 2) *reinterpret_cast<half4*>(dst + (real ? 32 : 8)) = lo;
    };
    // prologue: stages 0 and 1 requested, stage 0 converted
    issue_patch(0, std::integral_constant<int, 0>{});
    if (total_stages > 1) issue_patch(1, std::integral_constant<int, 1>{});
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) convert_slot(j, std::integral_constant<int, 0>{}, patch);
    ff_barrier();                                    // (prologue barrier)
    // stage S: request S+2 into the set stage S used, convert S+1 (the other set) into patch[(S+1)&1], a few slots per group
    auto stage = [&](int S, auto par_tag) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_tag)::value;                // = S & 1
      const int s = S % k_nstage;
      char* const nbuf = patch + (PAR ^ 1) * FF_PATCH_BYTES;
      const bool have1 = S + 1 < total_stages;
#pragma unroll
      for (int g = 0; g < GPS; ++g) {
        if (g == 0 && S + 2 < total_stages && !(k_abl & 4)) issue_patch(S + 2, std::integral_constant<int, PAR>{});
        if (have1 && !(k_abl & 2)) {
#pragma unroll
          for (int j = g; j < NSLOT; j += GPS) convert_slot(j, std::integral_constant<int, PAR ^ 1>{}, nbuf);
        }
        if (!(k_abl & 32)) ff_barrier();             // group barrier
      }
      if (s == k_nstage - 1) ff_barrier();           // tile barrier
    };
    for (int S = 0; S < total_stages; S += 2) {
      stage(S, std::integral_constant<int, 0>{});
      if (S + 1 < total_stages) stage(S + 1, std::integral_constant<int, 1>{});
    }
    return;
  }

  // ================================================= CONSUMERS (waves 0-3) =================================================
  const int kh = lane >> 5, p32 = lane & 31;
  floatx16 acc[2][NT];
  int base[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) base[mt] = (4 * wave + (p32 >> 3)) * FF_RS + (8 * mt + (p32 & 7)) * FF_PSB + kh * 16;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int RSRC_FLAGS = 0x00020000;
  const float wunscale = 1.0f / C16_WSCALE;
  const bool has_res = a_res != nullptr;

  ff_barrier();                                      // (prologue barrier)
  int slot = 0;
  for (int it = 0; it < my_items; ++it) {
    const int item = item_of(it);
    const int ng = item % k_n_groups, tile = item / k_n_groups;
    const int b = tile / k_tpi, tin = tile - b * k_tpi;
    const int ty0 = (tin / k_tiles_x) * FF_TILE, tx0 = (tin - (tin / k_tiles_x) * k_tiles_x) * FF_TILE;
    const int c_lane = ng * NT * 32 + kh * 4;
    const size_t tile_pix = (size_t)b * kH * kW + (size_t)ty0 * kW + tx0;
    const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_res ? a_res + tile_pix * kCout : a_out), 0, OOB, RSRC_FLAGS);
    int opix[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) opix[mt] = (4 * wave + (p32 >> 3)) * kW + 8 * mt + (p32 & 7);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int s = 0; s < k_nstage; ++s) {
      const int S = it * k_nstage + s;
      const char* const pb = patch + (S & 1) * FF_PATCH_BYTES;
      // the residual goes INTO the accumulators (x 2^8), one (M tile, cout tile) piece - 16 registers - at a time: piece p is
      // requested at the top of stage p / PPS and added after that stage's MFMAs (pieces the K loop is too short for: epilogue)
      float4 rv[PPS][4];
      const bool radd = has_res && s * PPS < 2 * NT;
      if (radd) {
#pragma unroll
        for (int j = 0; j < PPS; ++j) {
          const int pc = min(s * PPS + j, 2 * NT - 1);
          const int mt_ = pc / NT, nt_ = pc - mt_ * NT;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned off = (unsigned)((mt_ ? opix[1] : opix[0]) * kCout + c_lane + nt_ * 32 + q * 8) * 4u;
            const uint4f u = __builtin_amdgcn_raw_buffer_load_b128(res_r, off, 0, 0);
            rv[j][q] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
          }
        }
      }
      half8 wa[2][NT][NS], xb[2][2][NS];
      auto load_frags = [&](int buf, int step, int sl) __attribute__((always_inline)) {     // step: compile-time after unrolling
        const int ksub = step / 9, tap = step - ksub * 9;
        const int r = tap / 3, sx = tap - r * 3;
        const char* const wb = ring + sl * GB + (step % TG) * SB + lane * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int pl = 0; pl < NS; ++pl) wa[buf][nt][pl] = *reinterpret_cast<const half8*>(wb + (nt * NS + pl) * 1024);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int pl = 0; pl < NS; ++pl)
            xb[buf][mt][pl] = *reinterpret_cast<const half8*>(pb + base[mt] + r * FF_RS + sx * FF_PSB + (NS == 1 ? ksub * 32 : pl * 32));
      };
      load_frags(0, 0, slot);
#pragma unroll
      for (int step = 0; step < STEPS; ++step) {
        const int cur = step & 1;
        const bool last = step % TG == TG - 1;
        auto mma = [&](int mt, int nt) __attribute__((always_inline)) {
          if constexpr (NS == 2) {                   // small terms first: lo*hi, hi*lo, then hi*hi
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][nt][1], xb[cur][mt][0], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][nt][0], xb[cur][mt][1], acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cur][nt][0], xb[cur][mt][0], acc[mt][nt], 0, 0, 0);
        };
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (step + 1 < STEPS && !(k_abl & 64)) {
          int sl = slot;
          if (last) sl = slot + 1 == R ? 0 : slot + 1;
          load_frags(cur ^ 1, step + 1, sl);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < 2 * NT; ++i) {
          mma(i / NT, i % NT);
          if (NS == 2) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (last) {
          if (!(k_abl & 32)) ff_barrier();           // group barrier
          slot = slot + 1 == R ? 0 : slot + 1;
        }
      }
      if (radd) {
#pragma unroll
        for (int j = 0; j < PPS; ++j) {
          const int pc = s * PPS + j;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if (pc == mt * NT + nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  acc[mt][nt][q * 4 + 0] += rv[j][q].x * C16_WSCALE; acc[mt][nt][q * 4 + 1] += rv[j][q].y * C16_WSCALE;
                  acc[mt][nt][q * 4 + 2] += rv[j][q].z * C16_WSCALE; acc[mt][nt][q * 4 + 3] += rv[j][q].w * C16_WSCALE;
                }
              }
        }
      }
    }
    if (has_res && k_nstage * PPS < 2 * NT) {        // (short K loops: Cin < 96 - the remaining pieces, latency exposed)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (mt * NT + nt >= k_nstage * PPS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned off = (unsigned)(opix[mt] * kCout + c_lane + nt * 32 + q * 8) * 4u;
              const uint4f u = __builtin_amdgcn_raw_buffer_load_b128(res_r, off, 0, 0);
              acc[mt][nt][q * 4 + 0] += __uint_as_float(u.x) * C16_WSCALE; acc[mt][nt][q * 4 + 1] += __uint_as_float(u.y) * C16_WSCALE;
              acc[mt][nt][q * 4 + 2] += __uint_as_float(u.z) * C16_WSCALE; acc[mt][nt][q * 4 + 3] += __uint_as_float(u.w) * C16_WSCALE;
            }
          }
    }

    // ---- epilogue: (acc * 2^-8 + bias + temb) * out_scale, 16-byte stores, per-tile GroupNorm partials ----
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
      const __amdgpu_buffer_rsrc_t out_r =
          __builtin_amdgcn_make_buffer_rsrc(a_out + tile_pix * a_out_stride + a_out_coff, 0, OOB, RSRC_FLAGS);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 bq = bv[nt * 4 + q];
            acc[mt][nt][q * 4 + 0] = (acc[mt][nt][q * 4 + 0] * wunscale + bq.x) * a_out_scale;
            acc[mt][nt][q * 4 + 1] = (acc[mt][nt][q * 4 + 1] * wunscale + bq.y) * a_out_scale;
            acc[mt][nt][q * 4 + 2] = (acc[mt][nt][q * 4 + 2] * wunscale + bq.z) * a_out_scale;
            acc[mt][nt][q * 4 + 3] = (acc[mt][nt][q * 4 + 3] * wunscale + bq.w) * a_out_scale;
            uint4f ov;
            ov.x = __float_as_uint(acc[mt][nt][q * 4 + 0]); ov.y = __float_as_uint(acc[mt][nt][q * 4 + 1]);
            ov.z = __float_as_uint(acc[mt][nt][q * 4 + 2]); ov.w = __float_as_uint(acc[mt][nt][q * 4 + 3]);
            const unsigned off = (unsigned)(opix[mt] * a_out_stride + c_lane + nt * 32 + q * 8) * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(ov, out_r, off, 0, 0);
          }
    }
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
      // halving butterfly over the 32 pixel lanes of a K half (see conv_ff.hip): lane bits 0, 1 by DPP quad_perm, 2, 3 by ds_bpermute
#define FFP_HALVE(XCHG, BIT, H)                                                                  \
      {                                                                                          \
        const bool up = (lane >> BIT) & 1;                                                       \
        _Pragma("unroll") for (int i = 0; i < H; ++i) {                                          \
          const float ss = up ? vs[i] : vs[i + H], ks = up ? vs[i + H] : vs[i];                  \
          const float sq = up ? vq[i] : vq[i + H], kq = up ? vq[i + H] : vq[i];                  \
          vs[i] = ks + XCHG(ss, BIT);                                                            \
          vq[i] = kq + XCHG(sq, BIT);                                                            \
        }                                                                                        \
      }
#define FFP_X_DPP(v, BIT) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (BIT) == 0 ? 0xB1 : 0x4E, 0xF, 0xF, true))
#define FFP_X_SHFL(v, BIT) __shfl_xor(v, 1 << (BIT))
      FFP_HALVE(FFP_X_DPP, 0, NV / 2)
      FFP_HALVE(FFP_X_DPP, 1, NV / 4)
      FFP_HALVE(FFP_X_SHFL, 2, NV / 8)
      FFP_HALVE(FFP_X_SHFL, 3, NV / 16)
#undef FFP_HALVE
#undef FFP_X_DPP
#undef FFP_X_SHFL
      constexpr int NF = NV / 16;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        vs[i] += __shfl_xor(vs[i], 16);
        vq[i] += __shfl_xor(vq[i], 16);
      }
      if ((lane & 16) == 0) {
        const int sel = (lane & 1) * (NV / 2) + ((lane >> 1) & 1) * (NV / 4) + ((lane >> 2) & 1) * (NV / 8) +
                        ((lane >> 3) & 1) * (NV / 16);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const int idx = sel + i;                     // = nt*16 + r
          const int nt = idx >> 4, r = idx & 15;
          const int cl = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          red[(wave * NT * 32 + cl) * 2 + 0] = vs[i];
          red[(wave * NT * 32 + cl) * 2 + 1] = vq[i];
        }
      }
    }
    ff_barrier();                                    // tile barrier: the four waves' partials are in LDS
    if (a_stats && tid < NT * 32) {
      double sd = 0.0, qd = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        sd += (double)red[(wv * NT * 32 + tid) * 2 + 0];
        qd += (double)red[(wv * NT * 32 + tid) * 2 + 1];
      }
      double* dst = a_stats + ((size_t)tile * kCout + ng * NT * 32 + tid) * 2;
      dst[0] = sd;
      dst[1] = qd;
    }
  }
}

template <int NS, int NT>
static int launch_ffp(const ConvFFArgs& k, hipStream_t s) {
  auto kern = conv_ffp_kernel<NS, NT>;
  static bool attr_set = false;
  static int ncu = 0;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int dev = 0;
    hipDeviceProp_t prop;
    CSD_CHECK_HIP(hipGetDevice(&dev));
    CSD_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    ncu = prop.multiProcessorCount;
    attr_set = true;
  }
  // one persistent workgroup per CU; with more than one item per workgroup the grid is a multiple of 8 (XCD-aware item ranges)
  int grid = k.nblocks < ncu ? k.nblocks : (ncu / 8) * 8;
  if (grid < 1) grid = 1;
  const size_t lds = FFPCfg<NS, NT>::LDS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FFP_THREADS), lds, s, reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int convffp_launch(const ConvFFArgs& k, int ns, hipStream_t s) {
  if (ns == 1) return launch_ffp<1, 3>(k, s);
  return launch_ffp<2, 3>(k, s);
}

}  // namespace csd
