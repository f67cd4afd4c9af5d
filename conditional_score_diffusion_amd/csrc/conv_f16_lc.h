// conv_f16_lc.h - loader/consumer, persistent variant of the fp16-source 3x3 stride-1 convolution
// (same arithmetic, fragment layouts, LDS patch layout and epilogue as conv_f16_kernel.h; reference
// layers: models/layers.py:ddpm_conv3x3 inside ResnetBlockDDPM, models/ddpm.py:149-213).
//
// Why a second schedule.  tools/probes/phase_timing.py + an ablation (profiles/) showed two things about the
// single-role kernel at 160x160:
//   * a wave's VMEM loads return IN ORDER (one vmcnt), so the burst that fetches the next K stage's patch
//     (HBM latency) blocks every later weight-fragment load (L2 latency) of the same wave: each stage
//     stalls its MFMAs for most of an HBM round trip;
//   * a workgroup's phases (tables -> burst -> MFMA -> epilogue) are serial, and 255 registers allow only
//     two workgroups per CU to cover for each other.
// Here the roles are split by wave.  Waves 0..nw-1 are CONSUMERS: weight fragments (their only VMEM
// loads, L2 hits) -> LDS patch reads -> MFMAs -> epilogue.  Wave nw is the LOADER: it owns every patch
// burst, writes them into a double-buffered LDS patch and builds the per-tile tables.  The workgroup is
// persistent and one LDS-only barrier per K stage (no vmcnt drain) hands buffers over, so the loader runs
// ~1.5 stages ahead ACROSS tile boundaries: the next tile's first burst is in flight during this tile's
// last MFMA stage and epilogue.
#pragma once
#include "conv_f16_kernel.h"

namespace csd {

__device__ __forceinline__ void lds_barrier() {     // LDS visibility only: VMEM loads stay in flight across it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#define C16_LC_MAXPATCH 208     // patch pixels a loader wave can stage (host: conv16_plan_tiles)

// F32: the source is the fp32 NHWC residual stream itself (two-source virtual concat allowed) and the LOADER
// applies the GroupNorm affine + activation + fp16 (hi|lo) split while it writes the patch - the
// gn_apply16 pass and its fp16 planes disappear.  Host: KCS == 2, tiles inside one sample (MASK == false).
template <int MT, int NS, bool MASK, int PWC, int KCS, bool F32>
__global__ __launch_bounds__(256, 2) void conv_f16_lc_kernel(const void* __restrict__ g_hi,
                                                            const void* __restrict__ g_lo,
                                                            const char* __restrict__ g_wpack,
                                                            const Conv16KArgs k) {
  constexpr int TAPS = 9, KS = 3;
  constexpr int LO = 32 * KCS;               // byte offset of the lo plane inside a staged pixel
  constexpr int PSB = 32 * KCS * NS + 16;    // bytes per staged pixel: [hi KCS*16 ch][lo KCS*16 ch] + pad
  constexpr int NPIX = MT * 32;
  static_assert(!F32 || (KCS == 2 && !MASK), "fused fp32 source: 32-channel stages, unmasked tiles");
  constexpr int SPP = F32 ? 4 * KCS : 2 * NS * KCS;   // 16-byte slots per patch pixel per stage (fp32: 4 channels each)
  constexpr int NUL = (C16_LC_MAXPATCH * SPP + 63) / 64;     // slots per loader lane
  constexpr int rstride = PWC * PSB;
  constexpr int SSTEPS = KCS * TAPS;         // K steps per stage (sub-chunk major, tap minor = stream order)
  constexpr int STEP_BYTES = NS * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem16[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 5;
  const int patch_bytes = k.PH * PWC * PSB;
  int* const tabs = reinterpret_cast<int*>(smem16 + 2 * patch_bytes);     // [3 tiles][otab | btab | vtab][NPIX]

  // ---- this workgroup's work items (tile, cout group), XCD-aware: every item of a workgroup maps into the
  // contiguous item range of ITS XCD (gridDim.x is a multiple of 8 whenever a workgroup has > 1 item) ----
  const int nitems = k.nblocks;
  const int G = gridDim.x;
  const int my_items = (nitems - (int)blockIdx.x + G - 1) / G;
  const int nstage = k.nck / KCS;
  const int total = my_items * nstage;
  auto item_of = [&](int i) -> int {
    const int v = blockIdx.x + i * G;
    const int xcd = v & 7, slot = v >> 3;
    const int q = nitems >> 3, r = nitems & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  };
  const int Cin = k.C0 + k.C1;       // (fp16 source: one tensor, C1 == 0)
#ifdef CSD_LC_STAGGER
  if ((int)blockIdx.x >= G / 2 && my_items > 1) {      // experiment: start the second workgroup of a CU half an item late
    const long long t0 = clock64();
    while (clock64() - t0 < CSD_LC_STAGGER) __builtin_amdgcn_s_sleep(8);
  }
#endif

  if (wave == k.nw) {
    // =========================================== LOADER ===========================================
    const int npatch = k.PH * k.PW;
    const int total4 = npatch * SPP;
    int s_prc[NUL];             // (patch row << 16) | patch col of slot j (tile-independent)
    int s_off[NUL];             // source pixel index of slot j for the current tile, or -1
#pragma unroll
    for (int j = 0; j < NUL; ++j) {
      const int e = j * 64 + lane;
      const int pix = e / SPP;
      const int pr = pix / k.PW;
      s_prc[j] = (e < total4) ? ((pr << 16) | (pix - pr * k.PW)) : -1;
    }
    float4 sv[NUL];
    int tile_b = 0;             // F32: sample of the current tile (row of the GroupNorm scale/shift table)
    float4 n_sc = make_float4(0.f, 0.f, 0.f, 0.f), n_sh = n_sc;     // F32: scale/shift of this lane's 4 channels, per stage
    auto tile_setup = [&](int item, int tb) {      // addresses of the tile's patch + its epilogue/mask tables
      const int tile = item / k.n_groups;
      const int tile_y = tile / k.tiles_x;
      const int ov0 = tile_y * k.TH, ox0 = (tile - tile_y * k.tiles_x) * k.TW;
      const int prow0 = ov0 - 1, pcol0 = ox0 - 1;
      tile_b = ov0 / k.OH;
      // (MASK == false: the tile lies inside ONE sample and halo rows of its neighbours are padding)
      const int img_lo = MASK ? 0 : (ov0 / k.OH) * k.IH;
      const int img_hi = MASK ? k.B * k.IH : img_lo + k.IH;
#pragma unroll
      for (int j = 0; j < NUL; ++j) {
        const int vr = prow0 + (s_prc[j] >> 16), col = pcol0 + (s_prc[j] & 0xffff);
        const bool in = s_prc[j] >= 0 && vr >= img_lo && vr < img_hi && col >= 0 && col < k.IW;
        s_off[j] = in ? vr * k.IW + col : -1;
      }
      int* const otab = tabs + tb * 3 * NPIX;
      for (int m = lane; m < NPIX; m += 64) {
        const int ty = m / k.TW, tx = m - ty * k.TW;
        const int ov = ov0 + ty, ox = ox0 + tx;
        const bool mv = (m < k.TH * k.TW) && (ov < k.B * k.OH) && (ox < k.OW);
        const int b = ov / k.OH, oy = ov - b * k.OH;
        unsigned vb = 0;
#pragma unroll
        for (int r = 0; r < KS; ++r) {
          const int iy = oy + r - 1, ix = ox + r - 1;
          vb |= ((mv && iy >= 0 && iy < k.IH) ? 1u : 0u) << r;
          vb |= ((mv && ix >= 0 && ix < k.IW) ? 1u : 0u) << (3 + r);
        }
        otab[m] = mv ? (ov - ov0) * k.OW + ox : -1;
        otab[NPIX + m] = mv ? b : 0;
        otab[2 * NPIX + m] = (int)vb;
      }
    };
    auto issue = [&](int stage) {                  // the whole patch of one K stage, back to back
      const int cb = stage * KCS * C16_KC;
      if constexpr (F32) {
        // slot = (pixel, 4-channel group lane % 8): the channel group of a lane is the same for all its slots
        const int gch = lane & 7;
        const bool s1 = cb >= k.C0;
        const float* src = static_cast<const float*>(s1 ? g_lo : g_hi);
        const int Cs = s1 ? k.C1 : k.C0, c = (s1 ? cb - k.C0 : cb) + gch * 4;
#pragma unroll
        for (int j = 0; j < NUL; ++j) {
          const int sp = s_off[j] >= 0 ? s_off[j] : 0;
          sv[j] = gload4f(src + (size_t)sp * Cs + c);
        }
        n_sc = gload4f(k.a.nscale + (size_t)tile_b * Cin + cb + gch * 4);
        n_sh = gload4f(k.a.nshift + (size_t)tile_b * Cin + cb + gch * 4);
      } else {
#pragma unroll
        for (int j = 0; j < NUL; ++j) {
          const int e = j * 64 + lane;
          const int sub = e % SPP;
          const int pl = sub / (2 * KCS), rest = sub - pl * (2 * KCS);
          const _Float16* plane = static_cast<const _Float16*>((NS == 2 && pl) ? g_lo : g_hi);
          const int sp = s_off[j] >= 0 ? s_off[j] : 0;        // (out-of-image slots read pixel 0 and are zeroed below)
          sv[j] = gload4f(reinterpret_cast<const float*>(plane + (size_t)sp * Cin + cb + rest * 8));
        }
      }
    };
    auto write = [&](char* buf) {
      if constexpr (F32) {
        const int gch = lane & 7;
        const float4 sc = n_sc, sh = n_sh;
#pragma unroll
        for (int j = 0; j < NUL; ++j) {
          if (s_prc[j] >= 0) {
            half4 hi, lo;
            if (s_off[j] >= 0) {
              // same arithmetic as gn_apply16_kernel: fma, fast SiLU, round-to-nearest fp16, lo = h - hi
              const float h0 = act16(sv[j].x * sc.x + sh.x, k.a.act), h1 = act16(sv[j].y * sc.y + sh.y, k.a.act);
              const float h2 = act16(sv[j].z * sc.z + sh.z, k.a.act), h3 = act16(sv[j].w * sc.w + sh.w, k.a.act);
              hi[0] = (_Float16)h0; hi[1] = (_Float16)h1; hi[2] = (_Float16)h2; hi[3] = (_Float16)h3;
              lo[0] = (_Float16)(h0 - (float)hi[0]); lo[1] = (_Float16)(h1 - (float)hi[1]);
              lo[2] = (_Float16)(h2 - (float)hi[2]); lo[3] = (_Float16)(h3 - (float)hi[3]);
            } else {          // zero padding is applied to the ACTIVATED tensor: out-of-image taps are exactly 0
#pragma unroll
              for (int q = 0; q < 4; ++q) { hi[q] = (_Float16)0.f; lo[q] = (_Float16)0.f; }
            }
            char* dst = buf + ((s_prc[j] >> 16) * PWC + (s_prc[j] & 0xffff)) * PSB + gch * 8;
            *reinterpret_cast<half4*>(dst) = hi;
            if (NS == 2) *reinterpret_cast<half4*>(dst + LO) = lo;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < NUL; ++j) {
          if (s_prc[j] >= 0) {
            const int e = j * 64 + lane;
            const int sub = e % SPP;
            const int pl = sub / (2 * KCS), rest = sub - pl * (2 * KCS);
            const float4 v = s_off[j] >= 0 ? sv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(buf + ((s_prc[j] >> 16) * PWC + (s_prc[j] & 0xffff)) * PSB + pl * LO + rest * 16) = v;
          }
        }
      }
    };
    int li = 0, ls = 0;
    if (total > 0) {
      tile_setup(item_of(0), 0);
      issue(0);
    }
    for (int g = 0; g < total; ++g) {
      write(smem16 + (g & 1) * patch_bytes);       // stage g (waits for its burst)
      if (++ls == nstage) {
        ls = 0;
        ++li;
        if (li < my_items) tile_setup(item_of(li), li % 3);
      }
      if (g + 1 < total) issue(ls);                // stage g+1: in flight while the consumers work on g
      lds_barrier();                               // barrier g: stage g visible; buffer (g+1)&1 is free after it
    }
    return;
  }

  // ============================================= CONSUMERS =============================================
  // per-lane pixels: LDS byte offset of tap (0,0) (+ this lane's K half); tile-independent
  int base[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 32 + (lane & 31);
    const int ty = m / k.TW, tx = m - ty * k.TW;
    base[mt] = (m < k.TH * k.TW) ? ty * rstride + tx * PSB + kg * 16 : kg * 16;
  }
  constexpr int BR = (NS == 1) ? 9 : 6;            // weight-fragment ring (steps), divides SSTEPS
  static_assert(SSTEPS % BR == 0, "weight ring phase");
  constexpr int RING = (NS == 1) ? 4 : 3;          // A-fragment ring
  const size_t tile_stride = (size_t)k.nck * TAPS * STEP_BYTES;
  half8 breg[BR][NS];
  floatx16 acc[MT];
  unsigned vbits[MT];
  const char* wstep = g_wpack;
  int ci = 0, cs = 0;
  int ng = 0, tile = 0, ov0 = 0;
  // decode item i and start its weight stream (BR-1 steps of fragments); the caller zeroes acc
  auto item_begin = [&](int i, int& ng_, int& tile_, int& ov0_) {
    const int item = item_of(i);
    ng_ = item % k.n_groups;
    tile_ = item / k.n_groups;
    ov0_ = (tile_ / k.tiles_x) * k.TH;
    const int wtile = min(ng_ * k.nw + wave, k.ntiles_n - 1);
    wstep = g_wpack + (size_t)wtile * tile_stride + lane * 16;
#pragma unroll
    for (int q = 0; q < BR - 1; ++q)
#pragma unroll
      for (int p = 0; p < NS; ++p) breg[q][p] = gload_h8(wstep + (size_t)q * STEP_BYTES + p * 1024);
    wstep += (size_t)(BR - 2) * STEP_BYTES;        // points at the newest prefetched step
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  };
  if (total > 0) item_begin(0, ng, tile, ov0);
  zero_acc();
#ifdef CSD_C16_TIMING
  if (k.a.dbg && tid == 0 && blockIdx.x == 0) { k.a.dbg[4094 * 8 + 0] = clock64(); k.a.dbg[4094 * 8 + 1] = wall_clock64(); }
#endif

#ifdef CSD_C16_TIMING
  int ts_n = 0;
#define LC_TSTAMP() do { if (k.a.dbg && tid == 0 && blockIdx.x < 4095 && ci == 1 && ts_n < 8) k.a.dbg[blockIdx.x * 8 + ts_n++] = clock64(); } while (0)
#else
#define LC_TSTAMP() do { } while (0)
#endif
  for (int g = 0; g < total; ++g) {
    if (cs == 0) LC_TSTAMP();
    lds_barrier();                                 // barrier g: the loader has published stage g
    LC_TSTAMP();
    const char* buf = smem16 + (g & 1) * patch_bytes;
    const int* const otab = tabs + (ci % 3) * 3 * NPIX;
    if (MASK && cs == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) vbits[mt] = (unsigned)otab[2 * NPIX + mt * 32 + (lane & 31)];
    }
    {
      half8 areg[RING][NS];
      auto load_frag = [&](int q) {      // q = s * MT + mt, s = sub * TAPS + tap (compile-time after unrolling)
        const int s_ = q / MT, mt_ = q % MT;
        const int sub_ = s_ / TAPS, tap_ = s_ % TAPS;
        const char* p = buf + base[mt_] + (tap_ / KS) * rstride + (tap_ % KS) * PSB + sub_ * 32;
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
    }
    LC_TSTAMP();
    if (++cs < nstage) continue;
    cs = 0;
    // the next item's weight fragments go out BEFORE this item's stores, so they are not queued behind them
    int ng_n = 0, tile_n = 0, ov0_n = 0;
#if !defined(CSD_LC_ABL) || CSD_LC_ABL != 1
    if (ci + 1 < my_items) item_begin(ci + 1, ng_n, tile_n, ov0_n);
#endif

    // ---- epilogue of the item (see conv_f16_kernel.h): straight-line, buffer descriptors based at the tile ----
    {
      const int* const btab = otab + NPIX;
      const int ohw = k.OH * k.OW;
      const float wunscale = 1.0f / C16_WSCALE;
      const int col = (ng * k.nw + wave) * 32 + (lane & 31);
      const bool cv = col < k.Cout;
      const int colc = cv ? col : 0;
      const float bv = k.a.bias ? k.a.bias[colc] : 0.f;
      constexpr unsigned OOB = 0x80000000u;
      constexpr int RSRC_FLAGS = 0x00020000;
      const size_t o_base = (size_t)ov0 * k.OW;
      const int b0 = ov0 / k.OH;
#if defined(CSD_LC_ABL) && CSD_LC_ABL == 2
      const bool has_res = false, has_temb = k.a.temb != nullptr, nchw = k.a.out_nchw != 0;
#else
      const bool has_res = k.a.res != nullptr, has_temb = k.a.temb != nullptr, nchw = k.a.out_nchw != 0;
#endif
      float* const out_base = nchw ? k.a.out + (size_t)b0 * k.Cout * ohw : k.a.out + o_base * k.a.out_stride + k.a.out_coff;
      const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, OOB, RSRC_FLAGS);
      const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(has_res ? k.a.res + o_base * k.Cout : k.a.out), 0, OOB, RSRC_FLAGS);
      float tvu = 0.f;
      if (!MASK && has_temb) tvu = k.a.temb[(size_t)b0 * k.a.temb_stride + colc];
      const int pix0 = (int)(o_base - (size_t)b0 * ohw);
      // EVERY global read of the epilogue is issued before the first store: vmcnt retires in order and counts
      // stores too, so a load issued after a store cannot be consumed before that store is acknowledged by
      // memory - measured (tools/probes/phase_timing_lc.py): the second half of a two-batch epilogue sat 30k cycles
      // behind the first half's 32 stores.
      auto PIX = [&](int mt, int r) { return mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg; };
      float addv[MT][16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) addv[mt][r] = tvu;
      if (has_res) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = otab[PIX(mt, r)];
            const unsigned off = (cv && o >= 0) ? (unsigned)(o * k.Cout + col) * 4u : OOB;
            const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, off, 0, 0));
            addv[mt][r] = (MASK || !has_temb) ? v : v + tvu;
          }
      }
      if (MASK && has_temb) {
        if (!has_res) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) addv[mt][r] = k.a.temb[(size_t)btab[PIX(mt, r)] * k.a.temb_stride + colc];
        } else {      // (never in the network: a layer has either a time-embedding column or a residual)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) addv[mt][r] += k.a.temb[(size_t)btab[PIX(mt, r)] * k.a.temb_stride + colc];
        }
      }
      double st_s = 0.0, st_q = 0.0;
      if (nchw) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = otab[PIX(mt, r)];
            const int db = btab[PIX(mt, r)] - b0;
            const float val = ((acc[mt][r] * wunscale + bv) + addv[mt][r]) * k.a.out_scale;
            const unsigned off = (cv && o >= 0) ? (unsigned)((db * k.Cout + col) * ohw + (pix0 + o - db * ohw)) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), out_r, off, 0, 0);
          }
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = otab[PIX(mt, r)];
            // (acc*2^-8 + bias) + temb + residual: same association as the reference's h + Dense(temb), x + h
            const float val = ((acc[mt][r] * wunscale + bv) + addv[mt][r]) * k.a.out_scale;
            const unsigned off = (cv && o >= 0) ? (unsigned)(o * k.a.out_stride + col) * 4u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), out_r, off, 0, 0);
            const double dv = (cv && o >= 0) ? (double)val : 0.0;
            st_s += dv;
            st_q = fma(dv, dv, st_q);
          }
      }
      LC_TSTAMP();
      if (k.a.stats) {      // GroupNorm partials of the written tensor (host: only when a tile lies inside one sample)
        st_s += __shfl_xor(st_s, 32);
        st_q += __shfl_xor(st_q, 32);
        if (lane < 32 && cv) {
          double* dst = k.a.stats + ((size_t)tile * k.Cout + col) * 2;
          dst[0] = st_s;
          dst[1] = st_q;
        }
      }
    }
#if defined(CSD_LC_ABL) && CSD_LC_ABL == 1
    if (ci + 1 < my_items) item_begin(ci + 1, ng_n, tile_n, ov0_n);
#endif
    LC_TSTAMP();
    ++ci;
    ng = ng_n; tile = tile_n; ov0 = ov0_n;
    zero_acc();
  }
#ifdef CSD_C16_TIMING
  if (k.a.dbg && tid == 0 && blockIdx.x == 0) { k.a.dbg[4094 * 8 + 2] = clock64(); k.a.dbg[4094 * 8 + 3] = wall_clock64(); }
#endif
}

}  // namespace csd
