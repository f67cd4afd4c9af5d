// conv_f32.hip - 3x3 / 1x1 convolution as an implicit GEMM on the gfx950 fp32 matrix cores.
//
// Replaces (reference, behaviour only): models/layers.py:119-132 ddpm_conv3x3, :100-105
// ddpm_conv1x1, :555-564 NIN, :593-604 Upsample (nearest x2 fused into the gather), :607-629
// Downsample ((0,1,0,1) pad + stride 2 fused), and the elementwise glue of ResnetBlockDDPM
// (:658-675): the GroupNorm-affine + SiLU *prologue* runs while the source patch is staged into
// LDS, and bias + time-embedding + residual are the *epilogue*.
//
// Mapping (one workgroup = 4 waves = 256 threads):
//   M tile : TH x TW (<=128) output pixels of the "virtual tall image" [B*OH, OW] - image
//            borders are handled by a per-lane 9-bit tap-validity mask, so a tile may straddle
//            images (needed for the 5x5 / 10x10 / 20x20 levels).
//   N tile : NT x 32 output channels; wave w owns pixels [32w, 32w+32) x all NT*32 channels.
//   K      : Cin in chunks of KC channels x `taps`; per chunk the (PH x PW x KC) source patch is
//            staged ONCE into LDS (halo reuse across the 9 taps), double buffered, global->reg
//            loads issued before the MFMA block and written to LDS after it.
//   MFMA   : v_mfma_f32_32x32x2_f32 (A: 32 pixels x 2 channels, B: 2 channels x 32 couts).
//            Exact fp32 (bitwise an fmaf chain), 157 TFLOP/s peak on MI355X.
//   B operand: weights are pre-packed in fragment order, so each lane fetches one 16-byte
//            vector per 4 MFMAs straight from L2/L1 (1 KiB per wave-instruction, fully coalesced)
//            - no LDS traffic for weights; the stream is linear and prefetched one step ahead.
#include <stdlib.h>

#include "common.h"

namespace csd {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef CSD_CONV_ABLATE
#define CSD_CONV_ABLATE 0   // tuning aid: 1 no B prefetch loads, 2 no A LDS reads, 4 no staging/barriers, 8 no MFMA
#endif
#define CONV_THREADS 256
#define CONV_MAX_SLOTS 9     // float4 staging slots per thread per chunk
#define CONV_PAD 4           // floats of padding per staged pixel (LDS bank spread)

struct ConvKArgs {
  ConvArgs a;
  int B, IH, IW, OH, OW, C0, C1, Cout;
  int stride, pad, up;
  int TH, TW, PH, PW, tiles_x, n_groups, nblocks;
  int nck;                  // number of K chunks
};

// explicit global-address-space 16-byte load: keeps hipcc from falling back to flat_load (which
// also ticks lgkmcnt and would serialise the LDS reads behind the weight prefetch)
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const float* p) {
  const v4f_t v = *(const __attribute__((address_space(1))) v4f_t*)(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    // v_exp_f32 + v_rcp_f32 (each <= 1 ulp): ~6 VALU ops instead of ~35 for expf + IEEE divide; the
    // result differs from the libm form by < 1e-6 relative - this runs on every staged element
    case CSD_ACT_SWISH: return v * __frcp_rn(1.0f + __expf(-v));
    case CSD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CSD_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case CSD_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

template <int NT, int TAPS, int KC, int SLOTS>
__global__ __launch_bounds__(CONV_THREADS, 1) void conv_f32_kernel(const float* __restrict__ g_src0,
                                                                const float* __restrict__ g_src1,
                                                                const float* __restrict__ g_wpack,
                                                                const ConvKArgs k) {
  constexpr int KS = (TAPS == 9) ? 3 : 1;
  constexpr int PS = KC + CONV_PAD;          // floats per staged pixel
  constexpr int N4 = KC / 4;                 // float4 per staged pixel
  constexpr int KK = KC / 8;                 // 8-channel MFMA groups per chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;

  // ---- XCD-aware block -> work mapping: consecutive work items (same pixel tile, different
  // cout group) stay on one XCD so its L2 serves the shared source patch (guide T1) ----
  int w;
  {
    const int bid = blockIdx.x, nb = k.nblocks;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ng = w % k.n_groups;
  const int tile = w / k.n_groups;
  const int tile_y = tile / k.tiles_x;
  const int tile_x = tile - tile_y * k.tiles_x;
  const int ov0 = tile_y * k.TH;             // first virtual output row of the tile
  const int ox0 = tile_x * k.TW;

  const int S = k.stride, P = k.pad, U = k.up;
  const int prow0 = (ov0 * S - P) >> U;      // arithmetic shift == floor
  const int pcol0 = (ox0 * S - P) >> U;
  const int patch_floats = k.PH * k.PW * PS;
  float* const buf0 = smem;
  float* const buf1 = smem + patch_floats;
  int* const otab = reinterpret_cast<int*>(smem + 2 * patch_floats);   // [128] output pixel index
  int* const btab = otab + 128;                                        // [128] batch index

  // ---- per-lane pixel (A-operand row) ----
  int off[TAPS];
  unsigned vmask = 0;
  {
    const int m = wave * 32 + (lane & 31);
    const int ty = m / k.TW;
    const int tx = m - ty * k.TW;
    const int ov = ov0 + ty, ox = ox0 + tx;
    const bool mv = (m < k.TH * k.TW) && (ov < k.B * k.OH) && (ox < k.OW);
    const int b = ov / k.OH;
    const int oy = ov - b * k.OH;
    if (half == 0) {
      otab[m] = mv ? (ov - ov0) * k.OW + ox : -1;   // relative to the tile's first row
      btab[m] = mv ? b : 0;
    }
    const int IHe = k.IH << U, IWe = k.IW << U;
#pragma unroll
    for (int r = 0; r < KS; ++r) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int iy = oy * S + r - P, ix = ox * S + s - P;
        const bool v = mv && iy >= 0 && iy < IHe && ix >= 0 && ix < IWe;
        const int pr = ((ov * S + r - P) >> U) - prow0;
        const int pc = ((ox * S + s - P) >> U) - pcol0;
        off[r * KS + s] = v ? (pr * k.PW + pc) * PS + half * 4 : 0;
        vmask |= (v ? 1u : 0u) << (r * KS + s);
      }
    }
  }

  // ---- staging slots: which float4 of the patch this thread moves (same for every chunk) ----
  const int total4 = k.PH * k.PW * N4;
  const int Cin = k.C0 + k.C1;

  // Per-thread staging slots are the same for every K chunk: precompute, per slot, the source pixel
  // (or -1 outside the image), the LDS destination and the sample index (for the norm lookup).
  int s_pix[SLOTS], s_lds[SLOTS], s_nrm[SLOTS];
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    const int e = tid + j * CONV_THREADS;
    s_pix[j] = -1; s_lds[j] = -1; s_nrm[j] = 0;
    if (e < total4) {
      const int pix = e / N4;
      const int c4 = e - pix * N4;
      const int pr = pix / k.PW;
      const int pc = pix - pr * k.PW;
      const int vr = prow0 + pr, col = pcol0 + pc;
      s_lds[j] = pix * PS + c4 * 4;
      if (vr >= 0 && vr < k.B * k.IH && col >= 0 && col < k.IW) {
        s_pix[j] = vr * k.IW + col;
        s_nrm[j] = (vr / k.IH) * Cin + c4 * 4;
      }
    }
  }
  const int my_c4 = (tid % N4) * 4;   // CONV_THREADS % N4 == 0: same channel group in every slot

  // stage_load only ISSUES the global loads (raw values stay in flight under the MFMA block);
  // stage_write applies the GroupNorm affine + activation and stores to LDS afterwards.
  float4 stage[SLOTS];
  auto stage_load = [&](int ck) {
    const int cb = ck * KC;
    const float* src;
    int Cs, coff;
    if (cb < k.C0) { src = g_src0; Cs = k.C0; coff = cb; }
    else { src = g_src1; Cs = k.C1; coff = cb - k.C0; }
    src += coff + my_c4;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s_pix[j] >= 0) v = gload4(src + (size_t)s_pix[j] * Cs);
      stage[j] = v;
    }
  };
  auto stage_write = [&](float* buf, int ck) {
    const int cb = ck * KC;
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      if (s_lds[j] >= 0) {
        float4 v = stage[j];
        if (k.a.nscale && s_pix[j] >= 0) {
          const float4 sc = *reinterpret_cast<const float4*>(k.a.nscale + s_nrm[j] + cb);
          const float4 sh = *reinterpret_cast<const float4*>(k.a.nshift + s_nrm[j] + cb);
          v.x = act_apply(v.x * sc.x + sh.x, k.a.act);
          v.y = act_apply(v.y * sc.y + sh.y, k.a.act);
          v.z = act_apply(v.z * sc.z + sh.z, k.a.act);
          v.w = act_apply(v.w * sc.w + sh.w, k.a.act);
        }
        *reinterpret_cast<float4*>(buf + s_lds[j]) = v;
      }
    }
  };

  // ---- accumulators and the linear weight stream ----
  floatx16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  const size_t steps_per_tile = (size_t)k.nck * TAPS * KK;     // 1 KiB (256 floats) per step
  const size_t tile_stride = steps_per_tile * 256;             // floats between consecutive cout tiles
  const float* wstep = g_wpack + (size_t)(ng * NT) * tile_stride;   // wave-uniform, advances 256 floats/step
  const int lane4 = lane * 4;

  // two-stage register pipeline, statically indexed (everything below is fully unrolled):
  // while the 4*NT MFMAs of step s run, the B fragments (global, L2-resident) and the A fragment
  // (LDS) of step s+1 are already in flight.  sched_barrier pins that order - left alone, hipcc
  // sinks every load next to its first use and exposes a full L2 round trip per 4 MFMAs.
  float4 breg[2][NT];
  float4 areg[2];
#pragma unroll
  for (int n = 0; n < NT; ++n) breg[0][n] = gload4(wstep + n * tile_stride + lane4);

  stage_load(0);
  stage_write(buf0, 0);
  __syncthreads();

  constexpr int STEPS = TAPS * KK;                // steps per chunk
  for (int ck = 0; ck < k.nck; ++ck) {
    const float* buf = (ck & 1) ? buf1 : buf0;
    if (!(CSD_CONV_ABLATE & 4) && ck + 1 < k.nck) stage_load(ck + 1);       // global loads fly under the MFMA block
    areg[0] = *reinterpret_cast<const float4*>(buf + off[0]);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int cur = st & 1, nxt = cur ^ 1;
      const int tap = st / KK;
      // ---- issue the loads of step st+1 ----
      wstep += 256;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (CSD_CONV_ABLATE & 1) breg[nxt][n] = breg[cur][n];
        else breg[nxt][n] = gload4(wstep + n * tile_stride + lane4);   // 1 step of slack past the end
      }
      if (st + 1 < STEPS) {
        const int tap1 = (st + 1) / KK, kk1 = (st + 1) % KK;
        if (CSD_CONV_ABLATE & 2) areg[nxt] = areg[cur];
        else areg[nxt] = *reinterpret_cast<const float4*>(buf + off[tap1] + kk1 * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMAs of step st ----
      float4 a4 = areg[cur];
      if (!((vmask >> tap) & 1u)) a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (CSD_CONV_ABLATE & 8) { acc[n][0] += a4.x * breg[cur][n].x + a4.y * breg[cur][n].y + a4.z * breg[cur][n].z + a4.w * breg[cur][n].w; continue; }
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, breg[cur][n].x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, breg[cur][n].y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, breg[cur][n].z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, breg[cur][n].w, acc[n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (STEPS & 1) {   // odd step count: keep the ping-pong phase aligned across chunks
#pragma unroll
      for (int n = 0; n < NT; ++n) breg[0][n] = breg[1][n];
    }
    if (!(CSD_CONV_ABLATE & 4)) {
      if (ck + 1 < k.nck) stage_write((ck & 1) ? buf0 : buf1, ck + 1);
      __syncthreads();
    }
  }

  // ---- epilogue: bias + time embedding + residual, NHWC (or NCHW) store ----
  // Straight-line and branch-free: accesses go through buffer descriptors based at the tile's first
  // output row; a lane whose pixel / cout does not exist uses an out-of-range offset (load returns 0,
  // store is dropped).  Per-element branches made every access wait on vmcnt(0) - which on gfx9 also
  // counts stores - so the stores of a wave completed one memory round trip at a time.
  const int ohw = k.OH * k.OW;
  constexpr unsigned OOB = 0x80000000u;            // >= num_records of every descriptor below
  constexpr int RSRC_FLAGS = 0x00020000;           // raw dword buffer, gfx9 encoding
  const size_t o_base = (size_t)ov0 * k.OW;
  const int b0 = ov0 / k.OH;                       // first sample the tile touches
  const bool has_res = k.a.res != nullptr, has_temb = k.a.temb != nullptr, nchw = k.a.out_nchw != 0;
  float* const out_base = nchw ? k.a.out + (size_t)b0 * k.Cout * ohw : k.a.out + o_base * k.a.out_stride + k.a.out_coff;
  const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(out_base, 0, OOB, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_res ? k.a.res + o_base * k.Cout : k.a.out), 0, OOB, RSRC_FLAGS);
  const int pix0 = (int)(o_base - (size_t)b0 * ohw);
  int oidx[16], bidx[16];                          // pixel index relative to o_base (or -1), sample index
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    oidx[r] = otab[m];
    bidx[r] = btab[m];
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = (ng * NT + n) * 32 + (lane & 31);
    const bool cv = col < k.Cout;
    const int colc = cv ? col : 0;
    const float bv = k.a.bias ? k.a.bias[colc] : 0.f;
    float addv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) addv[r] = 0.f;
    if (has_res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = (cv && oidx[r] >= 0) ? (unsigned)(oidx[r] * k.Cout + col) * 4u : OOB;
        addv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(res_r, off, 0, 0));
      }
    }
    if (has_temb) {
      float tv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) tv[r] = k.a.temb[(size_t)bidx[r] * k.a.temb_stride + colc];
#pragma unroll
      for (int r = 0; r < 16; ++r) addv[r] += tv[r];
    }
    if (nchw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = oidx[r], db = bidx[r] - b0;
        const float val = ((acc[n][r] + bv) + addv[r]) * k.a.out_scale;
        const unsigned off = (cv && o >= 0) ? (unsigned)((db * k.Cout + col) * ohw + (pix0 + o - db * ohw)) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), out_r, off, 0, 0);
      }
    } else {
      double st_s = 0.0, st_q = 0.0;       // GroupNorm partials of this lane's column over the wave's 32 pixels
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float val = ((acc[n][r] + bv) + addv[r]) * k.a.out_scale;
        const unsigned off = (cv && oidx[r] >= 0) ? (unsigned)(oidx[r] * k.a.out_stride + col) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), out_r, off, 0, 0);
        const double dv = oidx[r] >= 0 ? (double)val : 0.0;
        st_s += dv;
        st_q = fma(dv, dv, st_q);
      }
      // statistics for the GroupNorm that reads this tensor next (see conv_f16_kernel.h): one (sum, sumsq)
      // pair per (tile, wave, cout); the host sets `stats` only when a tile lies inside one sample
      if (k.a.stats) {
        st_s += __shfl_xor(st_s, 32);
        st_q += __shfl_xor(st_q, 32);
        if (lane < 32 && cv) {
          double* dst = k.a.stats + (((size_t)tile * 4 + wave) * k.Cout + col) * 2;
          dst[0] = st_s;
          dst[1] = st_q;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int in_coord(int o, int r, int S, int P, int U) {
  int v = o * S + r - P;
  return v >= 0 ? (v >> U) : -((-v + (1 << U) - 1) >> U);   // floor
}

int conv_plan_tiles(ConvPlan* p) {
  const int Cin = p->C0 + p->C1;
  CSD_REQUIRE(p->taps == 1 || p->taps == 9, "conv: taps must be 1 or 9 (got %d)", p->taps);
  CSD_REQUIRE(Cin % 8 == 0 && p->C0 % 8 == 0, "conv: Cin (%d+%d) must be a multiple of 8", p->C0, p->C1);
  CSD_REQUIRE((p->IH << p->up) == p->OH * p->stride && (p->IW << p->up) == p->OW * p->stride,
              "conv: size mismatch IH=%d IW=%d OH=%d OW=%d stride=%d up=%d", p->IH, p->IW, p->OH,
              p->OW, p->stride, p->up);
  p->KC = (p->C0 % 16 == 0 && p->C1 % 16 == 0) ? 16 : 8;
  // 1x1: one K chunk is only 4*NT MFMAs per 16 channels - far less than an HBM round trip - so stage
  // 32 channels per chunk: 2x fewer dependent memory latencies per workgroup
  if (p->taps == 1 && p->C0 % 32 == 0 && p->C1 % 32 == 0) p->KC = 32;   // (256 threads % (KC/4) must be 0)
  const int ntiles = cdiv(p->Cout, 32);
  p->CoutPad = ntiles * 32;
  // tile: TW divides OW when possible; maximise covered pixels, then minimise the staged patch.
  // Patch extents are the worst case over tile origins (the +1 covers odd origins in `up` mode).
  const int KS = p->taps == 9 ? 3 : 1;
  auto extent = [&](int t) {
    return in_coord(t - 1, KS - 1, p->stride, p->pad, p->up) - in_coord(0, 0, p->stride, p->pad, p->up) + 1 +
           (p->up ? 1 : 0);
  };
  int best_tw = 0, best_th = 0, best_cov = -1, best_patch = 1 << 30;
  for (int tw = 1; tw <= 32 && tw <= p->OW; ++tw) {
    if (p->OW % tw != 0 && !(tw == 32 && p->OW > 32)) continue;
    for (int th = 128 / tw; th >= 1; --th) {
      const int patch = extent(th) * extent(tw);
      if (patch * (p->KC / 4) > CONV_MAX_SLOTS * CONV_THREADS) continue;
      const int cov = th * tw;
      if (cov > best_cov || (cov == best_cov && patch < best_patch)) {
        best_cov = cov; best_patch = patch; best_tw = tw; best_th = th;
      }
      break;   // smaller th only lowers coverage for this tw
    }
  }
  CSD_REQUIRE(best_tw > 0, "conv: no feasible tile for OW=%d", p->OW);
  p->TW = best_tw;
  p->TH = best_th;
  p->PH = extent(p->TH);
  p->PW = extent(p->TW);
  p->tiles_x = cdiv(p->OW, p->TW);
  p->tiles_y = cdiv(p->B * p->OH, p->TH);
  // cout tiles per workgroup: 3 amortises the staged patch best, but the low-resolution levels
  // (5x5 / 10x10 / 20x20) have so few pixel tiles that the chip is latency-bound on one
  // workgroup's serial K loop - there, narrower workgroups (more of them) win.
  {
    const int slots = 256 * 3;                    // resident workgroups (256 CUs x 3)
    const int tiles = p->tiles_x * p->tiles_y;
    double best = 1e30;
    int best_nt = 1;
    for (int nt = 3; nt >= 1; --nt) {
      if (ntiles % nt) continue;
      const int nwg = tiles * (ntiles / nt);
      const double t = (double)cdiv(nwg, slots) * nt * (1.0 + 0.05 * (3 - nt));
      if (t < best - 1e-9) { best = t; best_nt = nt; }
    }
    if (const char* f = CSD_TUNE_ENV("CSD_FORCE_NT")) {   // tuning aid
      const int v = atoi(f);
      if (v >= 1 && v <= 3 && ntiles % v == 0) best_nt = v;
    }
    p->NT = best_nt;
    p->n_groups = ntiles / p->NT;
  }
  p->lds_bytes = (size_t)2 * p->PH * p->PW * (p->KC + CONV_PAD) * sizeof(float) + 256 * sizeof(int);
  CSD_REQUIRE(p->PH * p->PW * (p->KC / 4) <= CONV_MAX_SLOTS * CONV_THREADS, "conv: patch too large");
  CSD_REQUIRE(p->lds_bytes <= 160 * 1024, "conv: LDS budget exceeded");
  return CSD_OK;
}

size_t conv_packed_floats(const ConvPlan& p) {
  const int Cin = p.C0 + p.C1;
  // + one step (256 floats) of slack per tensor: the kernel prefetches one step past the end
  return (size_t)(p.CoutPad / 32) * (Cin / 8) * p.taps * 256 + 256;
}

__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wpack, int layout,
                                 int cin_src, int cout_src, int cout_off, int taps, int KC, int nck,
                                 int nt_lo, int nt_hi) {
  // one thread per packed float of the n-tiles [nt_lo, nt_hi)
  const int KK = KC / 8;
  const size_t per_tile = (size_t)nck * taps * KK * 256;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = per_tile * (nt_hi - nt_lo);
  if (idx >= total) return;
  size_t rem = idx;
  const int nt = nt_lo + (int)(rem / per_tile); rem %= per_tile;
  const int ck = (int)(rem / ((size_t)taps * KK * 256)); rem %= (size_t)taps * KK * 256;
  const int tap = (int)(rem / (KK * 256)); rem %= (size_t)KK * 256;
  const int kk = (int)(rem / 256); rem %= 256;
  const int lane = (int)(rem / 4), q = (int)(rem % 4);
  const int cout = nt * 32 + (lane & 31) - cout_off;
  const int cin = ck * KC + kk * 8 + (lane >> 5) * 4 + q;
  // padding rows/columns stay zero: the buffer is cleared before the first tensor is packed
  if (cout >= 0 && cout < cout_src && cin < cin_src) {
    // layout 0: OIHW; 1: NIN [Cin][Cout]; 2: the OIHW weight of the TRANSPOSED convolution ([Cin][Cout][taps]), spatially flipped
    // (the data gradient of a convolution is a convolution with that weight)
    const float v = (layout == 0) ? w[((size_t)cout * cin_src + cin) * taps + tap]
                    : (layout == 1) ? w[(size_t)cin * cout_src + cout]
                                    : w[((size_t)cin * cout_src + cout) * taps + (taps - 1 - tap)];
    wpack[(size_t)nt * per_tile + (idx % per_tile)] = v;
  }
}

__global__ void conv_pack_zero_kernel(float* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

int conv_pack_weight(const ConvPlan& p, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                     float* wpack, hipStream_t s) {
  const int Cin = p.C0 + p.C1;
  const int nck = Cin / p.KC;
  const size_t nflt = conv_packed_floats(p);
  if (cout_off == 0) {   // first (or only) tensor of this packed buffer: clear padding + slack
    hipLaunchKernelGGL(conv_pack_zero_kernel, dim3((unsigned)cdiv64(nflt, 256)), dim3(256), 0, s, wpack, nflt);
    CSD_LAUNCH_CHECK();
  }
  const int nt_lo = cout_off / 32, nt_hi = cdiv(cout_off + cout_src, 32);
  const size_t total = (size_t)nck * p.taps * (p.KC / 8) * 256 * (nt_hi - nt_lo);
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, wpack, layout,
                     cin_src, cout_src, cout_off, p.taps, p.KC, nck, nt_lo, nt_hi);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

template <int NT, int TAPS, int KC, int SLOTS>
static int launch_one(const ConvKArgs& k, size_t lds, int nblocks, hipStream_t s) {
  auto kern = conv_f32_kernel<NT, TAPS, KC, SLOTS>;
  static bool attr_set = false;   // per instantiation
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(CONV_THREADS), lds, s, k.a.src0, k.a.src1, k.a.wpack, k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int conv_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t s) {
  ConvKArgs k;
  k.a = a;
  k.B = p.B; k.IH = p.IH; k.IW = p.IW; k.OH = p.OH; k.OW = p.OW;
  k.C0 = p.C0; k.C1 = p.C1; k.Cout = p.Cout;
  k.stride = p.stride; k.pad = p.pad; k.up = p.up;
  k.TH = p.TH; k.TW = p.TW; k.PH = p.PH; k.PW = p.PW;
  k.tiles_x = p.tiles_x; k.n_groups = p.n_groups;
  k.nblocks = p.tiles_x * p.tiles_y * p.n_groups;
  k.nck = (p.C0 + p.C1) / p.KC;
  const int slots = cdiv(p.PH * p.PW * (p.KC / 4), CONV_THREADS);
#define CSD_CONV_CASE(NT_, TAPS_, KC_)                                                              \
  if (p.NT == NT_ && p.taps == TAPS_ && p.KC == KC_) {                                              \
    if (slots <= 3) return launch_one<NT_, TAPS_, KC_, 3>(k, p.lds_bytes, k.nblocks, s);            \
    if (slots <= 4) return launch_one<NT_, TAPS_, KC_, 4>(k, p.lds_bytes, k.nblocks, s);            \
    return launch_one<NT_, TAPS_, KC_, CONV_MAX_SLOTS>(k, p.lds_bytes, k.nblocks, s);               \
  }
  CSD_CONV_CASE(1, 9, 8) CSD_CONV_CASE(2, 9, 8) CSD_CONV_CASE(3, 9, 8)
  CSD_CONV_CASE(1, 9, 16) CSD_CONV_CASE(2, 9, 16) CSD_CONV_CASE(3, 9, 16)
  CSD_CONV_CASE(1, 1, 8) CSD_CONV_CASE(2, 1, 8) CSD_CONV_CASE(3, 1, 8)
  CSD_CONV_CASE(1, 1, 16) CSD_CONV_CASE(2, 1, 16) CSD_CONV_CASE(3, 1, 16)
  CSD_CONV_CASE(1, 1, 32) CSD_CONV_CASE(2, 1, 32) CSD_CONV_CASE(3, 1, 32)
#undef CSD_CONV_CASE
  set_error("conv: no kernel for NT=%d taps=%d KC=%d", p.NT, p.taps, p.KC);
  return CSD_ERR_INVALID;
}

}  // namespace csd
