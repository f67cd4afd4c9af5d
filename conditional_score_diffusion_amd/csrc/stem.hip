// stem.hip - the DDPM-family network's first layer as ONE launch (gfx950): cat(x, y [+ sigma * z]), the 2v - 1 data centering, the
// NCHW -> NHWC layout change and the 3 x 3 convolution of the <= 8 assembled channels to nf channels, with the next GroupNorm's tile
// partials in the epilogue (models/ddpm.py:163-168 + :283 + the conv3x3 of :95; sampling/conditional.py:104-110).
//
// Replaces assemble_input_kernel (83 us at 160^2, B = 64) + the generic fp16 conv on a 16-channel padded copy (310 us): the layer is
// bound by its 629 MB of output, so the whole input side (6 of 8 channels real, 0.1 GB) is done inside the tile that needs it.
//
// Persistent: four 4-wave workgroups per CU, each walking 16 x 8 pixel tiles of its XCD's contiguous share (one 32-pixel M tile per wave:
// <= 128 VGPRs).  The packed weights of ALL cout groups (2^8-scaled hi + lo A/B fragments, 30 KB for 96 couts) and the bias stay in LDS for
// the workgroup's run.  Per tile the 18 x 10 patch is read from the NCHW sources (coalesced along W, straight-line loads), assembled, split
// into fp16 hi + lo ONCE per patch pixel and kept as two 16-byte planes in LDS; K = tap * 8 + channel (72, padded to 80 = 5 MFMA K steps;
// the padding tap reads a zero pixel).  Products: hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (NS = 2: the fp16x3 / fp16f8 modes - the
// raw inputs are not GroupNorm-ed, so the e4m3 correction form does not apply) or hi*hi only (NS = 1: the plain fp16 mode); the pixels are
// the M operand, so every output store covers whole 128-byte lines.  The loop's only wait on global memory (the next tile's patch, requested
// before the K loop) sits after the K loop: gfx9's in-order vmcnt makes a load wait also a wait for every older store.
#include <hip/hip_runtime.h>

#include "common.h"
#include "conv_ff.h"

namespace csd {

struct StemArgs {
  const float *x, *y, *yn;
  float ysig;
  const _Float16* w;
  const float* bias;
  float* out;
  double* stats;
  int B, S, Cx, Cy, Cout, tiles_x, tpi, n_groups, nblocks, centered;
  int abl;      // tuning aid (tune build, CSD_STEM_ABL): 1 no output stores, 2 no statistics, 4 no source loads, 8 no MFMAs
};

#define ST_PW 18
#define ST_PH 10
#define ST_TH 8                 // tile height (width 16)
#define ST_NPIX (ST_PW * ST_PH)
#define ST_ZERO ST_NPIX            // index of the all-zero pixel (the padding tap)
#define ST_PLANE 184               // pixels per LDS plane
#define ST_KSTEPS 5

template <int NT, int NS, bool YN>
__global__ __launch_bounds__(256, 4) void stem_kernel(const StemArgs k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half8* const phi = reinterpret_cast<half8*>(smem);
  half8* const plo = phi + ST_PLANE;
  float* const red = reinterpret_cast<float*>(smem + 2 * ST_PLANE * 16);      // [4 waves][NT * 32 couts][2]
  half8* const wl = reinterpret_cast<half8*>(smem + 2 * ST_PLANE * 16 + 4 * NT * 32 * 2 * 4);      // all cout groups' weight fragments
  float* const bl = reinterpret_cast<float*>(wl + k.n_groups * ST_KSTEPS * NT * NS * 64);           // bias [Cout]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = lane >> 5, p32 = lane & 31;
  const int S = k.S, Cx = k.Cx, Cy = k.Cy, Cout = k.Cout, n_groups = k.n_groups;
  const size_t HW = (size_t)S * S;
#ifdef CSD_TUNE
#define ST_ABL(bit) (k.abl & (bit))
#else
#define ST_ABL(bit) false
#endif

  // ---- persistent tile loop: workgroup p of gridDim.x (four per CU) runs the tiles j, j + P/8, ... of its XCD's contiguous share of the
  // launch (block p lands on XCD p % 8: neighbouring tiles of the same samples share their halo rows in one L2) ----
  const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, wstride = gridDim.x >> 3;
  const int xq = k.nblocks >> 3, xr = k.nblocks & 7;
  const int x_start = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, x_len = xq + (xcd < xr ? 1 : 0);
  if (wj >= x_len) return;

  // the packed weights stay in LDS for the workgroup's whole run (30 KB at 96 couts: re-reading them from L2 per tile and wave was
  // 1.5 GB per launch, more than twice the layer's output)
  {
    const int n16 = n_groups * ST_KSTEPS * NT * NS * 64;
    wg_copy_to_lds<256, 8>(reinterpret_cast<char*>(wl), reinterpret_cast<const char*>(k.w), n16 * 16, tid);
  }

  // the assembled values of a patch pixel (thread t < 184 owns patch pixel t of every tile).  Straight-line on purpose: every load is
  // unconditional (absent channels and out-of-image pixels read a valid dummy address and are replaced by a select) - with branches
  // around the loads hipcc put an s_waitcnt vmcnt(0) behind each of them, which on gfx9's in-order counter also waits for the previous
  // tile's 48 output stores (measured: 116 of the kernel's 250 us)
  const int py = tid / ST_PW, px = tid - (tid / ST_PW) * ST_PW;
  const float* cp[8];
  const float* cn[8];
  size_t cs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bool isx = c < Cx, isy = !isx && c < Cx + Cy;
    cp[c] = isx ? k.x + (size_t)c * HW : (isy ? k.y + (size_t)(c - Cx) * HW : k.x);
    cn[c] = (YN && isy) ? k.yn + (size_t)(c - Cx) * HW : k.x;
    cs[c] = isx ? (size_t)Cx * HW : (isy ? (size_t)Cy * HW : 0);
  }
  // load_pixel only REQUESTS the values (raw registers); finish_pixel - called where the wait belongs - assembles them
  struct Raw { float x[8], n[8]; bool inimg; };
  auto load_pixel = [&](int w, Raw& r) __attribute__((always_inline)) {
    const int b = w / k.tpi, tin = w - b * k.tpi;
    const int ty0 = (tin / k.tiles_x) * ST_TH, tx0 = (tin - (tin / k.tiles_x) * k.tiles_x) * 16;
    const int gy = ty0 + py - 1, gx = tx0 + px - 1;
    r.inimg = !ST_ABL(4) && tid < ST_NPIX && (unsigned)gy < (unsigned)S && (unsigned)gx < (unsigned)S;      // (zero padding applies to the
    const unsigned pix = r.inimg ? (unsigned)(gy * S + gx) : 0u;                                              // ASSEMBLED tensor: 0, not -1)
    // (uniform 64-bit base + 32-bit lane offset: the saddr form of global_load - one address register for all the loads)
#pragma unroll
    for (int c = 0; c < 8; ++c) r.x[c] = (cp[c] + (size_t)b * cs[c])[pix];
    if (YN) {
#pragma unroll
      for (int c = 0; c < 8; ++c) r.n[c] = (cn[c] + (size_t)b * cs[c])[pix];
    }
  };
  auto write_patch = [&](const Raw& r) __attribute__((always_inline)) {      // assemble, split once: plane hi [pixel][8 ch], plane lo [pixel][8 ch]
    if (tid < ST_PLANE) {
      half8 h, l;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float t = r.x[c];
        if (YN) t = (c >= Cx && c < Cx + Cy) ? t + r.n[c] * k.ysig : t;
        t = k.centered ? t : 2.f * t - 1.f;
        const float v = (r.inimg && c < Cx + Cy) ? t : 0.f;
        h[c] = (_Float16)v;
        l[c] = (_Float16)(v - (float)h[c]);
      }
      phi[tid] = h;
      if (NS == 2) plo[tid] = l;
    }
  };
  Raw vn;
  load_pixel(x_start + wj, vn);
  for (int i = tid; i < Cout; i += 256) bl[i] = k.bias[i];
  write_patch(vn);
  const int pix0 = (wave * 2 + (p32 >> 4)) * ST_PW + (p32 & 15);
  const float wunscale = 1.0f / C16_WSCALE;
  ff_barrier();

  // Per tile: request the NEXT tile's patch -> K loop -> barrier -> convert + write the next patch -> output stores -> statistics.
  // gfx9 counts loads and stores on ONE in-order counter, so a wait for a load also waits for every store issued before it: the only
  // vmcnt wait of the loop (the patch conversion) sits where the youngest older stores are a whole K loop old, and nothing else in the
  // loop reads global memory (bias and weights live in LDS).  ff_barrier: LDS hand-over without a release fence (__syncthreads() drains
  // vmcnt(0) - the tile's output stores - at every barrier).
  for (int local = wj; local < x_len; local += wstride) {
    const int tile = x_start + local;
    const int b = tile / k.tpi, tin = tile - b * k.tpi;
    const int ty0 = (tin / k.tiles_x) * ST_TH, tx0 = (tin - (tin / k.tiles_x) * k.tiles_x) * 16;
    const bool more = local + wstride < x_len;
    if (more) load_pixel(tile + wstride, vn);

    for (int ng = 0; ng < n_groups; ++ng) {
      // ---- K loop: 5 steps of 16 = 2 taps x 8 channels; lanes of K half kh read tap 2s + kh ----
      floatx16 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
      const half8* const wq = wl + (size_t)ng * ST_KSTEPS * NT * NS * 64 + lane;
#pragma unroll
      for (int s = 0; s < ST_KSTEPS; ++s) {
        constexpr int kNoTap = -1;
        const int t0 = 2 * s, t1 = 2 * s + 1;
        const int off0 = (t0 / 3) * ST_PW + t0 % 3;
        const int off1 = t1 < 9 ? (t1 / 3) * ST_PW + t1 % 3 : kNoTap;
        const int idx = kh ? (off1 == kNoTap ? ST_ZERO : pix0 + off1) : pix0 + off0;
        const half8 bh = phi[idx];
        half8 bl16;
        if (NS == 2) bl16 = plo[idx];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (ST_ABL(8)) continue;
          const half8 wh = wq[((s * NT + nt) * NS) * 64];
          if (NS == 2) {      // small products first
            const half8 wlo = wq[((s * NT + nt) * NS + NS - 1) * 64];
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, wlo, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl16, wh, acc[nt], 0, 0, 0);
          }
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, wh, acc[nt], 0, 0, 0);
        }
      }
      if (ng == n_groups - 1) {
        ff_barrier();                    // every wave has read this tile's patch (and the previous partials in red)
        if (more) write_patch(vn);
      } else {
        ff_barrier();                    // (several cout groups: the previous group's partials in red have been read)
      }

      // ---- epilogue: un-scale, bias, NHWC stores.  The pixels are the MFMA's M operand, so a lane holds ONE cout (ng * 32 NT + nt * 32 +
      //      lane % 32) of 16 pixels (register r = pixel 8 (r / 4) + 4 kh + r % 4 of the wave's 32): every store instruction writes two
      //      whole 128-byte lines (tools/probes/store_probe.hip: 5.4 TB/s against 3.3 TB/s for 16-byte pieces of four different instructions) ----
      const int c_lane = ng * NT * 32 + p32;
      float* const obase = k.out + (((size_t)b * S + ty0 + wave * 2) * S + tx0) * Cout + ng * NT * 32;      // uniform
      const unsigned lane_off = (unsigned)(4 * kh * Cout + p32);                                            // + one lane offset: saddr stores
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = bl[c_lane + nt * 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* const ub = obase + ((size_t)(r >> 3) * S + 8 * ((r >> 2) & 1) + (r & 3)) * Cout + nt * 32;
          acc[nt][r] = acc[nt][r] * wunscale + bv;
          if (!ST_ABL(1)) ub[lane_off] = acc[nt][r];
        }
      }

      // ---- GroupNorm partials of the written tile: (sum, sum of squares) per cout over its 128 pixels: in-lane over the 16 registers,
      //      one exchange between the K halves, the four waves through LDS ----
      if (k.stats && !ST_ABL(2)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float vs = 0.f, vq = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            vs += acc[nt][r];
            vq = fmaf(acc[nt][r], acc[nt][r], vq);
          }
          vs += __shfl_xor(vs, 32);
          vq += __shfl_xor(vq, 32);
          if (kh == 0) {
            red[(wave * NT * 32 + nt * 32 + p32) * 2 + 0] = vs;
            red[(wave * NT * 32 + nt * 32 + p32) * 2 + 1] = vq;
          }
        }
      }
      ff_barrier();                      // the partials and the next patch are in LDS
      if (k.stats && !ST_ABL(2) && tid < NT * 32) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          s += (double)red[(wv * NT * 32 + tid) * 2 + 0];
          q += (double)red[(wv * NT * 32 + tid) * 2 + 1];
        }
        double* dst = k.stats + ((size_t)tile * Cout + ng * NT * 32 + tid) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
static inline int stem_nt(int cout) { return cout % 96 == 0 ? 3 : 2; }

bool stem_supported(int Cx, int Cy, int Cout, int S, int ns) {
  if (CSD_TUNE_ENV("CSD_NO_STEM")) return false;
  return ns >= 1 && ns <= 3 && Cx > 0 && Cy >= 0 && Cx + Cy <= 8 && S % 16 == 0 && S >= 16 && (Cout % 96 == 0 || Cout % 64 == 0);
}

size_t stem_packed_bytes(int Cout, int ns) {
  const int nt = stem_nt(Cout), planes = ns >= 2 ? 2 : 1;
  return (size_t)(Cout / (32 * nt)) * ST_KSTEPS * nt * planes * 1024;
}

// weights in A-fragment order: [cout group][K step][cout tile][plane][lane = (k half << 5) | cout row][8 halves]; k = tap * 8 + channel,
// scaled by 2^8 (exact) so the lo plane stays out of the fp16 subnormals
__global__ void stem_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int Cin, int Cout, int planes, int nt,
                                 size_t n_halves) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_halves) return;
  const int e = (int)(idx & 7);
  const int ln = (int)((idx >> 3) & 63);
  size_t rest = idx >> 9;
  const int pl = (int)(rest % planes); rest /= planes;
  const int t = (int)(rest % nt); rest /= nt;
  const int s = (int)(rest % ST_KSTEPS);
  const int ng = (int)(rest / ST_KSTEPS);
  const int tap = 2 * s + (ln >> 5);
  const int cout = (ng * nt + t) * 32 + (ln & 31);
  float v = 0.f;
  if (tap < 9 && e < Cin && cout < Cout) v = w[((size_t)cout * Cin + e) * 9 + tap] * C16_WSCALE;
  const _Float16 hi = (_Float16)v;
  wpack[idx] = pl == 0 ? hi : (_Float16)(v - (float)hi);
}

int stem_pack_weight(const float* w, int Cin, int Cout, int ns, void* wpack, hipStream_t s) {
  const size_t n = stem_packed_bytes(Cout, ns) / 2;
  hipLaunchKernelGGL(stem_pack_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, w, (_Float16*)wpack, Cin, Cout,
                     ns >= 2 ? 2 : 1, stem_nt(Cout), n);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int stem_tiles_per_image(int S) { return (S / 16) * (S / ST_TH); }

int stem_launch(const float* x, const float* y, const float* y_noise, float y_sigma, const void* wpack, const float* bias, float* out,
                double* stats, int B, int Cx, int Cy, int Cout, int S, int centered, int ns, hipStream_t s) {
  CSD_REQUIRE(stem_supported(Cx, Cy, Cout, S, ns), "stem: unsupported shape (%d+%d -> %d channels, %dx%d)", Cx, Cy, Cout, S, S);
  StemArgs k;
  k.x = x; k.y = y; k.yn = y_noise; k.ysig = y_sigma;
  k.w = reinterpret_cast<const _Float16*>(wpack); k.bias = bias; k.out = out; k.stats = stats;
  k.B = B; k.S = S; k.Cx = Cx; k.Cy = y ? Cy : 0; k.Cout = Cout;
  k.tiles_x = S / 16; k.tpi = stem_tiles_per_image(S);
  const int nt = stem_nt(Cout);
  k.n_groups = Cout / (32 * nt);
  k.nblocks = B * k.tpi;               // tiles (every workgroup runs all cout groups of its tiles)
  k.centered = centered;
  k.abl = CSD_TUNE_ENV("CSD_STEM_ABL") ? atoi(CSD_TUNE_ENV("CSD_STEM_ABL")) : 0;
  const int planes = ns >= 2 ? 2 : 1;
  const size_t lds = 2 * ST_PLANE * 16 + 4 * (size_t)nt * 32 * 2 * 4 + (size_t)k.n_groups * ST_KSTEPS * nt * planes * 1024 + (size_t)Cout * 4;
  CSD_REQUIRE(lds <= 160 * 1024, "stem: %d couts need %zu bytes of LDS", Cout, lds);
  const int n_cu = device_cu_count8();
  int per_cu = (int)((160 * 1024) / lds);      // persistent: as many workgroups per CU as LDS (and 128 VGPRs) allow, a multiple of the 8 XCDs
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  const int want = n_cu * per_cu;
  const int grid = k.nblocks < want ? (k.nblocks + 7) / 8 * 8 : want;
  auto go = [&](auto kern) -> int {
    CSD_SET_MAX_LDS_ONCE(kern);           // (one flag array per kernel instantiation: the lambda's operator() is a template)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, k);
    return CSD_OK;
  };
  int rc;
  const bool yn = y_noise != nullptr && k.Cy > 0;
  if (nt == 3 && planes == 2) rc = yn ? go(stem_kernel<3, 2, true>) : go(stem_kernel<3, 2, false>);
  else if (nt == 3) rc = yn ? go(stem_kernel<3, 1, true>) : go(stem_kernel<3, 1, false>);
  else if (planes == 2) rc = yn ? go(stem_kernel<2, 2, true>) : go(stem_kernel<2, 2, false>);
  else rc = yn ? go(stem_kernel<2, 1, true>) : go(stem_kernel<2, 1, false>);
  if (rc) return rc;
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

}  // namespace csd
