// conv_pw16.hip - pointwise (1x1) convolution on the fp16 MFMA path: the NIN shortcut of a ResnetBlock
// (reference models/layerspp.py:ResnetBlockDDPMpp Conv_2 / layers.py:NIN) and the q/k/v/out projections
// of AttnBlockpp, in the F16 / F16X3 precision modes.
//
// A 1x1 convolution over NHWC fp32 activations is a streaming problem: 192 -> 96 channels at 160x160 reads
// 1.26 GB and writes 0.63 GB per launch (B = 64) for 60 GFLOP - 385 us of fp32 MFMA but 345 us of HBM, so
// on the fp32 kernel it is bound by BOTH.  Here the products run on v_mfma_f32_16x16x32_f16 (16x faster), no
// LDS staging is needed for the activations at all, and the kernel is a pure HBM stream:
//
//   D^T[cout][pixel] = W^T[cout][cin] * X^T[cin][pixel]
//   A operand = weights  : lane l holds cout (l & 15), cin (l >> 4)*8 .. +8 of the 32-channel K step;
//                          pre-packed in fragment order, copied once per workgroup into LDS
//   B operand = pixels   : lane l holds pixel (l & 15), the same 8 channels - 32 contiguous bytes of the
//                          fp32 row; the four lanes of a pixel cover one full 128-byte line
//   D                    : lane l holds couts (l >> 4)*4 .. +4 of pixel (l & 15): one 16-byte store
//
// A wave owns MTP x 16 pixels x 96 couts; a workgroup (4 waves) loops over pixel tiles (persistent), with
// the raw fp32 rows of the next TWO K steps always in flight (register ring of 2, re-issued right after the
// fp32 -> fp16 conversion), across tile boundaries.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace csd {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

#define PW_THREADS 256
#define PW_NTL 6                 // 16-cout tiles per workgroup pass (96 couts)
#define PW_WSCALE 256.0f         // weights are packed * 2^8 (keeps the lo half of F16X3 out of the subnormals)
#ifndef PW_DEPTH
#define PW_DEPTH 2          // K steps of raw rows in flight per wave on the big maps (4: no faster, measured)
#endif

struct PwKArgs {
  ConvArgs a;
  int C0, C1, Cout;
  int npix;        // B*OH*OW
  int hw;          // pixels per sample (row of the GroupNorm scale/shift table)
  int ks;          // K steps of 32 channels
  int ntiles;      // pixel tiles
};

__device__ __forceinline__ float4 pw_gload4(const float* p) {
  const v4f_t v = *(const __attribute__((address_space(1))) v4f_t*)(p);
  return make_float4(v.x, v.y, v.z, v.w);
}

template <int NS, int MTP, int OCC>
__global__ __launch_bounds__(PW_THREADS, OCC) void pw16_kernel(const float* __restrict__ g_src0,
                                                          const float* __restrict__ g_src1,
                                                          const char* __restrict__ g_wpack, const PwKArgs k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WPIX = MTP * 16;          // pixels per wave
  constexpr int TPIX = 4 * WPIX;          // pixels per workgroup tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l16 = lane & 15;
  const int kq = lane >> 4;
  const int ng = blockIdx.y;
  const int Cin = k.C0 + k.C1;

  const int wbytes = k.ks * PW_NTL * NS * 1024;
  float* const lds_bias = reinterpret_cast<float*>(smem + wbytes);      // bias of this group's 96 couts behind the weights
  const int col_base = ng * (PW_NTL * 16) + kq * 4;       // this lane's first cout of tile 0

  const int nt_mine = (k.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = nt_mine * k.ks;
  const bool has_norm = k.a.nscale != nullptr;
  const bool has_res = k.a.res != nullptr;
  const bool has_act = k.a.act == CSD_ACT_SWISH;          // (only next to a GroupNorm: the tap-partial form of a tiny-Cout 3x3, below)
  const int ntl = min(PW_NTL, (k.Cout - (int)blockIdx.y * (PW_NTL * 16) + 15) / 16);     // 16-cout tiles of this group that exist

  // load-side pipeline depth in K steps: small maps (one or two tiles per workgroup, MTP = 1) have nothing but their own K loop to
  // cover a global load with - four steps in flight, GroupNorm scale / shift rows included; big ones keep two (registers)
  constexpr int D = MTP == 1 ? 4 : PW_DEPTH;
  float4 raw[D][MTP][2];
  constexpr bool PFN = MTP == 1;          // the scale / shift rows ride with the prefetch (else: loaded at their use, as ever)
  float4 nsc[PFN ? D : 1][MTP][2], nsh[PFN ? D : 1][MTP][2];
  floatx4 acc[MTP][PW_NTL];
#pragma unroll
  for (int j = 0; j < MTP; ++j)
#pragma unroll
    for (int nt = 0; nt < PW_NTL; ++nt) acc[j][nt] = floatx4{0.f, 0.f, 0.f, 0.f};

  // (tile, K step) counters of the load side and of the MFMA side of the pipeline
  int l_tile = blockIdx.x, l_kk = 0, l_it = 0;
  int c_tile = blockIdx.x, c_kk = 0;

  auto issue = [&](float4 (&dst)[MTP][2], float4 (&dsc)[MTP][2], float4 (&dsh)[MTP][2]) {
    if (l_it < total) {
      const int kb = l_kk * 32;
      const bool s1 = kb >= k.C0;
      const float* src = s1 ? g_src1 : g_src0;
      const int C = s1 ? k.C1 : k.C0;
      const int ch = (s1 ? kb - k.C0 : kb) + kq * 8;
#pragma unroll
      for (int j = 0; j < MTP; ++j) {
        int p = l_tile * TPIX + wave * WPIX + j * 16 + l16;
        p = p < k.npix ? p : k.npix - 1;                   // tail pixels: any valid row, never stored
        const float* q = src + (size_t)p * C + ch;
        dst[j][0] = pw_gload4(q);
        dst[j][1] = pw_gload4(q + 4);
        if (PFN && has_norm) {
          const size_t row = (size_t)(p / k.hw) * Cin + l_kk * 32 + kq * 8;
          dsc[j][0] = pw_gload4(k.a.nscale + row); dsc[j][1] = pw_gload4(k.a.nscale + row + 4);
          dsh[j][0] = pw_gload4(k.a.nshift + row); dsh[j][1] = pw_gload4(k.a.nshift + row + 4);
        }
      }
      ++l_it;
      if (++l_kk == k.ks) { l_kk = 0; l_tile += gridDim.x; }
    }
  };

#pragma unroll
  for (int u = 0; u < D; ++u) issue(raw[u], nsc[PFN ? u : 0], nsh[PFN ? u : 0]);

  // ---- this cout group's weights -> LDS, once per workgroup (behind the first activation requests: their latency overlaps the copy) ----
  wg_copy_to_lds<PW_THREADS, 9>(smem, g_wpack + (size_t)ng * wbytes, wbytes, tid);      // (9 x 4 KiB per round: a 288-channel layer's 108 KiB in three)
  if (tid < PW_NTL * 16) {                                // (zero where the cout does not exist)
    const int c = ng * (PW_NTL * 16) + tid;
    lds_bias[tid] = (k.a.bias && c < k.Cout) ? k.a.bias[c] : 0.f;
  }
  __syncthreads();

  for (int it = 0; it < total; it += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (it + u < total) {
        // ---- fp32 rows -> fp16 B fragments (GroupNorm affine applied here when the layer has one) ----
        half8_t bh[MTP], bl[MTP];
#pragma unroll
        for (int j = 0; j < MTP; ++j) {
          float v[8] = {raw[u][j][0].x, raw[u][j][0].y, raw[u][j][0].z, raw[u][j][0].w,
                        raw[u][j][1].x, raw[u][j][1].y, raw[u][j][1].z, raw[u][j][1].w};
          if (has_norm) {
            float4 s0, s1, h0, h1;
            if constexpr (PFN) {
              s0 = nsc[u][j][0]; s1 = nsc[u][j][1]; h0 = nsh[u][j][0]; h1 = nsh[u][j][1];
            } else {
              int p = c_tile * TPIX + wave * WPIX + j * 16 + l16;
              p = p < k.npix ? p : k.npix - 1;
              const size_t row = (size_t)(p / k.hw) * Cin + c_kk * 32 + kq * 8;
              s0 = pw_gload4(k.a.nscale + row); s1 = pw_gload4(k.a.nscale + row + 4);
              h0 = pw_gload4(k.a.nshift + row); h1 = pw_gload4(k.a.nshift + row + 4);
            }
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaf(v[q], sc[q], sh[q]);     // same fma as the fp32 staging
            if (has_act) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q] = v[q] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[q]));     // SiLU as conv_ff / gn_apply16
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const _Float16 hi = (_Float16)v[q];
            bh[j][q] = hi;
            if (NS == 2) bl[j][q] = (_Float16)(v[q] - (float)hi);
          }
        }
        issue(raw[u], nsc[PFN ? u : 0], nsh[PFN ? u : 0]);       // this slot is free again: request K step it + u + D
        // ---- MFMAs: 6 cout tiles x MTP pixel tiles ----
        const char* wk = smem + (size_t)c_kk * PW_NTL * NS * 1024 + lane * 16;
#pragma unroll
        for (int nt = 0; nt < PW_NTL; ++nt) {
          if (nt >= ntl) continue;             // (uniform: a 28-cout layer runs 2 of the 6 tiles)
          const half8_t wh = *reinterpret_cast<const half8_t*>(wk + (nt * NS) * 1024);
          half8_t wl;
          if (NS == 2) wl = *reinterpret_cast<const half8_t*>(wk + (nt * NS + 1) * 1024);
#pragma unroll
          for (int j = 0; j < MTP; ++j) {
            if (NS == 2) {
              acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[j], acc[j][nt], 0, 0, 0);
              acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[j], acc[j][nt], 0, 0, 0);
            }
            acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[j], acc[j][nt], 0, 0, 0);
          }
        }
        // ---- end of a tile: epilogue (bias + residual, 16-byte buffer stores; tail lanes out of range) ----
        if (++c_kk == k.ks) {
          constexpr unsigned OOB = 0x80000000u;
          constexpr int RSRC_FLAGS = 0x00020000;
          const size_t p_base = (size_t)c_tile * TPIX + wave * WPIX;
          const int left = k.npix - (int)p_base;            // valid pixels from p_base on (may be <= 0)
          const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(
              k.a.out + p_base * k.a.out_stride + k.a.out_coff, 0, OOB, RSRC_FLAGS);
          const __amdgpu_buffer_rsrc_t res_r = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<float*>(has_res ? k.a.res + p_base * k.Cout : k.a.out), 0, OOB, RSRC_FLAGS);
          const float unscale = 1.0f / PW_WSCALE;
#pragma unroll
          for (int j = 0; j < MTP; ++j) {
            const int pl = j * 16 + l16;
            const bool pv = pl < left;
            uint4_t rv[PW_NTL];
#pragma unroll
            for (int nt = 0; nt < PW_NTL; ++nt) {
              const int col0 = col_base + nt * 16;
              const unsigned off = (pv && has_res && col0 < k.Cout) ? (unsigned)(pl * k.Cout + col0) * 4u : OOB;
              rv[nt] = __builtin_amdgcn_raw_buffer_load_b128(res_r, off, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < PW_NTL; ++nt) {
              const int col0 = col_base + nt * 16;
              const float4 b = *reinterpret_cast<const float4*>(lds_bias + nt * 16 + kq * 4);
              uint4_t ov;
              // (acc*2^-8 + bias) + residual, then the skip_rescale factor: association of the reference
              ov.x = __float_as_uint(((acc[j][nt][0] * unscale + b.x) + __uint_as_float(rv[nt].x)) * k.a.out_scale);
              ov.y = __float_as_uint(((acc[j][nt][1] * unscale + b.y) + __uint_as_float(rv[nt].y)) * k.a.out_scale);
              ov.z = __float_as_uint(((acc[j][nt][2] * unscale + b.z) + __uint_as_float(rv[nt].z)) * k.a.out_scale);
              ov.w = __float_as_uint(((acc[j][nt][3] * unscale + b.w) + __uint_as_float(rv[nt].w)) * k.a.out_scale);
              const unsigned off = (pv && col0 < k.Cout) ? (unsigned)(pl * k.a.out_stride + col0) * 4u : OOB;
              __builtin_amdgcn_raw_buffer_store_b128(ov, out_r, off, 0, 0);
              acc[j][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
          }
          c_kk = 0;
          c_tile += gridDim.x;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static size_t pw16_w_bytes(const ConvPlan& p, int ns) { return (size_t)((p.C0 + p.C1) / 32) * PW_NTL * ns * 1024; }
static size_t pw16_lds_bytes(const ConvPlan& p, int ns) { return pw16_w_bytes(p, ns) + PW_NTL * 16 * sizeof(float); }

bool pw16_supported(const ConvPlan& p, int ns) {
  return p.taps == 1 && p.stride == 1 && p.up == 0 && p.C0 > 0 && p.C0 % 32 == 0 && p.C1 % 32 == 0 && p.Cout % 4 == 0 &&
         (ns == 1 || ns == 2) && pw16_lds_bytes(p, ns) <= 150 * 1024;
}

size_t pw16_packed_bytes(const ConvPlan& p, int ns) {
  const int n_groups = cdiv(p.Cout, PW_NTL * 16);
  return (size_t)n_groups * pw16_w_bytes(p, ns);
}

__global__ void pw16_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int layout, int cin_src,
                                 int cout_src, int cout_off, int ks, int ns, int cout_pad) {
  // one thread per (padded cout, cin) inside [cout_off, cout_off + cout_src) rounded out to whole 16-tiles
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cin_tot = ks * 32;
  const int t_lo = cout_off / 16, t_hi = (cout_off + cout_src + 15) / 16;
  if (idx >= (size_t)(t_hi - t_lo) * 16 * cin_tot) return;
  const int cp = t_lo * 16 + (int)(idx / cin_tot);         // padded cout
  const int cin = (int)(idx % cin_tot);
  const int cout = cp - cout_off;
  if (cp >= cout_pad || cout < 0 || cout >= cout_src || cin >= cin_src) return;   // padding stays zero
  const float v = ((layout == 0) ? w[(size_t)cout * cin_src + cin] : w[(size_t)cin * cout_src + cout]) * PW_WSCALE;   // (layout 2 == 1 for 1x1)
  const int ng = cp / (PW_NTL * 16), nt = (cp % (PW_NTL * 16)) / 16, r = cp % 16;
  const int kk = cin / 32, kq = (cin % 32) / 8, q = cin % 8;
  const int lane = kq * 16 + r;
  _Float16* dst = wpack + ((((size_t)ng * ks + kk) * PW_NTL + nt) * ns) * 512 + lane * 8 + q;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  if (ns == 2) dst[512] = (_Float16)(v - (float)hi);
}

__global__ void pw16_zero_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

int pw16_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                     void* wpack, hipStream_t s) {
  CSD_REQUIRE(cout_off % 16 == 0, "pw16 pack: cout offset %d is not a multiple of 16", cout_off);
  const int ks = (p.C0 + p.C1) / 32;
  const int cout_pad = cdiv(p.Cout, PW_NTL * 16) * PW_NTL * 16;
  if (cout_off == 0) {
    const size_t n32 = pw16_packed_bytes(p, ns) / 4;
    hipLaunchKernelGGL(pw16_zero_kernel, dim3((unsigned)cdiv64(n32, 256)), dim3(256), 0, s, (uint32_t*)wpack, n32);
    CSD_LAUNCH_CHECK();
  }
  const size_t total = (size_t)(cdiv(cout_off + cout_src, 16) - cout_off / 16) * 16 * ks * 32;
  hipLaunchKernelGGL(pw16_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, (_Float16*)wpack, layout,
                     cin_src, cout_src, cout_off, ks, ns, cout_pad);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---------------------------------------------------------------------------------------------
// tap-partial form of a 3x3 convolution with a handful of output channels (the networks' last layer: nf -> 3 or 6 channels,
// reference models/ddpm.py:146, models/ncsnpp.py:234).  On the 3x3 kernels its 3 couts occupy one
// 32-cout MFMA tile (90 % zeros) behind a gn_apply16 pass: 0.97 ms per evaluation at 160^2, B = 64, for 8.5 GFLOP.  Here
//     P[b][y][x][tap*Co + co] = sum_ci act(GN(x))[b][y][x][ci] * W[co][ci][tap]       one POINTWISE contraction, nf -> 9*Co (27) couts,
//                                                                                      GroupNorm + SiLU applied in its loader
//     out[b][co][y][x] = bias[co] + sum_tap P[b][y+dy][x+dx][tap*Co + co]               9 shifted reads per output (zero padding =
//                                                                                      skipped taps: the padding is of the ACTIVATED tensor)
// so the activations are read once, as a stream.
// ---------------------------------------------------------------------------------------------
int pw16_taps_cout(int cout) { return (9 * cout + 3) / 4 * 4; }

__global__ void pw16_pack_taps_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int cin, int cout, int ks, int ns) {
  // w: [cout][cin][3][3]; one thread per (virtual cout = tap * cout + co, cin)
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)9 * cout * cin) return;
  const int cp = (int)(idx / cin), ci = (int)(idx % cin);
  const int tap = cp / cout, co = cp % cout;
  const float v = w[((size_t)co * cin + ci) * 9 + tap] * PW_WSCALE;
  const int ng = cp / (PW_NTL * 16), nt = (cp % (PW_NTL * 16)) / 16, r = cp % 16;
  const int kk = ci / 32, kq = (ci % 32) / 8, q = ci % 8;
  _Float16* dst = wpack + ((((size_t)ng * ks + kk) * PW_NTL + nt) * ns) * 512 + (kq * 16 + r) * 8 + q;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  if (ns == 2) dst[512] = (_Float16)(v - (float)hi);
}

int pw16_pack_weight_taps(const ConvPlan& p, int ns, const float* w, int cout, void* wpack, hipStream_t s) {
  CSD_REQUIRE(p.C1 == 0 && p.Cout == pw16_taps_cout(cout) && p.Cout <= PW_NTL * 16, "pw16 tap pack: Cout %d for %d real couts", p.Cout, cout);
  const int ks = p.C0 / 32;
  const size_t n32 = pw16_packed_bytes(p, ns) / 4;
  hipLaunchKernelGGL(pw16_zero_kernel, dim3((unsigned)cdiv64(n32, 256)), dim3(256), 0, s, (uint32_t*)wpack, n32);
  CSD_LAUNCH_CHECK();
  const size_t total = (size_t)9 * cout * p.C0;
  hipLaunchKernelGGL(pw16_pack_taps_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, (_Float16*)wpack, p.C0, cout, ks, ns);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// out = (bias + sum over the 9 taps, in tap order) * out_scale.  A workgroup = one 16 x 16 output tile: the 18 x 18 partial records
// (cp floats each, contiguous runs of 18 records per image row) go through LDS once - 9 float4 per thread, all in flight - and each
// thread adds its pixel's 9 x CO values from there (zero records outside the image = the zero padding of the activated tensor).
// (One thread per pixel gathering straight from global memory: 27 dword loads with a 112-byte lane stride, 183 us at 160^2, B = 64.)
template <int CO>
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                     const float* __restrict__ res, float* __restrict__ out, int H, int W, int cp,
                                                     int nchw, float out_scale) {
  extern __shared__ __attribute__((aligned(16))) float ts_tile[];      // [18 * 18][cp]
  const int tid = threadIdx.x;
  const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16;
  const size_t b = blockIdx.z;
  const int cp4 = cp >> 2, total = 324 * cp4;
  constexpr int NV = (324 * (((9 * CO + 3) / 4 * 4) / 4) + 255) / 256;      // float4 per thread: 9 for 3 couts, 18 for 6
  float4 v[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int i = min(tid + u * 256, total - 1);
    const int rec = i / cp4, q = i - rec * cp4;
    const int pr = rec / 18, pc = rec - pr * 18;
    const int y = ty0 - 1 + pr, x = tx0 - 1 + pc;
    v[u] = (y >= 0 && y < H && x >= 0 && x < W) ? *reinterpret_cast<const float4*>(part + ((b * H + y) * W + x) * cp + q * 4)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int i = tid + u * 256;
    if (i < total) *reinterpret_cast<float4*>(ts_tile + (size_t)i * 4) = v[u];      // (record-major, cp4 float4 per record: linear)
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  const int y = ty0 + ty, x = tx0 + tx;
  if (y >= H || x >= W) return;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const float* q = ts_tile + ((ty + tap / 3) * 18 + tx + tap % 3) * cp + tap * CO;
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] += q[c];
  }
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    // ((conv + bias) + residual) * out_scale: the association of the convolution kernels' epilogues (residual: NHWC, CO channels -
    // the coarser levels' upsampled sum of an NCSN++ output pyramid)
    const float r = ((acc[c] + bias[c]) + (res ? res[((b * H + y) * W + x) * CO + c] : 0.f)) * out_scale;
    if (nchw) out[((b * CO + c) * H + y) * W + x] = r;
    else out[((b * H + y) * W + x) * CO + c] = r;
  }
}

int tapsum_launch(const float* part, const float* bias, const float* res, float* out, int B, int H, int W, int cout, int nchw,
                  float out_scale, hipStream_t s) {
  const int cp = pw16_taps_cout(cout);
  const dim3 grid(cdiv(W, 16), cdiv(H, 16), B), block(256);
  const size_t lds = (size_t)324 * cp * sizeof(float);
  switch (cout) {
#define CSD_TS_CASE(CO)                                                                                                              \
  case CO: {                                                                                                                         \
    static bool attr_set = false;                                                                                                    \
    if (!attr_set && lds > 64 * 1024) {      /* 6 couts (the paired networks' score_x | score_y): 72.6 KB */                         \
      CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tapsum_kernel<CO>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        160 * 1024));                                                                               \
      attr_set = true;                                                                                                               \
    }                                                                                                                                \
    hipLaunchKernelGGL(tapsum_kernel<CO>, grid, block, lds, s, part, bias, res, out, H, W, cp, nchw, out_scale);                     \
    break;                                                                                                                           \
  }
    CSD_TS_CASE(1) CSD_TS_CASE(2) CSD_TS_CASE(3) CSD_TS_CASE(4) CSD_TS_CASE(5) CSD_TS_CASE(6)
#undef CSD_TS_CASE
    default:
      set_error("tapsum: %d output channels not instantiated", cout);
      return CSD_ERR_INVALID;
  }
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

bool pw16_taps_supported(int cin, int cout, int ns) {
  ConvPlan p;
  memset(&p, 0, sizeof(p));
  p.taps = 1; p.stride = 1; p.C0 = cin; p.Cout = pw16_taps_cout(cout);
  return cout >= 1 && cout <= 6 && pw16_supported(p, ns) && !CSD_TUNE_ENV("CSD_NO_TAPSUM");
}

template <int NS, int MTP, int OCC>
static int pw16_launch_t(const PwKArgs& k, const ConvPlan& p, hipStream_t s) {
  auto kern = pw16_kernel<NS, MTP, OCC>;
  const size_t lds = pw16_lds_bytes(p, NS);
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess) {
      set_error("pw16: cannot raise the dynamic LDS limit");
      return CSD_ERR_HIP;
    }
    attr_lds = 160 * 1024;
  }
  const int n_groups = cdiv(p.Cout, PW_NTL * 16);
  // persistent: two workgroups per CU (register bound) share the tiles of one cout group
  int gx = (OCC * 256) / n_groups;
  if (gx < 1) gx = 1;
  if (gx > k.ntiles) gx = k.ntiles;
  hipLaunchKernelGGL(kern, dim3(gx, n_groups), dim3(PW_THREADS), lds, s, k.a.src0, k.a.src1,
                     reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int pw16_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s) {
  CSD_REQUIRE(pw16_supported(p, ns), "pw16: unsupported layer (taps %d C0 %d C1 %d Cout %d)", p.taps, p.C0, p.C1, p.Cout);
  CSD_REQUIRE(!a.temb && !a.out_nchw && (a.act == CSD_ACT_NONE || (a.act == CSD_ACT_SWISH && a.nscale)),
              "pw16: temb / NCHW output / an activation without a GroupNorm are not supported");
  PwKArgs k;
  k.a = a;
  k.C0 = p.C0; k.C1 = p.C1; k.Cout = p.Cout;
  k.npix = p.B * p.OH * p.OW;
  k.hw = p.OH * p.OW;
  k.ks = (p.C0 + p.C1) / 32;
  // big layers: 128-pixel tiles (2 x 16 pixels per wave) - 150 registers, three workgroups per CU measured
  // fastest (more waves in flight beat more bytes per wave); small layers: 64-pixel tiles to fill the chip
  const bool big = cdiv(k.npix, 128) >= 768;
  k.ntiles = cdiv(k.npix, big ? 128 : 64);
  if (ns == 2) return big ? pw16_launch_t<2, 2, 2>(k, p, s) : pw16_launch_t<2, 1, 2>(k, p, s);
  return big ? pw16_launch_t<1, 2, 3>(k, p, s) : pw16_launch_t<1, 1, 2>(k, p, s);
}

}  // namespace csd
