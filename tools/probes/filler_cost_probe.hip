// Round-5 probe: what does ONE filler instruction of a given kind cost beside a dense MFMA stream (one wave per SIMD, 12 accumulators
// round robin, operands constant)?  N fillers of kind K after every v_mfma_f32_32x32x16_f16; reported: cycles per MFMA (floor 32).
//   kinds: 0 v_fma_f32 | 1 v_exp_f32 | 2 v_rcp_f32 | 3 v_cvt_pk_f16_f32 | 4 v_fma_mix_f32 | 5 v_cndmask_b32 (sgpr mask) | 6 ds_bpermute_b32 |
//          7 ds_read_b128 | 8 ds_write_b64 | 9 ds_write_b128 | 10 v_add_f32 dependent chain | 11 buffer_load_dwordx4 (L2 hit) | 12 s_nop 0
// hipcc --offload-arch=gfx950 -O3 tools/probes/filler_cost_probe.hip -o build/filler_cost_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float* out, const u4* gsrc, int iters, long long* clk, const u4* big) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = i * 1e-6f;
  __syncthreads();
  floatx16 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8 a[4], b[3];
  for (int q = 0; q < 8; ++q) {
    for (int i = 0; i < 4; ++i) a[i][q] = (_Float16)(lane * 0.01f + q + i);
    for (int i = 0; i < 3; ++i) b[i][q] = (_Float16)(0.5f * q - i);
  }
  float v[8] = {1.f, 2.f, 3.f, 4.f, 0.5f, 0.25f, 0.125f, 0.3f};
  int pk[4] = {0, 0, 0, 0};
  u4 rd[4], ld[4];
  for (int q = 0; q < 4; ++q) { rd[q] = u4{0, 0, 0, 0}; ld[q] = u4{0, 0, 0, 0}; }
  const unsigned lp = (unsigned)(size_t)(lds + lane * 16), bp = ((lane + 4) & 63) * 4;
  const unsigned long long msk = 0x5555aaaa3333ccccull;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      const int i = g % 12;
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i / 3]), "v"(b[i % 3]));
#pragma unroll
      for (int f = 0; f < N; ++f) {
        const int q = (g * N + f) & 3;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q]) : "v"(v[q + 4]), "v"(v[(q + 1) & 3]));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(v[q]) : "v"(v[q + 4]));
        else if (KIND == 2) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[q]) : "v"(v[q + 4]));
        else if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q]) : "v"(v[q]), "v"(v[q + 4]));
        else if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v[q]) : "v"(pk[q]), "v"(v[q + 4]));
        else if (KIND == 5) asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(v[q]) : "v"(v[q + 4]), "s"(msk));
        else if (KIND == 6) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[q]) : "v"(bp), "v"(v[q + 4]));
        else if (KIND == 7) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[q]) : "v"(lp), "n"((q + 4) * 1024));
        else if (KIND == 8) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(lp), "v"(*reinterpret_cast<unsigned long long*>(&ld[q])), "n"(32768) : "memory");
        else if (KIND == 9) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lp), "v"(ld[q]), "n"(32768) : "memory");
        else if (KIND == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(v[4]));
        else if (KIND == 11) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[q]) : "v"(gsrc + threadIdx.x + ((g * N + f) & 63) * 256));
        else if (KIND == 12) asm volatile("s_nop 0");
        else if (KIND >= 13) {
          // the patch request of conv_xp / conv_xw: 4 lanes share a pixel's 64 bytes, pixels 384 bytes apart; 13: a 6 MB footprint (L2), 14: 600 MB (HBM),
          // 15: the same bytes as whole 1 KiB pieces (HBM footprint)
          const size_t pix = (size_t)blockIdx.x * 9973 + (size_t)(it * 36 + g) * N + f;
          const size_t span = KIND == 13 ? (1u << 14) : (1u << 21);      // pixels
          const size_t p0 = (pix * 64) % span;
          const char* base = reinterpret_cast<const char*>(big);
          const char* ad = KIND == 15 ? base + p0 * 288 + threadIdx.x * 16 : base + (p0 + (threadIdx.x >> 2)) * 384 + (threadIdx.x & 3) * 16 + ((g & 3) * 64);
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[q]) : "v"(ad));
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 7");
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int q = 0; q < 8; ++q) s += v[q];
  for (int q = 0; q < 4; ++q) s += __uint_as_float(rd[q][0]) + __uint_as_float(ld[q][1]) + pk[q];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

static const u4* g_big = nullptr;
template <int KIND, int N>
void run(float* out, const u4* gsrc, long long* clk, const char* what, const u4* big = nullptr) {
  if (big) g_big = big; big = g_big;
  const int iters = 300;
  auto kern = probe<KIND, N>;
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, gsrc, 10, clk, big);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, gsrc, iters, clk, big);
  (void)hipDeviceSynchronize();
  long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 36;
  printf("%-22s x %d per MFMA: %6.1f cycles per MFMA  (%.1f per filler beyond the floor of 32.3)  at %.2f GHz\n", what, N, h[0] / mf,
         N ? (h[0] / mf - 32.3) / N : 0.0, h[0] / (h[1] * 10.0));
}

#define RUNK(K, name) run<K, 2>(out, gsrc, clk, name); run<K, 4>(out, gsrc, clk, name); run<K, 6>(out, gsrc, clk, name);
int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  long long* clk; (void)hipMalloc(&clk, 256 * 16);
  u4* gsrc; (void)hipMalloc(&gsrc, 256 * 256 * 16); (void)hipMemset(gsrc, 0x3c, 256 * 256 * 16);
  u4* big; (void)hipMalloc(&big, (size_t)900 << 20); (void)hipMemset(big, 0x3c, (size_t)900 << 20);
  run<0, 0>(out, gsrc, clk, "none", big);
  run<13, 1>(out, gsrc, clk, "patch-like load, L2"); run<13, 2>(out, gsrc, clk, "patch-like load, L2");
  run<14, 1>(out, gsrc, clk, "patch-like load, HBM"); run<14, 2>(out, gsrc, clk, "patch-like load, HBM");
  run<15, 1>(out, gsrc, clk, "1 KiB pieces, HBM"); run<15, 2>(out, gsrc, clk, "1 KiB pieces, HBM");
  run<11, 1>(out, gsrc, clk, "1 KiB pieces, L2 small");
  RUNK(0, "v_fma_f32") RUNK(1, "v_exp_f32") RUNK(2, "v_rcp_f32") RUNK(3, "v_cvt_pk_f16_f32") RUNK(4, "v_fma_mix_f32") RUNK(5, "v_cndmask_b32 sgpr")
  RUNK(6, "ds_bpermute_b32") RUNK(7, "ds_read_b128") RUNK(8, "ds_write_b64") RUNK(9, "ds_write_b128") RUNK(10, "v_add_f32 dependent")
  RUNK(11, "global_load_dwordx4") RUNK(12, "s_nop 0")
  return 0;
}
