// Feasibility probe for a 128-pixel x 96-cout wave tile (gfx950): ONE wave per SIMD (4 waves per CU, 512 registers), per tap pair
// 24 x v_mfma_f32_32x32x16_f16 + 12 x v_mfma_scale_f32_32x32x64_f8f6f4 on 12 accumulators, their LDS fragment reads (2 x 7 fp16 +
// 7 x 2 fp8 ds_read_b128, conflict-free), optionally NV "conversion-like" vector instructions per pair (fma / exp2 / rcp mix) interleaved.
// Ideal matrix time per pair: 24 * 32 + 12 * 64 = 1536 cycles.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/bigtile_probe.hip -o build/bigtile && build/bigtile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int int8v __attribute__((ext_vector_type(8)));
typedef unsigned int uint4f __attribute__((ext_vector_type(4)));

template <int MODE, int NV>      // MODE 0: MFMAs on register operands; 1: + LDS fragment reads; NV: conversion slots per pair (each ~ 12 VALU incl. 2 transcendentals)
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = 0x38383838 + i % 3;
  __syncthreads();
  half8 a[4], b[3];
  int8v a8[4], b8[3];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(0.001f * lane + j + i); a8[i][j] = 0x38383838; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) { b[i][j] = (_Float16)(1.f - 0.01f * j); b8[i][j] = 0x3c3c3c3c; }
  f16v c[12];
  for (int i = 0; i < 12; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = seed + 0.01f * lane + j;
  const char* base = smem + lane * 16 + wave * 1024;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int o = (it & 1) * 49152;
#pragma unroll
    for (int tap = 0; tap < 2; ++tap) {
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(base + o + (tap * 7 + i) * 4096);
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = *reinterpret_cast<const half8*>(base + o + (tap * 7 + 4 + i) * 4096);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
        if (NV > 0 && (i % 3) == 0) {      // conversion work in the MFMA shadow: NV slots per pair, spread over the 8 x 3 + 12 slots
#pragma unroll
          for (int s = 0; s < (NV + 7) / 8; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              float h = v[j * 4 + (i / 3)] * 1.0001f + 0.5f;
              h = h * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h));
              v[j * 4 + (i / 3)] = h * 0.999f + (float)s;
            }
          }
        }
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint4f x = *reinterpret_cast<const uint4f*>(base + o + 57344 / 2 + i * 2048), y = *reinterpret_cast<const uint4f*>(base + o + 57344 / 2 + i * 2048 + 1024);
        a8[i] = int8v{(int)x.x, (int)x.y, (int)x.z, (int)x.w, (int)y.x, (int)y.y, (int)y.z, (int)y.w};
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const uint4f x = *reinterpret_cast<const uint4f*>(base + o + 40960 + i * 2048), y = *reinterpret_cast<const uint4f*>(base + o + 40960 + i * 2048 + 1024);
        b8[i] = int8v{(int)x.x, (int)x.y, (int)x.z, (int)x.w, (int)y.x, (int)y.y, (int)y.z, (int)y.w};
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 3], b8[i >> 2], c[i], 0, 0, 0, 127, 0, 116);
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int i = 0; i < 12; ++i) r += c[i][i];
  for (int j = 0; j < 8; ++j) r += v[j];
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NV>
static void run(const char* what) {
  float* d; long long* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 8);
  const int iters = 2000;
  auto kern = k<MODE, NV>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 100 * 1024, 0, d, c, iters, 0.5f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 100 * 1024, 0, d, c, iters, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
  printf("%-52s %.3f ms; %.0f cycles per tap pair (matrix ideal 1536: %.0f %%); clock %.2f GHz\n", what, ms, (double)cy / iters,
         100.0 * 1536 / ((double)cy / iters), (double)cy / (ms * 1e6));
  (void)hipFree(d); (void)hipFree(c);
}

int main() {
  run<0, 0>("36 MFMAs, register operands");
  run<1, 0>("36 MFMAs + 28 ds_read_b128");
  run<1, 8>("  + 16 SiLU-like element conversions per pair");
  run<1, 16>("  + 32 element conversions per pair");
  run<1, 32>("  + 64 element conversions per pair");
  run<1, 64>("  + 128 element conversions per pair");
  return 0;
}
