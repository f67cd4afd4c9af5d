// Floor of conv_ff's inner loop (gfx950): the exact matrix-instruction mix of one tap pair of the fp16f8 form - 12 x v_mfma_f32_32x32x16_f16
// + 6 x v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands) on 6 accumulators - alone, and with the loop's ds_read_b128 traffic
// (5 + 5 fp16 fragments, 10 fp8 fragments per pair) from a conflict-free LDS layout, 1 / 2 waves per SIMD, no barriers.
// Ideal: 12 * 32 + 6 * 64 = 768 cycles per pair and wave.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_mix_probe.hip -o build/mix && build/mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int int8v __attribute__((ext_vector_type(8)));
typedef unsigned int uint4f __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: fp16 MFMAs only; 1: fp16 + fp8 mix, operands in registers; 2: mix + LDS fragment reads; 3: fp8 only
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters, int nwave) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nwave) return;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = 0x38383838 + i % 3;
  __syncthreads();
  half8 a[3], b[2];
  int8v a8[3], b8[2];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(0.001f * lane + j + i); a8[i][j] = 0x38383838; } }
  for (int i = 0; i < 2; ++i) { for (int j = 0; j < 8; ++j) { b[i][j] = (_Float16)(1.f - 0.01f * j); b8[i][j] = 0x3c3c3c3c; } }
  f16v c[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
  const char* base = smem + lane * 16 + wave * 4096;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const half8*>(base + i * 1024 + (it & 1) * 8192);
#pragma unroll
      for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const half8*>(base + 3072 + i * 1024 + (it & 1) * 8192);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const uint4f x = *reinterpret_cast<const uint4f*>(base + 16384 + i * 2048), y = *reinterpret_cast<const uint4f*>(base + 16384 + i * 2048 + 1024);
        a8[i] = int8v{(int)x.x, (int)x.y, (int)x.z, (int)x.w, (int)y.x, (int)y.y, (int)y.z, (int)y.w};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4f x = *reinterpret_cast<const uint4f*>(base + 24576 + i * 2048), y = *reinterpret_cast<const uint4f*>(base + 24576 + i * 2048 + 1024);
        b8[i] = int8v{(int)x.x, (int)x.y, (int)x.z, (int)x.w, (int)y.x, (int)y.y, (int)y.z, (int)y.w};
      }
    }
    if (MODE != 3) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % 3], b[i / 3], c[i], 0, 0, 0);
      if (MODE == 2) {       // the second tap's fp16 fragments
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const half8*>(base + 4096 + i * 1024 + (it & 1) * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const half8*>(base + 7168 + i * 1024 - 3072 + (it & 1) * 8192);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % 3], b[i / 3], c[i], 0, 0, 0);
    }
    if (MODE != 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i % 3], b8[i / 3], c[i], 0, 0, 0, 116, 0, 127);
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int i = 0; i < 6; ++i) r += c[i][i];
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(int nwave, const char* what, double ideal) {
  float* d; long long* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 8);
  const int iters = 4000;
  auto kern = k<MODE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, d, c, iters, nwave);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, d, c, iters, nwave);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long cy; (void)hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
  const int wps = nwave / 4;
  printf("%-34s %d wave(s)/SIMD: %.3f ms; %.0f wave-cycles per iteration (ideal %.0f per wave -> %.0f per SIMD); clock %.2f GHz; density %.0f %%\n", what, wps,
         ms, (double)cy / iters, ideal, ideal * wps, (double)cy / (ms * 1e6), 100.0 * ideal * wps / ((double)cy / iters));
  (void)hipFree(d); (void)hipFree(c);
}

int main() {
  for (int nw : {4, 8}) {
    run<0>(nw, "12 x fp16 32x32x16", 384);
    run<3>(nw, "6 x fp8 scale 32x32x64", 384);
    run<1>(nw, "12 fp16 + 6 fp8, register operands", 768);
    run<2>(nw, "12 fp16 + 6 fp8 + 20 ds_read_b128", 768);
  }
  return 0;
}
