// Does a chain of dependent v_mfma_f32_32x32x16_f16 (same accumulator back to back - the hi/lo triple of the split mode) issue
// at the independent rate?  KIND 0: 6 accumulators round-robin (dependent distance 6); 1: triples on one accumulator, then the
// next accumulator (distance 1,1,4); 2: one accumulator only (distance 1).   build: hipcc --offload-arch=gfx950 -O3 ... -o x.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(a0 + threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
  floatx16 acc[6];
  for (int n = 0; n < 6; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int n = 0; n < 6; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    } else if (KIND == 1) {
#pragma unroll
      for (int n = 0; n < 6; ++n)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 18; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
    }
  }
  float s = 0;
  for (int n = 0; n < 6; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name) {
  float* d; hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = 256 * wgs_per_cu, iters = 20000;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 18.0 * 2 * 32 * 32 * 16;
    printf("%-28s waves/SIMD=%d: %.2f ms  %.0f TFLOP/s\n", name, wgs_per_cu, ms, fl / ms / 1e9);
  }
}
int main() {
  run<0>("round-robin 6 accumulators");
  run<1>("triples on one accumulator");
  run<2>("single accumulator");
  return 0;
}
