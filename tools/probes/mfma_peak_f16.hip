// Practical ceiling probe for the fp16 matrix cores: v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 back
// to back, no memory traffic; reports TFLOP/s and the shader clock (clock64 ticks per wall-clock second).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak_f16.hip -o tools/mfma_peak_f16.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters, float a0) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(a0 + threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  float s = 0;
  if (KIND == 0) {
    floatx16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
    }
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  } else {
    floatx4 acc[8];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
    }
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int KIND>
void run(const char* name, double flop_per_iter_per_wave) {
  float* d; hipMalloc(&d, 256 * 4096 * 4);
  long long* c; hipMalloc(&c, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = 256 * wgs_per_cu, iters = 40000;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, c, 100, 1.f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, c, iters, 1.f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
      const double fl = (double)grid * 4 * iters * flop_per_iter_per_wave;
      printf("%s wgs/cu=%d rep %d: %.2f ms  %.0f TFLOP/s   clock64/wall_clock64 = %.3f (x100 MHz = %.0f MHz)\n", name,
             wgs_per_cu, rep, ms, fl / ms / 1e9, (double)h[0] / h[1], 100.0 * h[0] / h[1]);
    }
  }
}
int main() {
  run<0>("32x32x16 f16", 16.0 * 2 * 32 * 32 * 16);
  run<1>("16x16x32 f16", 32.0 * 2 * 16 * 16 * 32);
  return 0;
}
