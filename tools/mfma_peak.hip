// Practical ceiling probe: v_mfma_f32_32x32x2_f32 back to back, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  floatx16 acc[3];
  for (int n = 0; n < 3; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
  }
  float s = 0;
  for (int n = 0; n < 3; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
    const int grid = 256 * wgs_per_cu, iters = 20000;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * 12 * 4096.0;
    printf("wgs/cu=%d  %.2f ms  %.1f TFLOP/s (fp32 mfma 32x32x2)\n", wgs_per_cu, ms, fl / ms / 1e9);
  }
  // sustained: ~2 s of back-to-back launches (does the clock hold once the power limiter reacts?)
  {
    const int grid = 256 * 3, iters = 20000;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 20.0 * grid * 4 * iters * 12 * 4096.0;
      printf("sustained rep %d: %.1f ms  %.1f TFLOP/s\n", rep, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
