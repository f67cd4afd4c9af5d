// Shader-clock probe: ratio of s_memtime-class shader cycles (clock64) to the constant 100 MHz
// wall clock (wall_clock64) inside an MFMA-heavy kernel, with and without LDS/global traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, long long* clk, int iters) {
  __shared__ float lds[8192];
  floatx16 acc[3];
  for (int n = 0; n < 3; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
  __syncthreads();
  float a = threadIdx.x, b = 2.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  const float* p = in + (blockIdx.x % 64) * 65536 + threadIdx.x * 4;
  for (int i = 0; i < iters; ++i) {
    float4 bv = make_float4(b, b, b, b), av = make_float4(a, a, a, a);
    if (MODE >= 1) av = *reinterpret_cast<const float4*>(&lds[((i * 64 + threadIdx.x) * 4) & 8188]);
    if (MODE >= 2) bv = *reinterpret_cast<const float4*>(p + ((i * 1024) & 65535 & ~1023));
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[n], 0, 0, 0);
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int n = 0; n < 3; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int MODE> void run(float* d, float* in, long long* clk, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, in, clk, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int l = 0; l < 10; ++l) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, in, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double fl = 10.0 * grid * 4 * iters * 12 * 4096.0;
  printf("mode %d grid %d: %.1f TFLOP/s  clock64/wall = %.3f (x100MHz => %.0f MHz if clock64 counts shader cycles)\n",
         MODE, grid, fl / ms / 1e9, (double)h[0] / h[1], 100.0 * h[0] / h[1]);
}
int main() {
  float *d, *in; long long* clk;
  hipMalloc(&d, 256 * 4096 * 4); hipMalloc(&in, 64 * 65536 * 4 + 65536); hipMalloc(&clk, 4096 * 16);
  hipMemset(in, 0, 64 * 65536 * 4 + 65536);
  for (int g = 1; g <= 3; g += 2) { run<0>(d, in, clk, 256 * g); run<1>(d, in, clk, 256 * g); run<2>(d, in, clk, 256 * g); }
  return 0;
}
