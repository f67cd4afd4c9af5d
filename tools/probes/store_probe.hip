// Write-bandwidth probe (gfx950): 629 MB of fp32 NHWC output (64 x 160 x 160 x 96) written (a) lane-contiguous float4, (b) in the MFMA
// accumulator pattern of the conv epilogues (a lane owns 16 bytes of a pixel's 384-byte row; 32 pixels x 32 bytes per store instruction),
// (c) pattern (b) staged through LDS into full 384-byte rows per 24 lanes.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/store_probe.hip -o build/store_probe && build/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 96, S = 160, B = 64;
__global__ __launch_bounds__(256) void k_lin(float4* out, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// one workgroup per 16 x 8 tile, wave w owns rows 2w, 2w+1 (32 pixels), lane = (kh << 5) | pixel
__global__ __launch_bounds__(256) void k_acc(float* out) {
  const int tile = blockIdx.x, b = tile / 200, tin = tile % 200, ty0 = (tin / 10) * 8, tx0 = (tin % 10) * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, p32 = lane & 31;
  const int oy = ty0 + wave * 2 + (p32 >> 4), ox = tx0 + (p32 & 15);
  float* orow = out + (((size_t)b * S + oy) * S + ox) * C + 4 * kh;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(orow + nt * 32 + q * 8) = make_float4(1.f, 2.f, (float)nt, (float)q);
}
// same tile, but each store instruction writes whole pixel rows: lane l of 24 consecutive lanes writes bytes [16 l, 16 l + 16) of a pixel
__global__ __launch_bounds__(256) void k_row(float* out) {
  const int tile = blockIdx.x, b = tile / 200, tin = tile % 200, ty0 = (tin / 10) * 8, tx0 = (tin % 10) * 16;
  // 128 pixels x 24 float4 = 3072 float4 per tile, 12 per thread
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int i = j * 256 + threadIdx.x;
    const int px = i / 24, c4 = i % 24;
    const int oy = ty0 + (px >> 4), ox = tx0 + (px & 15);
    *reinterpret_cast<float4*>(out + (((size_t)b * S + oy) * S + ox) * C + c4 * 4) = make_float4(1.f, 2.f, (float)j, (float)c4);
  }
}
// transposed accumulator pattern (operands swapped: lane = (kh << 5) | cout, register r = pixel 8 (r / 4) + 4 kh + r % 4): one dword per lane,
// every store instruction writes two whole 128-byte lines
__global__ __launch_bounds__(256) void k_accT(float* out) {
  const int tile = blockIdx.x, b = tile / 200, tin = tile % 200, ty0 = (tin / 10) * 8, tx0 = (tin % 10) * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, c32 = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = 8 * (r >> 2) + 4 * kh + (r & 3);
      const int oy = ty0 + wave * 2 + (p >> 4), ox = tx0 + (p & 15);
      out[(((size_t)b * S + oy) * S + ox) * C + nt * 32 + c32] = (float)r;
    }
}
// the 16x16 MFMA accumulator pattern (pw16 / conv_f16_q): lane = (kq << 4) | pixel, 16 bytes at cout 16 nt + 4 kq: 64-byte pieces
__global__ __launch_bounds__(256) void k_acc16(float* out) {
  const size_t p0 = (size_t)blockIdx.x * 128 + (threadIdx.x >> 6) * 32;
  const int lane = threadIdx.x & 63, kq = lane >> 4, l16 = lane & 15;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) *reinterpret_cast<float4*>(out + (p0 + j * 16 + l16) * C + nt * 16 + kq * 4) = make_float4(1.f, 2.f, (float)nt, (float)j);
}
int main() {
  const size_t n = (size_t)B * S * S * C;
  float* d; (void)hipMalloc(&d, n * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](const char* what, auto launch) {
    launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f us  %.2f TB/s\n", what, ms * 100, n * 4 / (ms / 10 * 1e-3) / 1e12);
  };
  time("lane-contiguous float4, 8192 wgs", [&] { hipLaunchKernelGGL(k_lin, dim3(8192), dim3(256), 0, 0, (float4*)d, n / 4); });
  time("lane-contiguous float4, 1024 wgs", [&] { hipLaunchKernelGGL(k_lin, dim3(1024), dim3(256), 0, 0, (float4*)d, n / 4); });
  time("accumulator pattern (32 B pieces)", [&] { hipLaunchKernelGGL(k_acc, dim3(12800), dim3(256), 0, 0, d); });
  time("pixel rows (384 B per 24 lanes)", [&] { hipLaunchKernelGGL(k_row, dim3(12800), dim3(256), 0, 0, d); });
  time("transposed accumulators (dword, 128 B)", [&] { hipLaunchKernelGGL(k_accT, dim3(12800), dim3(256), 0, 0, d); });
  time("16x16 accumulators (64 B pieces)", [&] { hipLaunchKernelGGL(k_acc16, dim3(12800), dim3(256), 0, 0, d); });
  return 0;
}
