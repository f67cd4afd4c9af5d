// Probe (gfx950): semantics of v_cvt_scalef32_pk_fp8_f32 - does the scale divide or multiply, and does it saturate beyond e4m3's 448
// (v_cvt_pk_fp8_f32 produces NaN there)?  Build: hipcc --offload-arch=gfx950 -O2 tools/probes/cvt_probe.hip -o build/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef short short2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float scale, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // MODE.FP16_OVFL = 1
  const float a = in[threadIdx.x];
  short2v old = {0, 0};
  short2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, a, scale, false);
  const int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false);
  out[threadIdx.x] = ((unsigned)(unsigned short)r[0] & 0xffu) | ((unsigned)(p & 0xff) << 8);
}
static float e4m3(unsigned b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  if (e == 15 && m == 7) return NAN;
  const float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
  return s ? -v : v;
}
int main() {
  const float h[16] = {1.f, 3.f, 447.f, 448.f, 460.f, 500.f, 1e4f, -700.f, 1e-3f, 0.3f, -0.3f, 0.0019f, 2048.f * 0.2f, 2048.f * 0.3f, 1e30f, -1e30f};
  float* din; unsigned* dout;
  hipMalloc(&din, 64); hipMalloc(&dout, 64);
  hipMemcpy(din, h, 64, hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl)
  for (float scale : {1.f, 2048.f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(16), 0, 0, din, dout, scale, ovfl);
    unsigned o[16];
    hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
    printf("scale %g, MODE.FP16_OVFL = %d:\n", scale, ovfl);
    for (int i = 0; i < 16; ++i)
      printf("  in %12g  scalef32 -> 0x%02x = %10g   plain cvt_pk -> 0x%02x = %10g\n", h[i], o[i] & 255, e4m3(o[i] & 255), (o[i] >> 8) & 255, e4m3((o[i] >> 8) & 255));
  }
  return 0;
}
