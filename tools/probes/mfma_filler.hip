// How many VALU instructions of the SAME wave hide in the shadow of its MFMAs (gfx950)?  Each wave runs
//     loop { 4 x ( v_mfma_f32_32x32x16_f16 on accumulator c_i ; NF x v_fma_f32 on independent chains ) }
// with the order pinned by inline asm; 1 or 2 waves per SIMD.  Prints cycles per MFMA for NF = 0 .. 12 and the FMA-only time.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_filler.hip -o /tmp/mf && /tmp/mf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NF, bool MFMA>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.f - i * 0.01f); }
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x[12];
  for (int i = 0; i < 12; ++i) x[i] = lane * 0.01f + i;
  const float m = 1.0001f, ad = 0.5f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define FILL(BASE)                                                                                            \
    _Pragma("unroll") for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(BASE + f) % 12]) : "v"(m), "v"(ad));
#define MM(C) if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));
    MM(c0) FILL(0) MM(c1) FILL(3) MM(c2) FILL(6) MM(c3) FILL(9)
  }
  const long long t1 = clock64();
  float r = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 12; ++i) r += x[i];
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NF, bool MFMA>
static void run(int threads, const char* what) {
  float* d; long long* c; hipMalloc(&d, 4096); hipMalloc(&c, 8);
  const int iters = 5000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NF, MFMA>), dim3(256), dim3(threads), 0, 0, d, c, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NF, MFMA>), dim3(256), dim3(threads), 0, 0, d, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  printf("%s NF=%2d waves/SIMD=%d: %.3f ms, %.1f wave-cycles per group (1 MFMA + NF fma), %.1f SIMD-cycles per MFMA\n", what, NF, wps, ms,
         (double)cy / (iters * 4.0), (double)cy / (iters * 4.0) / wps);
  hipFree(d); hipFree(c);
}

int main() {
  for (int threads : {256, 512}) {
    run<0, true>(threads, "mfma+fill");
    run<2, true>(threads, "mfma+fill");
    run<4, true>(threads, "mfma+fill");
    run<5, true>(threads, "mfma+fill");
    run<6, true>(threads, "mfma+fill");
    run<8, true>(threads, "mfma+fill");
    run<12, true>(threads, "mfma+fill");
    run<4, false>(threads, "fill only");
    run<8, false>(threads, "fill only");
    run<12, false>(threads, "fill only");
  }
  return 0;
}
