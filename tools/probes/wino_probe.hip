// Round-5 probe: what would a 1-D Winograd F(2,3) form of conv_xp's stream cost at the chip's power limit?
// One wave per SIMD, one 4-wave workgroup per CU, operands re-read every group from RANDOM fp16 data in LDS (real switching activity).
// A "stage" = 16 input channels of a 16x16-pixel x 96-cout tile in fp16x3 (three MFMAs per product):
//   direct (conv_xp today) : 9 taps     x [18 MFMAs round robin over  6 accumulators, 10 ds_read_b128, fillers]  = 162 MFMAs
//   F(2,3) along the row   : 3 row taps x [36 MFMAs round robin over 12 accumulators, 20 ds_read_b128, fillers]  = 108 MFMAs
//     (a wave owns 2 of the 4 transform components x 2 M tiles of 32 pixel pairs x 3 cout tiles; the pair of waves that share
//      a pixel block exchange one component through LDS in the epilogue)
// fillers per group: F plain v_fma_f32, T transcendentals (v_exp_f32), W ds_write_b64, L buffer_load_dwordx4 (L2 hits), spread evenly
// over the group's gaps.  Filler counts of the real streams (per wave and stage, NORM = true):
//   direct  : 150 plain + 48 transcendental + 26 LDS stores + 20 loads + 90 fragment reads                    (2.1 per MFMA)
//   F(2,3)  : 240 plain + 48 transcendental + 42 LDS stores + 24 loads + 60 fragment reads                    (3.8 per MFMA)
// hipcc --offload-arch=gfx950 -O3 tools/probes/wino_probe.hip -o /tmp/wino_probe && /tmp/wino_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// KS: transform components per wave (each with its own weights), MA: M tiles, NBT: cout tiles; accumulators = KS * MA * NBT, fragments per
// plane: KS * MA pixel-side + KS * NBT weight-side, MFMAs per group = 3 x accumulators
template <int KS, int MA, int NBT, int GROUPS, int F, int T, int W, int L>
__global__ __launch_bounds__(256, 1) void probe(float* out, const u4* gsrc, int iters, long long* clk) {
  constexpr int NA = KS * MA, NB = KS * NBT, NACC = KS * MA * NBT, G = 3 * NACC, R = 2 * (NA + NB);
  const long long c0 = clock64(), w0 = wall_clock64();
  __shared__ __attribute__((aligned(16))) char lds[131072];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768; i += 256) {
    unsigned h = (i + blockIdx.x * 32768) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo16 = (h & 0x3ff) | (((h >> 10) % 12 + 9) << 10) | ((h >> 20 & 1) << 15), hi16 = (h >> 21 & 0x3ff) | (((h >> 3) % 12 + 9) << 10) | ((h >> 31) << 15);
    reinterpret_cast<unsigned*>(lds)[i] = lo16 | (hi16 << 16);
  }
  __syncthreads();
  floatx16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  float tv[2] = {0.3f, 0.7f};
  u4 ld[8];
  for (int q = 0; q < 8; ++q) ld[q] = u4{0, 0, 0, 0};
    half8 fa[2][2][NA], fb[2][2][NB];      // [register buffer][plane][fragment]
  auto rd = [&](int buf, int grp, int r) __attribute__((always_inline)) {
    // fragment r of the group: planes interleaved; a different window per group (compiler-visible LDS reads: hipcc places the waits)
    const int pl = r & 1, idx = r >> 1;
    const int off = ((grp * R + r) % 96) * 1024;
    if (idx < NA) fa[buf][pl][idx] = *reinterpret_cast<const half8*>(lds + off + lane * 16);
    else fb[buf][pl][idx - NA] = *reinterpret_cast<const half8*>(lds + off + lane * 16);
  };
#pragma unroll
  for (int r = 0; r < R; ++r) rd(0, 0, r);
  for (int it = 0; it < iters; it += 2) {      // two stages per iteration (the register buffer parity of an odd group count)
#pragma unroll
    for (int grp2 = 0; grp2 < 2 * GROUPS; ++grp2) {
      const int buf = grp2 & 1, grp = grp2 % GROUPS;
      // the fragments of this group were read during the previous one
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int p = g / NACC, i = g % NACC;
        __builtin_amdgcn_sched_barrier(0);
        const int kc = i / (MA * NBT), mt = (i / NBT) % MA, nt = i % NBT;
        MFMA(acc[i], fa[buf][p == 1 ? 1 : 0][kc * MA + mt], fb[buf][p == 0 ? 1 : 0][kc * NBT + nt]);
        __builtin_amdgcn_sched_barrier(0);
        // fillers of gap g
        const int r0 = R * g / G, r1 = R * (g + 1) / G;
#pragma unroll
        for (int r = r0; r < r1; ++r) rd(buf ^ 1, (grp + 1) % GROUPS, r);
        const int f0 = F * g / G, f1 = F * (g + 1) / G;
#pragma unroll
        for (int f = f0; f < f1; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 3]) : "v"(v[(f + 1) & 3]), "v"(v[(f + 2) & 3]));
        const int t0 = T * g / G, t1 = T * (g + 1) / G;
#pragma unroll
        for (int t = t0; t < t1; ++t) asm volatile("v_exp_f32 %0, %0" : "+v"(tv[t & 1]));
        const int s0 = W * g / G, s1 = W * (g + 1) / G;
#pragma unroll
        for (int s = s0; s < s1; ++s)      // (compiler-visible, so that its lgkmcnt bookkeeping stays exact; data = filler results, not the loads)
          *reinterpret_cast<float2*>(lds + 98304 + threadIdx.x * 8 + (s & 7) * 2048) = make_float2(v[s & 3], tv[s & 1]);
        const int l0 = L * g / G, l1 = L * (g + 1) / G;
#pragma unroll
        for (int l = l0; l < l1; ++l) ld[l & 7] = gsrc[(threadIdx.x + ((it * 16 + grp * 4 + l) & 255) * 256)];      // consumed after the loop
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7");
  float s = v[0] + v[1] + v[2] + v[3] + tv[0] + tv[1];
  for (int q = 0; q < 8; ++q) s += __uint_as_float(ld[q][0]) + __uint_as_float(ld[q][3]);
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = clock64() - c0; clk[blockIdx.x * 2 + 1] = wall_clock64() - w0; }
}

template <int KS, int MA, int NBT, int GROUPS, int F, int T, int W, int L>
double run(float* out, const u4* gsrc, long long* clk, const char* what) {
  const int iters = 600;      // stages
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = probe<KS, MA, NBT, GROUPS, F, T, W, L>;
  constexpr int NA = KS * MA, NB = KS * NBT;
  fprintf(stderr, "launch %s %d %d %d %d: F %d\n", what, KS, MA, NBT, GROUPS, F);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, gsrc, 10, clk);
  { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { fprintf(stderr, "error %s\n", hipGetErrorString(e)); exit(1); } }
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, gsrc, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * GROUPS * 3 * KS * MA * NBT;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ns_stage = ms * 1e6 / iters;
  printf("%-8s %2d acc x %d groups  F %3d T %2d W %2d L %2d per group: %7.1f ns per stage  %5.1f cycles per MFMA at %.3f GHz  (%.2f fillers per MFMA)  issued %4.0f TF/s\n",
         what, KS * MA * NBT, GROUPS, F, T, W, L, ns_stage, h[0] / mf, h[0] / (h[1] * 10.0), (double)(F + T + W + L + 2 * (NA + NB)) / (3 * KS * MA * NBT),
         mf * 256 * 4 * 32768.0 / (ms * 1e-3) / 1e12);
  return ns_stage;
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  long long* clk; hipMalloc(&clk, 256 * 16);
  u4* gsrc; hipMalloc(&gsrc, 256 * 256 * 16); hipMemset(gsrc, 0x3c, 256 * 256 * 16);
  for (int rep = 0; rep < 3; ++rep) {
    // direct: per tap 150/9 plain, 48/9 transcendental, 26/9 stores, 20/9 loads
    const double d0 = run<1, 2, 3, 9, 0, 0, 0, 0>(out, gsrc, clk, "direct");
    const double d1 = run<1, 2, 3, 9, 17, 5, 3, 2>(out, gsrc, clk, "direct");
    // F(2,3), k split over wave pairs: per row tap 80 plain, 16 transcendental, 14 stores, 8 loads
    const double w0 = run<2, 2, 3, 3, 0, 0, 0, 0>(out, gsrc, clk, "wino-B");
    const double w1 = run<2, 2, 3, 3, 60, 16, 14, 8>(out, gsrc, clk, "wino-B");
    const double w2 = run<2, 2, 3, 3, 80, 16, 14, 8>(out, gsrc, clk, "wino-B");
    const double w3 = run<2, 2, 3, 3, 100, 16, 14, 8>(out, gsrc, clk, "wino-B");
    // F(2,3), one component per wave (4 M tiles x 3 cout tiles): fewer fragment reads
    const double x2 = run<1, 4, 3, 3, 80, 16, 14, 8>(out, gsrc, clk, "wino-D");
    // NORM = false (training graph): split only
    const double n0 = run<1, 2, 3, 9, 8, 0, 3, 2>(out, gsrc, clk, "dir-raw");
    const double n1 = run<2, 2, 3, 3, 56, 0, 14, 8>(out, gsrc, clk, "wino-raw");
    printf("  K-loop ratio F(2,3) / direct: bare %.3f, with fillers %.3f (60) %.3f (80) %.3f (100); raw %.3f\n", w0 / d0, w1 / d1, w2 / d1, w3 / d1, n1 / n0);
    (void)x2;
  }
  return 0;
}
