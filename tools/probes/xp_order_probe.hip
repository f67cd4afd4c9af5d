// Tuning probe for conv_xp: how many "filler" instructions hide behind the MFMAs of ONE wave per SIMD, by MFMA order.
//   order 0: chains of three dependent MFMAs per accumulator (6 chains per tap), fillers after each chain      (6 gaps)
//   order 1: round robin over the six accumulators (an accumulator recurs every 6 MFMAs), fillers after each MFMA (18 gaps)
//   order 2: two accumulators interleaved (A B A B A B), fillers after each MFMA                                  (18 gaps)
//   order 3-5: operands re-read every tap from RANDOM fp16 data in LDS, MFMAs pixel-tile major / cout-tile major / snake
//   order 6-8: as 3, with 3 / 6 / all significand bits of the two "lo" operand planes cleared (bit activity vs clock at the power limit)
// fillers per tap: F v_fma_f32 (four independent chains) + R ds_read_b128 (conflict-free) spread evenly over the gaps.
// hipcc --offload-arch=gfx950 -O3 tools/probes/xp_order_probe.hip -o /tmp/xp_order_probe && /tmp/xp_order_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int ORDER, int F, int R>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, long long* clk) {
  const long long c0 = clock64(), w0 = wall_clock64();
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) {
    // ORDER 3: random fp16 pairs (what real operands look like to the matrix pipe: every bit toggles); otherwise a smooth ramp
    unsigned h = (i + blockIdx.x * 16384) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned lo16 = (h & 0x3ff) | (((h >> 10) % 12 + 9) << 10) | ((h >> 20 & 1) << 15), hi16 = (h >> 21 & 0x3ff) | (((h >> 3) % 12 + 9) << 10) | ((h >> 31) << 15);
    if (ORDER >= 3) reinterpret_cast<unsigned*>(lds)[i] = lo16 | (hi16 << 16);
    else reinterpret_cast<float*>(lds)[i] = i * 1e-6f;
  }
  __syncthreads();
  floatx16 acc[6];
  for (int i = 0; i < 6; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8 a[2], b[3];
  for (int q = 0; q < 8; ++q) { a[0][q] = (_Float16)(lane * 0.01f + q); a[1][q] = (_Float16)(q - lane * 0.02f); b[0][q] = (_Float16)(0.5f * q); b[1][q] = (_Float16)(1.f - q); b[2][q] = (_Float16)(lane & 3); }
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  f4 rd[4];
  for (int q = 0; q < 4; ++q) rd[q] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned lp = (unsigned)(size_t)(lds + lane * 16);      // (LDS aperture: the low 32 bits are the LDS byte address)
  int fcount = 0, rcount = 0;
  auto fill = [&](int gap, int ngaps) __attribute__((always_inline)) {
    // fillers of this gap: an even share of F fmas and R reads
    const int f0 = F * gap / ngaps, f1 = F * (gap + 1) / ngaps, r0 = R * gap / ngaps, r1 = R * (gap + 1) / ngaps;
#pragma unroll
    for (int r = r0; r < r1; ++r) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[r & 3]) : "v"(lp), "n"((r & 15) * 1024));
    }
#pragma unroll
    for (int f = f0; f < f1; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 3]) : "v"(v[(f + 1) & 3]), "v"(v[(f + 2) & 3]));
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (ORDER == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %4, %0"
                       : "+a"(acc[i]) : "v"(a[i / 3]), "v"(b[i % 3]), "v"(a[1 - i / 3]), "v"(b[(i + 1) % 3]));
          fill(i, 6);
        }
      } else if (ORDER >= 3) {
        // operands of this tap come from LDS (10 fragments, a different 10 KB window per tap), read during the previous tap
        half8 fa[2][2], fb[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int m = 0; m < 2; ++m) fa[q][m] = *reinterpret_cast<const half8*>(lds + ((tap * 10 + q * 2 + m) & 63) * 1024 + lane * 16);
#pragma unroll
          for (int m = 0; m < 3; ++m) fb[q][m] = *reinterpret_cast<const half8*>(lds + ((tap * 10 + 4 + q * 3 + m) & 63) * 1024 + lane * 16);
        }
        if (ORDER >= 6) {      // the "lo" fragments with 3 (ORDER 6) / 6 (7) trailing significand bits cleared, or all zero (8): what the
                               // multipliers' switching activity is worth at the power limit
          const unsigned msk = ORDER == 6 ? 0xFFF8FFF8u : ORDER == 7 ? 0xFFC0FFC0u : 0u;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            u4 t = __builtin_bit_cast(u4, fa[1][m]);
            t = u4{t[0] & msk, t[1] & msk, t[2] & msk, t[3] & msk};
            fa[1][m] = __builtin_bit_cast(half8, t);
          }
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            u4 t = __builtin_bit_cast(u4, fb[1][m]);
            t = u4{t[0] & msk, t[1] & msk, t[2] & msk, t[3] & msk};
            fb[1][m] = __builtin_bit_cast(half8, t);
          }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            // ORDER 3: pixel tile major (the A operand repeats three times); 4: cout tile major (B repeats twice, A alternates);
            // 5: snake (A A A' A' ... with B reversed on the way back: one operand changes per MFMA)
            const int i = ORDER == 4 ? (j % 2) * 3 + j / 2 : ORDER == 5 ? (j < 3 ? j : 8 - j) : j;
            MFMA(acc[i], fa[p == 1 ? 1 : 0][i / 3], fb[p == 0 ? 1 : 0][i % 3]);
            fill(p * 6 + j, 18);
          }
      } else if (ORDER == 1) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            MFMA(acc[i], a[(i / 3 + p) & 1], b[(i + p) % 3]);
            fill(p * 6 + i, 18);
          }
      } else {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int i = pr * 2 + h;
              MFMA(acc[i], a[(i / 3 + p) & 1], b[(i + p) % 3]);
              fill(pr * 6 + p * 2 + h, 18);
            }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int q = 0; q < 4; ++q) s += rd[q][0] + rd[q][1] + rd[q][2] + rd[q][3];
  for (int i = 0; i < 6; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = clock64() - c0; clk[blockIdx.x * 2 + 1] = wall_clock64() - w0; }
}

template <int ORDER, int F, int R>
void run(float* out, long long* clk) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<ORDER, F, R>), dim3(256), dim3(256), 0, 0, out, 10, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<ORDER, F, R>), dim3(256), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 9 * 18;
  const double tf = mf * 256 * 4 * 32768.0 / (ms * 1e-3) / 1e12;
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("order %d  fma %3d  ds_read %2d per tap (18 MFMAs): %7.1f us  %6.0f TF/s  %5.1f ns per MFMA  | %.1f cycles per MFMA at %.3f GHz\n", ORDER, F, R, ms * 1e3, tf, ms * 1e6 / mf, h[0] / mf, h[0] / (h[1] * 10.0));
}

template <int ORDER, int F>
void run_long(float* out, long long* clk, int launches) {      // a few seconds of one stream: for tools/power_trace.sh
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((probe<ORDER, F, 0>), dim3(256), dim3(256), 0, 0, out, 400, clk);
  hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("order %d fma %d: %d launches, last one at %.3f GHz\n", ORDER, F, launches, h[0] / (h[1] * 10.0));
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'l') {                // "long": random operands, 36 fillers per tap, ~6 s; then constant operands
    float* out; hipMalloc(&out, 256 * 256 * 4);
    long long* clk; hipMalloc(&clk, 256 * 16);
    run_long<3, 36>(out, clk, 3500);
    run_long<1, 36>(out, clk, 5000);
    return 0;
  }
  float* out; hipMalloc(&out, 256 * 256 * 4);
  long long* clk; hipMalloc(&clk, 256 * 16);
  run<0, 0, 0>(out, clk); run<1, 0, 0>(out, clk); run<2, 0, 0>(out, clk);
  run<0, 18, 10>(out, clk); run<1, 18, 10>(out, clk); run<2, 18, 10>(out, clk);
  run<0, 36, 10>(out, clk); run<1, 36, 10>(out, clk); run<2, 36, 10>(out, clk);
  run<0, 54, 10>(out, clk); run<1, 54, 10>(out, clk); run<2, 54, 10>(out, clk);
  run<0, 72, 10>(out, clk); run<1, 72, 10>(out, clk); run<2, 72, 10>(out, clk);
  run<0, 90, 10>(out, clk); run<1, 90, 10>(out, clk); run<2, 90, 10>(out, clk);
  run<1, 108, 10>(out, clk); run<1, 36, 18>(out, clk);
  run<3, 0, 0>(out, clk); run<3, 36, 0>(out, clk); run<3, 54, 0>(out, clk);
  for (int rep = 0; rep < 3; ++rep) { run<3, 36, 0>(out, clk); run<4, 36, 0>(out, clk); run<5, 36, 0>(out, clk); }
  for (int rep = 0; rep < 3; ++rep) { run<3, 36, 0>(out, clk); run<6, 16, 0>(out, clk); run<7, 16, 0>(out, clk); run<8, 16, 0>(out, clk); }
  return 0;
}
