// Follow-up to mfma_valu_overlap.hip (which found a second wave's VALU stream overlapping a wave's dense MFMA stream by only ~15 %):
// is that the issue arbiter (oldest / highest priority first, an MFMA that waits for the busy matrix pipe holding the slot)?
// 8 waves per workgroup = 2 per SIMD; one half runs a dense v_mfma_f32_32x32x16_f16 loop, the other half a VALU loop (fma or the
// SiLU + split mix of conv_ff's prologue).  Variants: which half is dispatched first (age), s_setprio of each role.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_prio.hip -o /tmp/prio && /tmp/prio
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode_a: 0 none, 1 dense MFMA (4 accumulators round robin), 2 MFMA pairs on the same accumulator
// mode_b: 0 none, 1 fma chains, 2 SiLU + fp16 split mix
__global__ __launch_bounds__(512, 1) void k(float* out, int n_mfma, int n_valu, int mode_a, int mode_b, int swap, int prio_a, int prio_b) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool role_a = swap ? wave >= 4 : wave < 4;
  float r = 0.f;
  if (role_a) {
    if (prio_a == 1) __builtin_amdgcn_s_setprio(1);
    if (prio_a == 3) __builtin_amdgcn_s_setprio(3);
    if (mode_a) {
      half8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.f - i * 0.01f); }
      f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int it = 0; it < n_mfma; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else {
    if (prio_b == 1) __builtin_amdgcn_s_setprio(1);
    if (prio_b == 3) __builtin_amdgcn_s_setprio(3);
    float x0 = lane * 0.01f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    if (mode_b == 1) {
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
          x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
        }
      }
    } else if (mode_b == 2) {
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float g0 = x0 * 1.01f + 0.1f, g1 = x1 * 1.01f + 0.1f, g2 = x2 * 1.01f + 0.1f, g3 = x3 * 1.01f + 0.1f;
          g0 = g0 * __builtin_amdgcn_rcpf(1.f + __expf(-g0)); g1 = g1 * __builtin_amdgcn_rcpf(1.f + __expf(-g1));
          g2 = g2 * __builtin_amdgcn_rcpf(1.f + __expf(-g2)); g3 = g3 * __builtin_amdgcn_rcpf(1.f + __expf(-g3));
          const _Float16 h0 = (_Float16)g0, h1 = (_Float16)g1, h2 = (_Float16)g2, h3 = (_Float16)g3;
          x0 = g0 - (float)h0 + 0.3f; x1 = g1 - (float)h1 + 0.3f; x2 = g2 - (float)h2 + 0.3f; x3 = g3 - (float)h3 + 0.3f;
        }
      }
    }
    r = x0 + x1 + x2 + x3;
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

static float run(int n_mfma, int n_valu, int a, int b, int swap, int pa, int pb) {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, n_mfma, n_valu, a, b, swap, pa, pb);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, n_mfma, n_valu, a, b, swap, pa, pb);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms / 5;
}

int main() {
  const int NM = 20000;
  const float tm = run(NM, 0, 1, 0, 0, 0, 0);
  printf("MFMA only: %.3f ms -> %.0f TF/s\n", tm, 256.0 * 4 * NM * 4 * 32768.0 / tm / 1e9);
  const char* names[3] = {"", "fma", "silu+cvt"};
  for (int b = 1; b <= 2; ++b) {
    int nv = 20000;
    float tv = run(0, nv, 0, b, 0, 0, 0);
    nv = (int)(nv * tm / tv);
    tv = run(0, nv, 0, b, 0, 0, 0);
    for (int swap = 0; swap < 2; ++swap)
      for (int pa : {0, 3})
        for (int pb : {0, 3}) {
          if (pa && pb) continue;
          const float tb = run(NM, nv, 1, b, swap, pa, pb);
          printf("%-9s %s prio(mfma)=%d prio(valu)=%d: valu alone %.3f | together %.3f | sum %.3f -> overlap %.0f %%\n", names[b],
                 swap ? "valu waves older" : "mfma waves older", pa, pb, tv, tb, tm + tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
        }
    // VALU at half / quarter of the MFMA time: does it disappear entirely?
    for (int frac : {2, 4}) {
      const float tb0 = run(NM, nv / frac, 1, b, 0, 0, 0), tb1 = run(NM, nv / frac, 1, b, 0, 0, 3), tb2 = run(NM, nv / frac, 1, b, 1, 0, 0);
      printf("%-9s valu = 1/%d of the mfma time: together %.3f (default) %.3f (valu prio 3) %.3f (valu older) vs mfma alone %.3f\n", names[b], frac,
             tb0, tb1, tb2, tm);
    }
  }
  return 0;
}
