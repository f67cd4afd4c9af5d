// Do VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD (gfx950)?  8 waves per workgroup = 2 per SIMD:
// waves 0-3 run an MFMA loop, waves 4-7 nothing / an FMA loop / a conversion-like mix (exp, rcp, cvt) / LDS traffic.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void k(float* out, int n_mfma, int n_valu, int mode_a, int mode_b) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float r = 0.f;
  if (wave < 4) {
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
    float x0 = lane * 0.01f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    if (mode_b == 1) {               // plain FMAs, 4 independent chains
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
          x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
        }
      }
    } else if (mode_b == 2) {        // conversion-like: affine, SiLU (exp + rcp), fp16 round trip
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
    } else if (mode_b == 3) {        // LDS traffic
      for (int it = 0; it < n_valu; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          lds[(wave - 4) * 1024 + ((lane * 4 + u * 256) & 1023)] = x0;
          x0 += lds[(wave - 4) * 1024 + ((lane * 4 + u * 256 + 64) & 1023)];
        }
      }
    }
    r = x0 + x1 + x2 + x3;
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

static float run(int n_mfma, int n_valu, int a, int b) {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, n_mfma, n_valu, a, b);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, n_mfma, n_valu, a, b);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms / 5;
}

int main() {
  const int NM = 20000;
  const float tm = run(NM, 0, 1, 0);
  printf("MFMA only (4 waves/CU-SIMD set, %d x 4 MFMA 32x32x16): %.3f ms -> %.0f TF/s\n", NM, tm, 256.0 * 4 * NM * 4 * 32768.0 / tm / 1e9);
  const char* names[4] = {"", "fma", "silu+cvt mix", "lds"};
  for (int b = 1; b <= 3; ++b) {
    // size the VALU loop so that alone it takes about as long as the MFMA loop
    int nv = 20000;
    float tv = run(0, nv, 0, b);
    nv = (int)(nv * tm / tv);
    tv = run(0, nv, 0, b);
    const float tb = run(NM, nv, 1, b);
    printf("%-13s alone %.3f ms | together %.3f ms | sum %.3f | max %.3f  -> overlap %.0f %%\n", names[b], tv, tb, tm + tv, tm > tv ? tm : tv,
           100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
  }
  return 0;
}
