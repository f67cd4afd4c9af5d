// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 operands) on gfx950: which (lane, byte) of A pairs with which of B, the
// block-scale semantics and v_cvt_pk_fp8_f32.  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f8_probe.hip -o /tmp/f8probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const uint8_t* a, const uint8_t* b, float* d, int scale_a, int scale_b) {
  const int lane = threadIdx.x;
  v8i av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = reinterpret_cast<const int*>(a + lane * 32)[i];
    bv[i] = reinterpret_cast<const int*>(b + lane * 32)[i];
  }
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 1000.f;      // accumulates into C
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 16; ++i) d[lane * 16 + i] = acc[i];
}

__global__ void cvt(const float* f, uint8_t* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 4 >= n) return;
  int p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(f[i * 4], f[i * 4 + 1], p, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(f[i * 4 + 2], f[i * 4 + 3], p, true);
  reinterpret_cast<int*>(o)[i] = p;
}

static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -r : r;
}

int main() {
  srand(1);
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  for (auto& v : A) { do v = rand() & 255; while ((v & 0x7f) == 0x7f); }
  for (auto& v : B) { do v = rand() & 255; while ((v & 0x7f) == 0x7f); }
  uint8_t *da, *db; float* dd;
  hipMalloc(&da, A.size()); hipMalloc(&db, B.size()); hipMalloc(&dd, 64 * 16 * 4);
  hipMemcpy(da, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(db, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int trial = 0; trial < 2; ++trial) {
    const int sa = trial ? 116 : 127, sb = 127;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
    std::vector<float> D(64 * 16);
    hipMemcpy(D.data(), dd, D.size() * 4, hipMemcpyDeviceToHost);
    // hypothesis: lane l = (kb = l >> 5, row/col = l & 31); byte j of A lane (i, kb) pairs with byte j of B lane (n, kb);
    // D[row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][col = lane & 31] in register r
    double worst = 0, ref_max = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        double acc = 0;
        for (int kb = 0; kb < 2; ++kb)
          for (int j = 0; j < 32; ++j) acc += (double)e4m3(A[(kb * 32 + row) * 32 + j]) * (double)e4m3(B[(kb * 32 + col) * 32 + j]);
        acc = acc * ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127) + 1000.0;
        worst = fmax(worst, fabs(acc - (double)D[lane * 16 + r]));
        ref_max = fmax(ref_max, fabs(acc));
      }
    printf("scale_a=%d: max |gpu - hypothesis| = %.4g (|ref| up to %.4g)\n", sa, worst, ref_max);
  }
  // conversion: round to nearest even, saturation behaviour
  const float in[16] = {0.f, 1.f, 1.0625f, 1.1875f, -3.3f, 447.f, 448.f, 460.f, 500.f, 1e4f, 0.001f, 0.002f, 0.0009765625f, -0.3f, 17.f, 19.f};
  float* df; uint8_t* dob;
  hipMalloc(&df, sizeof(in)); hipMalloc(&dob, 16);
  hipMemcpy(df, in, sizeof(in), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt, dim3(1), dim3(4), 0, 0, df, dob, 16);
  uint8_t ob[16];
  hipMemcpy(ob, dob, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("cvt %g -> 0x%02x = %g\n", in[i], ob[i], e4m3(ob[i]));
  return 0;
}
