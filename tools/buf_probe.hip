// tuning aid: checks raw buffer load/store semantics (out-of-range offsets dropped) on the device
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int FLAGS>
__global__ void k(float* out, const float* in, unsigned nrec) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, nrec, FLAGS);
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nrec, FLAGS);
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 4;
  float v = __builtin_amdgcn_raw_buffer_load_b32(ri, off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(v + 1.f, r, (threadIdx.x & 1) ? 0x80000000u + threadIdx.x * 4 : off, 0, 0);
}
int main() {
  float *o, *i; hipMalloc(&o, 1024); hipMalloc(&i, 1024);
  float h[64]; for (int j = 0; j < 64; ++j) h[j] = j;
  unsigned nrecs[3] = {0x80000000u, 0x7FFFFFFFu, 256u};
  for (int f = 0; f < 2; ++f) for (unsigned nr : nrecs) {
    hipMemcpy(i, h, 256, hipMemcpyHostToDevice); hipMemset(o, 0, 256);
    if (f == 0) k<0x00020000><<<1, 64>>>(o, i, nr); else k<0x00027000><<<1, 64>>>(o, i, nr);
    float r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    printf("flags %d nrec %x: %g %g %g %g %g %g\n", f, nr, r[0], r[1], r[2], r[3], r[4], r[62]);
  }
}
