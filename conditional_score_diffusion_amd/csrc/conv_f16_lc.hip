// conv_f16_lc.hip - instantiations + launch of the loader/consumer fp16-source convolution (conv_f16_lc.h)
#include <stdlib.h>

#include "conv_f16_lc.h"

namespace csd {

size_t conv16_lc_lds_bytes(const ConvPlan& p, int ns, int pitch_px) {
  const int psb = 32 * ns * p.KCS + 16;
  return (size_t)2 * p.PH * pitch_px * psb + (size_t)9 * p.MT * 32 * sizeof(int);
}

template <int MT, int NS, bool MASK, int PWC, int KCS, bool F32 = false>
static int launch_lc(const Conv16KArgs& k, const ConvPlan& p, hipStream_t s) {
  auto kern = conv_f16_lc_kernel<MT, NS, MASK, PWC, KCS, F32>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  size_t lds = conv16_lc_lds_bytes(p, NS, PWC);
  static const int lds_pad = CSD_TUNE_ENV("CSD_C16_LDS_PAD") ? atoi(CSD_TUNE_ENV("CSD_C16_LDS_PAD")) : 0;   // tuning aid: lower the occupancy
  lds += (size_t)lds_pad;
  CSD_REQUIRE(lds <= 160 * 1024 && p.PH * p.PW <= C16_LC_MAXPATCH, "conv16 lc: patch %dx%d does not fit", p.PH, p.PW);
  // persistent: two workgroups per CU; a multiple of 8 keeps every item of a workgroup on its XCD's item range
  int grid = k.nblocks;
  if (grid > 512) grid = 512;
  static const int force_grid = CSD_TUNE_ENV("CSD_C16_LC_GRID") ? atoi(CSD_TUNE_ENV("CSD_C16_LC_GRID")) : 0;   // tuning aid
  if (force_grid >= 8 && force_grid % 8 == 0 && force_grid < k.nblocks) grid = force_grid;
  else if (force_grid >= k.nblocks) grid = k.nblocks;
  hipLaunchKernelGGL(kern, dim3(grid), dim3((k.nw + 1) * 64), lds, s, static_cast<const void*>(k.a.src0),
                     static_cast<const void*>(k.a.src1), reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

template <int MT, int NS, int KCS>
static int pick_lc(const Conv16KArgs& k, const ConvPlan& p, bool mask, hipStream_t s) {
  if (p.PW <= 24) {
    if (mask) return launch_lc<MT, NS, true, 24, KCS>(k, p, s);
    return launch_lc<MT, NS, false, 24, KCS>(k, p, s);
  }
  if (mask) return launch_lc<MT, NS, true, 34, KCS>(k, p, s);
  return launch_lc<MT, NS, false, 34, KCS>(k, p, s);
}

int conv16_launch_lc(const Conv16KArgs& k, const ConvPlan& p, int ns, bool mask, hipStream_t s, bool f32src) {
  CSD_REQUIRE(p.stride == 1 && p.up == 0 && p.KCS > 1, "conv16 lc: stride-1 layers staged in bursts only");
  if (f32src) {
    // fp32 source with the GroupNorm affine + activation fused into the loader
    CSD_REQUIRE(!mask && p.KCS == 2 && p.C0 % 32 == 0 && p.C1 % 32 == 0 && k.a.nscale && k.a.nshift,
                "conv16 lc (fused norm): needs unmasked tiles, 32-channel stages and scale/shift tables");
#define CSD_LCF_CASE(MT_, NS_)                                                              \
  if (p.MT == MT_ && ns == NS_) {                                                           \
    if (p.PW <= 24) return launch_lc<MT_, NS_, false, 24, 2, true>(k, p, s);               \
    return launch_lc<MT_, NS_, false, 34, 2, true>(k, p, s);                               \
  }
    CSD_LCF_CASE(4, 1) CSD_LCF_CASE(2, 1) CSD_LCF_CASE(4, 2) CSD_LCF_CASE(2, 2)
#undef CSD_LCF_CASE
    set_error("conv16 lc (fused norm): no kernel for MT=%d ns=%d", p.MT, ns);
    return CSD_ERR_INVALID;
  }
  CSD_REQUIRE(p.C1 == 0, "conv16 lc: fp16 sources are single-tensor");
#define CSD_LC_CASE(MT_, NS_, KCS_) \
  if (p.MT == MT_ && ns == NS_ && p.KCS == KCS_) return pick_lc<MT_, NS_, KCS_>(k, p, mask, s);
  CSD_LC_CASE(4, 1, 3) CSD_LC_CASE(4, 1, 2) CSD_LC_CASE(2, 1, 3) CSD_LC_CASE(2, 1, 2)
  CSD_LC_CASE(4, 2, 2) CSD_LC_CASE(2, 2, 2)
#undef CSD_LC_CASE
  set_error("conv16 lc: no kernel for MT=%d ns=%d KCS=%d", p.MT, ns, p.KCS);
  return CSD_ERR_INVALID;
}

}  // namespace csd
