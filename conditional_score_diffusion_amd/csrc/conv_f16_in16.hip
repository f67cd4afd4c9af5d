// conv_f16_in16.hip - instantiations of conv_f16_kernel for fp16 sources (written by gn_apply16).
#include <stdlib.h>

#include "conv_f16_kernel.h"

namespace csd {

template <int MT, int NS, bool MASK, int PWC, int KCS>
static int launch_in16(const Conv16KArgs& k, size_t lds, hipStream_t s) {
  auto kern = conv_f16_kernel<MT, NS, MASK, PWC, true, KCS>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  static const int lds_pad = CSD_TUNE_ENV("CSD_C16_LDS_PAD") ? atoi(CSD_TUNE_ENV("CSD_C16_LDS_PAD")) : 0;   // tuning aid: lower the occupancy
  lds += (size_t)lds_pad;
  hipLaunchKernelGGL(kern, dim3(k.nblocks), dim3(k.nw * 64), lds, s, static_cast<const void*>(k.a.src0),
                     static_cast<const void*>(k.a.src1), reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

template <int MT, int NS, int KCS>
static int pick(const Conv16KArgs& k, const ConvPlan& p, bool mask, hipStream_t s) {
  if (p.PW <= 24) {
    if (mask) return launch_in16<MT, NS, true, 24, KCS>(k, p.lds_bytes, s);
    return launch_in16<MT, NS, false, 24, KCS>(k, p.lds_bytes, s);
  }
  if (mask) return launch_in16<MT, NS, true, 34, KCS>(k, p.lds_bytes, s);
  return launch_in16<MT, NS, false, 34, KCS>(k, p.lds_bytes, s);
}

int conv16_launch_in16(const Conv16KArgs& k, const ConvPlan& p, int ns, bool mask, hipStream_t s) {
#define CSD_IN16_CASE(MT_, NS_, KCS_) if (p.MT == MT_ && ns == NS_ && p.KCS == KCS_) return pick<MT_, NS_, KCS_>(k, p, mask, s);
  CSD_IN16_CASE(4, 1, 3) CSD_IN16_CASE(4, 1, 2) CSD_IN16_CASE(4, 1, 1)
  CSD_IN16_CASE(2, 1, 3) CSD_IN16_CASE(2, 1, 2) CSD_IN16_CASE(2, 1, 1)
  CSD_IN16_CASE(4, 2, 2) CSD_IN16_CASE(4, 2, 1)
  CSD_IN16_CASE(2, 2, 2) CSD_IN16_CASE(2, 2, 1)
#undef CSD_IN16_CASE
  set_error("conv16: no fp16-source kernel for MT=%d ns=%d KCS=%d", p.MT, ns, p.KCS);
  return CSD_ERR_INVALID;
}

}  // namespace csd
