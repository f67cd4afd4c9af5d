// conv_f16.hip - host side of the fp16-MFMA convolution (planning, weight packing, launch) + the
// instantiations for fp32 sources; the fp16-source (IN16) instantiations live in conv_f16_in16.hip.
#include "conv_f16_kernel.h"

namespace csd {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int in_coord16(int o, int r, int S, int P, int U) {
  int v = o * S + r - P;
  return v >= 0 ? (v >> U) : -((-v + (1 << U) - 1) >> U);
}

bool conv16_supported(const ConvPlan& p) {
  const int Cin = p.C0 + p.C1;
  return p.taps == 9 && (p.stride == 1 || (p.stride == 2 && p.up == 0)) && Cin % C16_KC == 0 && p.C0 % C16_KC == 0;
}

int conv16_kcs(int ns, int cin) {   // sub-chunks per stage for an fp16-source convolution
  if (const char* f = CSD_TUNE_ENV("CSD_FORCE_KCS")) {   // tuning aid
    const int v = atoi(f);
    if (v == 1 || (cin % (16 * v) == 0 && (v == 2 || (v == 3 && ns == 1)))) return v;
  }
  if (ns == 1) return cin % 48 == 0 ? 3 : (cin % 32 == 0 ? 2 : 1);
  return cin % 32 == 0 ? 2 : 1;
}

int conv16_plan_tiles(ConvPlan* p, int ns, int kcs, bool lc) {
  lc = lc && kcs > 1 && p->stride == 1 && p->up == 0;
  const int Cin = p->C0 + p->C1;
  CSD_REQUIRE(conv16_supported(*p), "conv16: unsupported shape (taps=%d stride=%d Cin=%d+%d)", p->taps, p->stride,
              p->C0, p->C1);
  CSD_REQUIRE((p->IH << p->up) == p->OH * p->stride && (p->IW << p->up) == p->OW * p->stride, "conv16: size mismatch");
  p->KC = C16_KC;
  const int ntiles = cdiv(p->Cout, 32);
  p->CoutPad = ntiles * 32;
  const int KS = p->taps == 9 ? 3 : 1;
  auto extent = [&](int t) {
    return in_coord16(t - 1, KS - 1, p->stride, p->pad, p->up) - in_coord16(0, 0, p->stride, p->pad, p->up) + 1 +
           (p->up ? 1 : 0);
  };
  // waves (= cout tiles) per workgroup: 3 when the tile count allows, else 2 / 1
  // (a single-tile layer - the 96 -> 3 output conv - still runs 3 waves: staging bandwidth, not MFMA, is its limit;
  // the two extra waves recompute tile 0 and store nothing)
  const int nw = (ntiles % 3 == 0 || ntiles == 1) ? 3 : (ntiles % 2 == 0) ? 2 : 1;
  p->NT = nw;                       // (field reused: cout tiles per workgroup)
  p->n_groups = cdiv(ntiles, nw);
  // pixels per workgroup MT*32, MT in {8,4,2}: the largest that still yields >= ~2 workgroups per CU;
  // tile shape: TW divides OW when possible, maximise covered pixels, then minimise the staged patch
  int best_mt = 0, best_tw = 0, best_th = 0;
  int mt_max = 4;     // (MT = 8 is not instantiated) 128-pixel tiles: 4 workgroups (12 waves) per CU overlap each other's prologue/epilogue
  if (const char* f = CSD_TUNE_ENV("CSD_FORCE_MT")) {   // tuning aid
    const int v = atoi(f);
    if (v == 2 || v == 4) mt_max = v;
  }
  const int psb = 32 * ns * kcs + 16;
  const int nbuf = (kcs > 1 && !lc) ? 1 : 2;
  const int max_units = kcs > 1 ? (ns == 1 ? 7 : 9) : C16_MAX_PIECES * C16_PIECE;
  const int spp = kcs > 1 ? 2 * ns * kcs : 4;
  auto pitch = [&](int tw) { return extent(tw) <= 24 ? 24 : 34; };
  for (int mt = mt_max; mt >= 2; mt >>= 1) {
    const int npix = mt * 32;
    int btw = 0, bth = 0, blds = 1 << 30;
    double bcov = -1;
    bool bunm = false;
    for (int tw = 1; tw <= 32 && tw <= p->OW; ++tw) {
      if (p->OW % tw != 0 && !(tw == 32 && p->OW > 32)) continue;
      if (extent(tw) > 34) continue;
      for (int th = npix / tw; th >= 1; --th) {
        const int patch = extent(th) * extent(tw);
        const int lds = nbuf * extent(th) * pitch(tw) * psb;
        if (lds > C16_LDS_LIMIT) continue;
        if (lc ? patch > 208 : patch * spp > max_units * nw * 64) continue;   // (208 = C16_LC_MAXPATCH)
        // valid output pixels per tile (a TW that does not divide OW wastes the last tile of every row), then
        // tiles that stay inside one sample (no masks, fused GroupNorm statistics), then the smaller patch
        const double cov = (double)th * tw * p->OW / ((double)cdiv(p->OW, tw) * tw);
        const bool unm = (p->OH % th) == 0;
        if (cov > bcov + 1e-9 || (cov > bcov - 1e-9 && ((unm && !bunm) || (unm == bunm && lds < blds)))) {
          bcov = cov; blds = lds; btw = tw; bth = th; bunm = unm;
        }
        break;
      }
    }
    if (btw == 0) continue;
    best_mt = mt; best_tw = btw; best_th = bth;
    const long nwg = (long)cdiv(p->OW, btw) * cdiv(p->B * p->OH, bth) * p->n_groups;
    if (nwg >= 512 || mt == 2) break;
  }
  CSD_REQUIRE(best_tw > 0, "conv16: no feasible tile for OW=%d", p->OW);
  p->TW = best_tw; p->TH = best_th;
  p->KC = C16_KC;
  p->PH = extent(p->TH); p->PW = extent(p->TW);
  p->tiles_x = cdiv(p->OW, p->TW);
  p->tiles_y = cdiv(p->B * p->OH, p->TH);
  const int mt_sel = best_mt;
  p->KCS = kcs;
  p->LC = lc ? 1 : 0;
  p->lds_bytes = (size_t)nbuf * p->PH * pitch(p->TW) * psb + (size_t)2 * mt_sel * 32 * sizeof(int) +
                 (size_t)3 * p->PH * p->PW * sizeof(int);
  p->MT = mt_sel;
  CSD_REQUIRE(p->lds_bytes <= 160 * 1024, "conv16: LDS budget exceeded");
  (void)Cin;
  return CSD_OK;
}

size_t conv16_packed_bytes(const ConvPlan& p, int ns) {
  const int Cin = p.C0 + p.C1;
  return (size_t)(p.CoutPad / 32) * (Cin / C16_KC) * p.taps * ns * 1024 + (size_t)9 * ns * 1024;   // + prefetch slack
}

__global__ void conv16_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ wpack, int layout, int cin_src,
                                   int cout_src, int cout_off, int taps, int nck, int ns, int nt_lo, int nt_hi) {
  // one thread per (tile, chunk, tap, lane, q): writes the hi (and lo) half of one weight
  const size_t per_tile = (size_t)nck * taps * 512;       // weights per cout tile
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_tile * (nt_hi - nt_lo)) return;
  size_t rem = idx;
  const int nt = nt_lo + (int)(rem / per_tile); rem %= per_tile;
  const int ck = (int)(rem / ((size_t)taps * 512)); rem %= (size_t)taps * 512;
  const int tap = (int)(rem / 512); rem %= 512;
  const int lane = (int)(rem / 8), q = (int)(rem % 8);
  const int cout = nt * 32 + (lane & 31) - cout_off;
  const int cin = ck * C16_KC + (lane >> 5) * 8 + q;
  if (cout < 0 || cout >= cout_src || cin >= cin_src) return;     // padding stays zero
  const float v = ((layout == 0) ? w[((size_t)cout * cin_src + cin) * taps + tap]
                   : (layout == 1) ? w[(size_t)cin * cout_src + cout]
                                   : w[((size_t)cin * cout_src + cout) * taps + (taps - 1 - tap)]) *
                  C16_WSCALE;
  const size_t step = ((size_t)nt * nck + ck) * taps + tap;
  _Float16* dst = wpack + step * (size_t)ns * 512 + lane * 8 + q;
  const _Float16 hi = (_Float16)v;
  dst[0] = hi;
  if (ns == 2) dst[512] = (_Float16)(v - (float)hi);
}

__global__ void conv16_zero_kernel(uint32_t* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

int conv16_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                       void* wpack, hipStream_t s) {
  const int Cin = p.C0 + p.C1;
  const int nck = Cin / C16_KC;
  if (cout_off == 0) {
    const size_t n32 = conv16_packed_bytes(p, ns) / 4;
    hipLaunchKernelGGL(conv16_zero_kernel, dim3((unsigned)cdiv64(n32, 256)), dim3(256), 0, s, (uint32_t*)wpack, n32);
    CSD_LAUNCH_CHECK();
  }
  const int nt_lo = cout_off / 32, nt_hi = cdiv(cout_off + cout_src, 32);
  const size_t total = (size_t)nck * p.taps * 512 * (nt_hi - nt_lo);
  hipLaunchKernelGGL(conv16_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, w, (_Float16*)wpack,
                     layout, cin_src, cout_src, cout_off, p.taps, nck, ns, nt_lo, nt_hi);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

template <int MT, int NS, bool MASK, int PWC>
static int launch16(const Conv16KArgs& k, size_t lds, hipStream_t s) {
  auto kern = conv_f16_kernel<MT, NS, MASK, PWC, false, 1>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(k.nblocks), dim3(k.nw * 64), lds, s, static_cast<const void*>(k.a.src0),
                     static_cast<const void*>(k.a.src1), reinterpret_cast<const char*>(k.a.wpack), k);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int conv16_launch_in16(const Conv16KArgs& k, const ConvPlan& p, int ns, bool mask, hipStream_t s);   // conv_f16_in16.hip
int conv16_launch_lc(const Conv16KArgs& k, const ConvPlan& p, int ns, bool mask, hipStream_t s, bool f32src);     // conv_f16_lc.hip

long long* g_c16_dbg = nullptr;     // tuning builds only
int g_c16_dbg_blocks = 0;
int g_c16_dbg_max = 0, g_c16_dbg_n = 0;

int conv16_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s, bool in16) {
  Conv16KArgs k;
  k.a = a;
  k.a.dbg = nullptr;
  k.B = p.B; k.IH = p.IH; k.IW = p.IW; k.OH = p.OH; k.OW = p.OW;
  k.C0 = p.C0; k.C1 = p.C1; k.Cout = p.Cout;
  k.stride = p.stride; k.pad = p.pad; k.up = p.up;
  k.TH = p.TH; k.TW = p.TW; k.PH = p.PH; k.PW = p.PW;
  k.tiles_x = p.tiles_x; k.n_groups = p.n_groups;
  k.nblocks = p.tiles_x * p.tiles_y * p.n_groups;
  k.nck = (p.C0 + p.C1) / C16_KC;
  k.nw = p.NT;
  k.ntiles_n = p.CoutPad / 32;
  CSD_REQUIRE(p.taps == 9, "conv16: only 3x3 kernels");
  CSD_REQUIRE(!in16 || (p.C1 == 0 && a.nscale == nullptr), "conv16: fp16 sources are single-tensor and pre-normalised");
  CSD_REQUIRE(in16 || p.KCS == 1 || p.LC, "conv16: multi-chunk stages need an fp16 source");
  // masks are needed iff a tile can straddle two images (or the x2-upsample parity map is in use)
  const bool mask = (p.OH % p.TH) != 0 || p.up != 0;
  if (g_c16_dbg && k.nblocks == g_c16_dbg_blocks && g_c16_dbg_n < g_c16_dbg_max) {   // (tuning builds)
    k.a.dbg = g_c16_dbg + (size_t)g_c16_dbg_n * 4096 * 8;
    if (hipMemsetAsync(k.a.dbg, 0, 4096 * 8 * 8, s) != hipSuccess) return -1;
    long long hdr[4] = {p.C0 + p.C1, p.Cout, in16 ? 1 : 0, (a.res ? 1 : 0) | (a.temb ? 2 : 0)};
    if (hipMemcpyAsync(k.a.dbg + 4095 * 8, hdr, sizeof(hdr), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    g_c16_dbg_n++;
  }
  if (p.LC) return conv16_launch_lc(k, p, ns, mask, s, !in16);
  if (in16) return conv16_launch_in16(k, p, ns, mask, s);
#define CSD_C16_CASE(MT_, NS_)                                                   \
  if (p.MT == MT_ && ns == NS_) {                                                \
    if (p.PW <= 24) {                                                            \
      if (mask) return launch16<MT_, NS_, true, 24>(k, p.lds_bytes, s);          \
      return launch16<MT_, NS_, false, 24>(k, p.lds_bytes, s);                   \
    }                                                                            \
    if (mask) return launch16<MT_, NS_, true, 34>(k, p.lds_bytes, s);            \
    return launch16<MT_, NS_, false, 34>(k, p.lds_bytes, s);                     \
  }
  CSD_C16_CASE(4, 1) CSD_C16_CASE(2, 1)
  CSD_C16_CASE(4, 2) CSD_C16_CASE(2, 2)
#undef CSD_C16_CASE
  set_error("conv16: no kernel for MT=%d ns=%d", p.MT, ns);
  return CSD_ERR_INVALID;
}

}  // namespace csd

extern "C" int csd_debug_timing(long long* buf, int nblocks, int max_launches) {   // tuning aid, not part of include/csd.h
  csd::g_c16_dbg = buf;
  csd::g_c16_dbg_blocks = nblocks;
  csd::g_c16_dbg_max = max_launches;
  csd::g_c16_dbg_n = 0;
  return 0;
}
