// common.h - shared declarations for libcsd_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/csd.h"

// Tuning switches (kernel selection A/B, occupancy pads, ablations) exist in the TUNING build only (`make tune` ->
// libcsd_hip_tune.so, selected with CSD_LIB_PATH): in the product library CSD_TUNE_ENV is a null constant, the branches it guards
// are dead code and nothing in libcsd_hip.so depends on the process environment - except the three weight-gradient A/B switches
// that tests/test_gpu_training.py::test_weight_gradient_ab_schedules_agree exercises (backward.hip, wgrad_bf16.hip).
#ifdef CSD_TUNE
#define CSD_TUNE_ENV(name) getenv(name)
#else
#define CSD_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace csd {

// thread-local last-error text behind csd_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

#define CSD_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      csd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return CSD_ERR_HIP;                                                           \
    }                                                                               \
  } while (0)

#define CSD_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      csd::set_error(__VA_ARGS__);             \
      return CSD_ERR_INVALID;                  \
    }                                          \
  } while (0)

#define CSD_LAUNCH_CHECK()                                                          \
  do {                                                                              \
    hipError_t _e = hipGetLastError();                                              \
    if (_e != hipSuccess) {                                                         \
      csd::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      return CSD_ERR_HIP;                                                           \
    }                                                                               \
  } while (0)

// Per-device launch state (a process may drive several GPUs): the CU count of the current device, rounded down to a multiple of the 8
// XCDs, and a per-(kernel instantiation, device) flag for attributes that are per device (hipFuncAttributeMaxDynamicSharedMemorySize).
#define CSD_MAX_DEVICES 64
static inline int current_device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CSD_MAX_DEVICES) dev = 0;
  return dev;
}
static inline int device_cu_count8() {
  static int n_cu[CSD_MAX_DEVICES] = {0};
  const int dev = current_device_index();
  if (!n_cu[dev]) {
    hipDeviceProp_t prop;
    int n = 8;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount / 8 * 8;
    n_cu[dev] = n < 8 ? 8 : n;
  }
  return n_cu[dev];
}
// once per (call site = kernel instantiation, device): allow up to 160 KB of dynamic LDS
#define CSD_SET_MAX_LDS_ONCE(kern)                                                                                       \
  do {                                                                                                                   \
    static bool _set[CSD_MAX_DEVICES] = {false};                                                                         \
    const int _dev = csd::current_device_index();                                                                        \
    if (!_set[_dev]) {                                                                                                   \
      CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      _set[_dev] = true;                                                                                                 \
    }                                                                                                                    \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// Workgroup copy global -> LDS of `bytes` (a multiple of 16): DEPTH 16-byte loads per thread IN FLIGHT before the first store.  The plain
// loop `for (i = tid * 16; i < bytes; i += THREADS * 16) lds[i] = src[i]` compiles to load / s_waitcnt vmcnt(0) / ds_write per
// iteration - one L2 round trip (~0.7 us) per 4 KiB of the workgroup: the 72 - 144 KiB weight copy of a pw16 workgroup was 12 - 25 us of
// every launch, and the whole ~20 us floor of its launches on the small maps (NOTEBOOK round 6).  Addresses are clamped, not branched
// around: a guarded load sits in its own basic block behind a wait.
template <int THREADS, int DEPTH = 8>
__device__ __forceinline__ void wg_copy_to_lds(char* __restrict__ dst, const char* __restrict__ src, int bytes, int tid) {
  typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
  for (int i0 = tid * 16; i0 < bytes; i0 += THREADS * 16 * DEPTH) {
    u4_t v[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int i = i0 + u * THREADS * 16;
      v[u] = *reinterpret_cast<const u4_t*>(src + (i < bytes ? i : i0));
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int i = i0 + u * THREADS * 16;
      if (i < bytes) *reinterpret_cast<u4_t*>(dst + i) = v[u];
    }
  }
}
#endif
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------
// Convolution as implicit GEMM on the fp32 matrix cores (conv_f32.hip)
//   M = output pixels (B*OH*OW), N = Cout, K = taps*Cin.  Activations NHWC fp32.
// ---------------------------------------------------------------------------------------
struct ConvPlan {
  // problem
  int B, IH, IW, OH, OW;    // source / output spatial size (source = before optional x2 upsample)
  int C0, C1;               // channels of source 0 / source 1 (virtual concat), Cin = C0 + C1
  int Cout;                 // real output channels
  int taps;                 // 1 or 9
  int stride, pad, up;      // out(o) reads in((o*stride + r - pad) >> up)
  // tiling (filled by conv_plan_tiles)
  int KC;                   // channels per LDS chunk (8 or 16)
  int NT;                   // 32-wide cout tiles per workgroup (1..3)
  int MT;                   // fp16 kernel: 32-pixel M tiles per workgroup (2/4); fp32 kernel: unused
  int KCS;                  // fp16 kernel, fp16 source: 16-channel sub-chunks per K stage (1 = chunk-wise)
  int LC;                   // fp16 kernel, fp16 source: 1 = loader/consumer persistent schedule (conv_f16_lc.h)
  int TH, TW;               // output tile, TH*TW <= 128
  int PH, PW;               // staged source patch
  int tiles_x, tiles_y;     // tiles over (B*OH virtual rows, OW)
  int n_groups;             // Cout tiles / NT
  int CoutPad;              // n_groups*NT*32
  size_t lds_bytes;
  int qnt;                  // quad schedule (conv_f16_q.hip): 16-cout tiles per N half; 0 = by Cout (3: Cout % 96 == 0, else 4); 1: 32-cout groups
};

struct ConvArgs {
  const float* src0;
  const float* src1;
  const float* wpack;       // [CoutPad/32][Cin/KC][taps][KC/8][64][4]
  const float* bias;        // [CoutPad] or null
  const float* temb;        // [B][temb_stride] (already offset to this layer's columns) or null
  const float* res;         // NHWC residual [B,OH,OW,Cout] or null
  const float* nscale;      // [B][Cin] GroupNorm scale (rstd*gamma) or null
  const float* nshift;      // [B][Cin]
  float* out;
  int temb_stride;
  int out_stride;           // channels of the destination tensor (NHWC) - ignored for nchw
  int out_coff;             // channel offset inside the destination tensor
  int out_nchw;             // 1: write [B,Cout,OH,OW]
  int act;                  // activation applied after the norm prologue
  float out_scale;          // multiplies the final value (1, or 1/sqrt(2) for skip_rescale)
  double* stats;            // fp16 3x3 kernel: per-tile GroupNorm partials [tile][Cout][2] (sum, sum of squares) or null
  long long* dbg;           // tuning builds only (CSD_C16_TIMING): per-workgroup phase timestamps, else null
};

int conv_plan_tiles(ConvPlan* p);                       // chooses KC/NT/tile, returns csd_status
size_t conv_packed_floats(const ConvPlan& p);           // floats of the packed weight (+ slack)
// pack one weight tensor: layout 0 = OIHW [Cout_src][Cin][kh][kw], 1 = NIN [Cin][Cout_src];
// cout_off places it inside a wider (concatenated) packed tensor.
int conv_pack_weight(const ConvPlan& p, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                     float* wpack, hipStream_t s);
int conv_launch(const ConvPlan& p, const ConvArgs& a, hipStream_t s);

// fp16-MFMA variant (conv_f16.hip): ns = 2 -> split hi/lo operands, 3 MFMAs per product (F16X3);
// ns = 1 -> plain fp16 operands (F16).  Covers 3x3 stride-1 (optionally nearest-x2-fused) layers whose
// Cin is a multiple of 16; everything else stays on the fp32 kernel.
bool conv16_supported(const ConvPlan& p);
int conv16_kcs(int ns, int cin);
int conv16_plan_tiles(ConvPlan* p, int ns, int kcs = 1, bool lc = false);
size_t conv16_packed_bytes(const ConvPlan& p, int ns);
int conv16_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src,
                       int cout_off, void* wpack, hipStream_t s);
int conv16_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s, bool in16 = false);
// "quad" schedule (conv_f16_q.hip): fp16-source 3x3 stride-1 layers with Cout % 96 == 0, four balanced waves
bool conv16q_supported(const ConvPlan& p, int ns);
size_t conv16q_packed_bytes(const ConvPlan& p, int ns);
int conv16q_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                        void* wpack, hipStream_t s);
int conv16q_plan_tiles(ConvPlan* p, int ns);
// phase-decomposed Upsample on the quad schedule (ConvPlan.up == 2; conv_f16_q.hip)
bool conv16q_up4_supported(const ConvPlan& p, int ns);
bool conv16q_up4_stats_ok(const ConvPlan& p);      // (after conv16q_plan_tiles) its epilogue statistics are a function of the sample alone
size_t conv16q_up4_packed_bytes(const ConvPlan& p, int ns);
int conv16q_pack_weight_up4(const ConvPlan& p, int ns, const float* w, void* wpack, hipStream_t s);
// raw: src0 is the fp32 NHWC source itself (ns = 2, a layer without a GroupNorm in front: the kernel's staging does the hi | lo split)
int conv16q_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s, bool raw = false);
// fused-prologue schedule (conv_ff.hip): 3x3 stride-1 layers on 16 x 16 tiles that lie inside one sample, fp32 NHWC sources
// (two-source virtual concat), GroupNorm affine + activation + fp16 split applied while staging (no gn_apply16 pass)
bool convff_supported(const ConvPlan& p, int ns);
size_t convff_packed_bytes(const ConvPlan& p, int ns);
int convff_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                       void* wpack, hipStream_t s);
int convff_plan_tiles(ConvPlan* p, int ns);
// the layers conv_ff runs as the software-pipelined stream of conv_xk.hip (fp32-class operands, an even number >= 4 of 16-channel stages)
bool convff_pipelined(const ConvPlan& p, int ns);
int convff_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s);
// pointwise (1x1) layers on the fp16 MFMA path (conv_pw16.hip): fp32 NHWC in / out, optional GroupNorm affine on the
// input, bias + residual epilogue; no LDS staging of activations - a pure HBM stream
bool pw16_supported(const ConvPlan& p, int ns);
size_t pw16_packed_bytes(const ConvPlan& p, int ns);
int pw16_pack_weight(const ConvPlan& p, int ns, const float* w, int layout, int cin_src, int cout_src, int cout_off,
                     void* wpack, hipStream_t s);
int pw16_launch(const ConvPlan& p, int ns, const ConvArgs& a, hipStream_t s);
// tap-partial form of a 3x3 convolution with <= 6 output channels: one pointwise contraction to 9*cout partial channels + a 9-tap gather
bool pw16_taps_supported(int cin, int cout, int ns);
int pw16_taps_cout(int cout);
int pw16_pack_weight_taps(const ConvPlan& p, int ns, const float* w, int cout, void* wpack, hipStream_t s);
int tapsum_launch(const float* part, const float* bias, const float* res, float* out, int B, int H, int W, int cout, int nchw,
                  float out_scale, hipStream_t s);
// stem.hip: cat(x, y [+ sigma z]) + 2v - 1 + NCHW -> NHWC + the first 3 x 3 conv (<= 8 -> Cout channels) + GroupNorm tile partials, one launch
bool stem_supported(int Cx, int Cy, int Cout, int S, int ns);
size_t stem_packed_bytes(int Cout, int ns);
int stem_pack_weight(const float* w, int Cin, int Cout, int ns, void* wpack, hipStream_t s);
int stem_tiles_per_image(int S);
int stem_launch(const float* x, const float* y, const float* y_noise, float y_sigma, const void* wpack, const float* bias, float* out,
                double* stats, int B, int Cx, int Cy, int Cout, int S, int centered, int ns, hipStream_t s);
// y16 = fp16 split of act(x*scale + shift): hi plane (and lo plane when ns == 2), NHWC halves [B*HW][C0+C1]
// statistics + finalize + apply + split of a small map in one launch (norm.hip); gn_fused16_groups() == 0: use the three-launch form
int gn_fused16_groups(int HW, int C0, int C1, int G);
int gn_fused16_launch(const float* src0, const float* src1, int C0, int C1, const float* gamma, const float* beta, float eps,
                      void* hi, void* lo, int B, int HW, int G, int act, hipStream_t s, int f8, float* nscale = nullptr,
                      float* nshift = nullptr);      // hi == nullptr: statistics only (scale / shift out)
int gn_apply16_launch(const float* src0, const float* src1, int C0, int C1, const float* nscale, const float* nshift,
                      void* hi, void* lo, int B, int HW, int act, hipStream_t s, int f8 = 0);
static inline int precision_ns(int precision) {   // CSD_PREC_* -> number of fp16 planes (0: fp32 kernel)
  return (precision == CSD_PREC_F16X3 || precision == CSD_PREC_F16F8) ? 2 : (precision == CSD_PREC_F16 ? 1 : 0);
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics -> per-(b,c) scale/shift consumed by the conv staging prologue (norm.hip)
// ---------------------------------------------------------------------------------------
struct GNPlan {
  int B, HW, C0, C1, G, nchunk;
};
size_t gn_partial_bytes(const GNPlan& p);
int gn_plan(GNPlan* p, int B, int HW, int C0, int C1, int G);
// per_channel: partial = [B * nchunk][C][2] (sum, sum of squares) per channel - the layout of the conv epilogues' tile partials
int gn_stats_launch(const GNPlan& p, const float* src0, const float* src1, double* partial,
                    hipStream_t s, int per_channel = 0);
// rs / ms (optional): the plain statistics (rstd, -mean * rstd) per (sample, channel) as well
int gn_finalize_launch(const GNPlan& p, const double* partial, const float* gamma, const float* beta,
                       float eps, float* nscale, float* nshift, hipStream_t s, float* rs = nullptr, float* ms = nullptr);
// finalize from the per-tile partials the conv epilogues wrote (ConvArgs::stats): source s has Cs channels and
// tpi_s tiles per sample, laid out [B*tpi_s][Cs][2]; p1 may be null (single source)
int gn_finalize_tiles_launch(const double* p0, int tpi0, int C0, const double* p1, int tpi1, int C1, int B, int HW, int G,
                             const float* gamma, const float* beta, float eps, float* nscale, float* nshift,
                             hipStream_t s);
// stand-alone apply: y = act(x*scale + shift), NHWC
// mask != null: y = dropout(act(x*scale + shift)) with the mask csd_dropout would draw for (seed, stream_id), written to mask
// hi (/ lo) != null: the fp16 hi (| lo) planes of y too ([B*HW][C] halves each), as gn_apply16 would split y
int gn_apply_launch(const float* x, const float* nscale, const float* nshift, float* y, int B, int HW,
                    int C, int act, hipStream_t s, float* mask = nullptr, float p_drop = 0.f, uint64_t seed = 0, uint64_t stream_id = 0,
                    void* hi = nullptr, void* lo = nullptr);
// two csd_sum_rows of one shape in one launch (backward.hip)
int sum_rows2(const float* x0, float* out0, const float* x1, float* out1, int R, int C, void* stream);
// csd_groupnorm_act_nhwc with the training forward's dropout fused into the apply pass (train_nhwc.hip)
int groupnorm_act_dropout_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* rs, float* ms, float* mask,
                               float p_drop, uint64_t seed, uint64_t stream_id, int B, int C, int HW, int groups, float eps, int act,
                               void* scratch, void* stream, void* planes = nullptr, int plane_count = 0);
// the operand planes conv2d_impl would split an NHWC source into (quad schedule): 0 = it would not (other kernel), else 1 or 2 planes of
// B*H*W*Cin halves each, hi first
int conv2d_operand_planes(int B, int Cin, int Cout, int H, int W, int ksize, int stride, int pad_mode, int up2, int precision);

// ---------------------------------------------------------------------------------------
// attention core on NHWC qkv (attention.hip): qkv [B, L, ld] with q at +0, k at +C, v at +2C
// ---------------------------------------------------------------------------------------
int attention_launch(const float* qkv, int ld, float* out, int B, int L, int C, hipStream_t s);
// the same core on the fp16 matrix cores: ns = 2 split operands (3 MFMAs per product, fp32-class), ns = 1 plain fp16
int attention16_launch(const float* qkv, int ld, float* out, int B, int L, int C, int ns, hipStream_t s);

// ---------------------------------------------------------------------------------------
// small kernels (elementwise.hip)
// ---------------------------------------------------------------------------------------
int timestep_embedding_launch(const float* t, float* out, int B, int dim, hipStream_t s);
// out[b][n] = g(sum_k f(in[b][k]) * W[n][k] + bias[n]);  f = act_in, g = act_out
int linear_launch(const float* in, const float* W, const float* bias, float* out, int B, int K, int N,
                  int act_in, hipStream_t s, int act_out = 0);
// NCHW x (+ y (+ sigma*noise)) -> NHWC [B,H,W,Cpad], v -> 2v-1 unless centered
int assemble_input_launch(const float* x, const float* y, const float* y_noise, float y_sigma,
                          float* out, int B, int Cx, int Cy, int HW, int Cpad, int centered,
                          hipStream_t s);
// scale multiplies the kernel (after the forward gain), flip reverses the taps: the transposed operators of the training graph
int fir_resample_nhwc_launch(const float* in, float* out, int B, int H, int W, int C, const float* taps4, int up, hipStream_t s,
                             float scale = 1.f, int flip = 0);
// out_x = FIR(in), out_h = FIR(act(in * nscale + nshift)) in one pass (the up / down BigGAN block's two resampled tensors)
int fir_resample2_nhwc_launch(const float* in, const float* nscale, const float* nshift, float* out_x, float* out_h, int B, int H, int W,
                              int C, const float* taps4, int up, int act, hipStream_t s);
int fourier_embedding_launch(const float* t, const float* W, float* out, int B, int E, hipStream_t s);
// per-operator convolution of the C ABI (ops_api.hip) with an optional NHWC residual added in the epilogue; GroupNorm backward with
// an optional addend (train_nhwc.hip): the fused forms the planned training graph (train_graph.h) uses
int conv2d_impl(const float* x, const float* weight, const float* bias, const float* res, const float* temb, float* y, int B, int Cin,
                int Cout, int H, int W, int ksize, int stride, int pad_mode, int up2, int precision, int layout, void* scratch, void* stream,
                const void* planes = nullptr);      // planes: the pre-split operand (conv2d_operand_planes() > 0), x is then not read
int groupnorm_act_backward_nhwc_add(const float* x, const float* gamma, const float* beta, const float* rs, const float* ms,
                                    const float* dy, const float* add, float* dx, float* dgamma_rows, float* dbeta_rows,
                                    int row_stride, int B, int C, int HW, int groups, int act, void* scratch, void* stream);
int axpby_launch(const float* a, const float* b, float* out, float alpha, float beta, float gamma, float post, size_t n,
                 hipStream_t s);
int bias_add_nchw_launch(const float* x, const float* bias, float* out, int B, int C, size_t inner, int bias_stride, int act,
                         hipStream_t s);
int nchw_to_nhwc_launch(const float* in, float* out, int B, int C, int HW, int Cw, int ld, hipStream_t s);
int nhwc_to_nchw_launch(const float* in, float* out, int B, int C, int HW, int Cstride, hipStream_t s);
int bridge_update_launch(const float* y0, float* state, const float* z, float w0, float w1, float sd, int has_prev, size_t n,
                         hipStream_t s);
int avgpool2_launch(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);
int nearest_up2_nhwc_launch(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);

// wgrad_bf16.hip: weight gradient of a stride-1 3x3 / 1x1 convolution with split-bf16 operands (NHWC x, dy; partial [S][Cout][Cin][taps])
int wgrad_bf16_launch(const float* xh, const float* dyh, float* partial, int B, int H, int W, int Cin, int Cout, int ksize, int S,
                      int per_split, hipStream_t s);

// sampler.hip
// `net` may be the x-part of a wider network output: sample b starts at net + b*net_stride
int sumsq_rows_launch(const float* net, int64_t net_stride, const float* z, double* partial, int B,
                      int64_t per, int nchunk, hipStream_t s);
int langevin_update_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z,
                           const double* partial, int nchunk, float std, float snr, float alpha, int B,
                           int64_t per, hipStream_t s, int* nonfinite = nullptr);
int finite_check_launch(const float* x, size_t total, int* nonfinite, hipStream_t s);
int norm_sums_launch(const double* partial, int nchunk, float std, int B, float* sums, hipStream_t s);
int langevin_update_global_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z,
                                  const float* sums, int Bg, float std, float snr, float alpha, int B, int64_t per,
                                  hipStream_t s, int* nonfinite = nullptr);
int affine_net_update_launch(float* x, float* x_mean, const float* net, int64_t net_stride, const float* z, float std, float p,
                             float a, float c, int B, int64_t per, hipStream_t s);
int reverse_diffusion_update_launch(float* x, float* x_mean, const float* net, int64_t net_stride,
                                    const float* z, float std, float G, int B, int64_t per, hipStream_t s);
int randn_launch(float* out, int64_t n, uint64_t seed, uint64_t stream_id, hipStream_t s);
int scale_rows_launch(float* out, const float* in, const float* scale, int divide, int B, int64_t per,
                      hipStream_t s);
int sumsq_nchunk(int64_t per);

}  // namespace csd
