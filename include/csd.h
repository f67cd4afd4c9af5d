/*
 * csd.h - C ABI of libcsd_hip.so: the MI355X (gfx950) score-network + predictor-corrector
 * sampler hot path of GBATZOLIS/conditional_score_diffusion.
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *   - every function returns 0 on success or a negative csd_status; csd_last_error() gives the
 *     text of the calling thread's last failure.  Nothing throws.
 *   - the CALLER owns every device buffer (inputs, outputs, packed weights, workspace); the
 *     library owns only opaque host-side handles.  All work is enqueued on the caller's
 *     hipStream_t (passed as void*) and never synchronises it, except where stated.
 *   - boundary tensors are the reference's: NCHW fp32 contiguous.  (Internally activations are
 *     NHWC fp32; that layout never leaks.)
 *   - one handle per (device, stream, host thread); handles are independent of each other.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository root).
 */
#ifndef CSD_H_
#define CSD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum csd_status {
  CSD_OK = 0,
  CSD_ERR_INVALID = -1,      /* bad argument / unsupported configuration            */
  CSD_ERR_HIP = -2,          /* a HIP runtime call or kernel launch failed           */
  CSD_ERR_STATE = -3,        /* call sequence violated (e.g. forward before pack)    */
  CSD_ERR_NOT_FOUND = -4,    /* unknown parameter name                               */
  CSD_ERR_WORKSPACE = -5,    /* caller-supplied buffer too small                     */
  CSD_ERR_NONFINITE = -6     /* the sampler's state left the finite range (csd_pc_sample / csd_pc_step_end of the last step:
                                an fp16-operand mode met an operand beyond 65504 - run the network in CSD_PREC_F32)          */
} csd_status;

/* activation ids (models/layers.py:29-41 get_act) */
enum { CSD_ACT_NONE = 0, CSD_ACT_SWISH = 1, CSD_ACT_RELU = 2, CSD_ACT_LRELU = 3, CSD_ACT_ELU = 4 };

/* arithmetic of the convolution / 1x1 contractions */
enum {
  CSD_PREC_F32 = 0,          /* fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 fmaf chain */
  CSD_PREC_F16X3 = 1,        /* split-fp16 (hi+lo) operands, 3 fp16 MFMAs, fp32 accumulate        */
  CSD_PREC_F16 = 2,          /* fp16 operands, fp32 accumulate                                      */
  CSD_PREC_F16F8 = 3         /* split operands; hi*hi on the fp16 MFMA, the two correction products K-concatenated on the fp8 MFMA
                                (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 operands, block scale 2^-11) in the fused-prologue block
                                convolution; every other layer as CSD_PREC_F16X3.  Measured network error, full-size SR3-160 against the
                                reference: 1.3e-5 norm-wise / 4.5e-5 element-wise per evaluation (fp16x3: 1.9e-6 / 5.8e-6; fp16: 9e-4 / 3.9e-3) */
};

const char* csd_version(void);
const char* csd_last_error(void);

/* In-library profiler: while enabled, every kernel launch of csd_unet_forward / csd_pc_sample is
 * bracketed by HIP events on the stream the kernel runs on.  csd_profile_stop synchronises the last event
 * and returns, per launch class, total milliseconds, launch count, the algorithmic flops (2 * MACs) and the bytes
 * THE KERNEL has to move (its operand planes in, fp32 out, a residual read where it has one); csd_profile_stop_ex
 * also returns the ALGORITHMIC bytes of SURVEY.md 8(d): input + output tensor of the layer, fp32, nothing else. */
enum {
  CSD_PROF_CONV3X3 = 0,           /* 3x3 stride-1 convolutions on the mode's dominant kernel (fp16x3: conv_xk_kernel;
                                     a mode without a fused-prologue kernel: every 3x3 stride-1 convolution)             */
  CSD_PROF_CONV3X3_RESAMPLE = 1,  /* stride-2 / nearest-x2-fused 3x3 convolutions            */
  CSD_PROF_CONV1X1 = 2,           /* NIN / 1x1 contractions                                   */
  CSD_PROF_GN_STATS = 3,
  CSD_PROF_GN_FINAL = 4,
  CSD_PROF_ATTENTION = 5,
  CSD_PROF_SAMPLER = 6,           /* predictor / corrector updates (+ norms)                  */
  CSD_PROF_OTHER = 7,             /* input assembly, time embedding, dense layers             */
  CSD_PROF_GN_APPLY = 8,          /* GroupNorm+activation+fp16 split pass (fp16 conv modes)   */
  CSD_PROF_CONV3X3_OTHER = 9,     /* 3x3 stride-1 convolutions NOT on the dominant kernel: the first layer, the <= 20^2 levels */
  CSD_PROF_NUM_CLASSES = 10
};
/* Restrict the events to the launch classes whose bit (1u << CSD_PROF_*) is set and, inside csd_pc_sample, to every
 * step_stride-th PC step (defaults: all classes, every step).  An event pair costs ~5-15 us of stream time (the records
 * serialise back-to-back launches): ~5 ms per PC step with everything on - a timed region that only needs the dominant
 * kernel selects that class on a sample of the steps. */
int csd_profile_select(unsigned class_mask, int step_stride);
int csd_profile_start(void);
int csd_profile_stop(int n_classes, double* ms, int64_t* launches, double* flops, double* bytes);
int csd_profile_stop_ex(int n_classes, double* ms, int64_t* launches, double* flops, double* bytes, double* alg_bytes);

/* ------------------------------------------------------------------------------------------
 * Score network (U-Net) - replaces models/ddpm.py:80-213 (DDPM), :275-298 (DDPM_paired_SR3,
 * DDPM_paired) behind models/utils.py:27-47,114-120 (registry / create_model).
 * ---------------------------------------------------------------------------------------- */
#define CSD_MAX_LEVELS 8
#define CSD_MAX_ATTN 8

typedef struct csd_unet_config {
  int32_t arch;                 /* 0 = DDPM family (models/ddpm.py); 1 = NCSN++ (models/ncsnpp.py), see below */
  int32_t nf;                   /* config.model.nf                                           */
  int32_t n_levels;             /* len(config.model.ch_mult)                                 */
  int32_t ch_mult[CSD_MAX_LEVELS];
  int32_t num_res_blocks;
  int32_t n_attn;               /* len(config.model.attn_resolutions)                        */
  int32_t attn_resolutions[CSD_MAX_ATTN];
  int32_t image_size;           /* config.data.effective_image_size                          */
  int32_t x_channels;           /* channels of x                                             */
  int32_t y_channels;           /* channels of the condition y (0: unconditional 'ddpm')     */
  int32_t out_channels;         /* config.model.output_channels                              */
  int32_t resamp_with_conv;
  int32_t conditional;          /* config.model.conditional (time embedding on/off)          */
  int32_t centered;             /* config.data.centered: 0 -> h = 2x-1 (models/ddpm.py:163-168)*/
  int32_t act;                  /* CSD_ACT_*                                                 */
  int32_t precision;            /* CSD_PREC_*                                                */
  /* ---- arch 1 (NCSN++, models/ncsnpp.py:44-236) only; resblock_type 'biggan', fir = True, combine 'sum' ---- */
  int32_t skip_rescale;         /* config.model.skip_rescale: (x + h)/sqrt(2)                */
  int32_t progressive;          /* 0 'none', 1 'output_skip'                                 */
  int32_t progressive_input;    /* 0 'none', 1 'input_skip'                                  */
  int32_t embedding_type;       /* 0 'positional', 1 'fourier' (W [nf] is parameter all_modules.0.W) */
  int32_t n_fir;                /* taps of config.model.fir_kernel (<= 8)                    */
  float fir_kernel[8];
} csd_unet_config;

typedef struct csd_unet csd_unet;

int csd_unet_create(const csd_unet_config* cfg, csd_unet** out);
void csd_unet_destroy(csd_unet* net);

/* Parameter table, in reference state_dict order/naming ("all_modules.3.Conv_0.weight", ...).
 * Layouts are the reference's: conv OIHW, NIN.W [in,out], Linear [out,in]. */
int csd_unet_num_params(const csd_unet* net);
int csd_unet_param_info(const csd_unet* net, int index, const char** name, int* ndim, int64_t shape[4]);
/* Register the device address of one parameter (fp32, contiguous, reference layout). */
int csd_unet_set_param(csd_unet* net, const char* name, const void* dev_ptr, int64_t numel);

/* Repack every registered parameter into the library's MFMA-fragment layout.  `packed` is a
 * caller-owned device buffer of csd_unet_packed_bytes(); must be re-run after weights change. */
size_t csd_unet_packed_bytes(const csd_unet* net);
int csd_unet_pack(csd_unet* net, void* packed, void* stream);

/* Activation workspace needed for batch size B. */
size_t csd_unet_workspace_bytes(csd_unet* net, int B);

/* One network evaluation = model(x | {'x','y'}, labels) of models/utils.py:134-150 in eval mode.
 *   x      [B, x_channels, S, S]   y [B, y_channels, S, S] (NULL iff y_channels == 0)
 *   labels [B]                     out [B, out_channels, S, S]
 *   y_noise/y_sigma: optional fused perturbation y_t = y + y_sigma * y_noise
 *                    (sampling/conditional.py:104-110); pass NULL / 0 to use y as is. */
int csd_unet_forward(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes,
                     const float* x, const float* y, const float* labels, float* out, int B,
                     const float* y_noise, float y_sigma, void* stream);

/* number of kernel launches / algorithmic flops+bytes of one forward at batch B (for bench) */
int csd_unet_stats(csd_unet* net, int B, int64_t* launches, double* flops, double* bytes);

/* ------------------------------------------------------------------------------------------
 * Fused predictor-corrector sampler - replaces the loop of sampling/conditional.py:180-226 and
 * sampling/unconditional.py:194-226 for the (reverse_diffusion, langevin) VE pair
 * (sampling/predictors.py:79-102, sampling/correctors.py:51-108, sde_lib.py:353-362,410-418).
 * Per-step scalars are computed by the host mirror in fp32 exactly as the reference does and
 * handed over as arrays of length n_steps.
 * ---------------------------------------------------------------------------------------- */
typedef struct csd_pc_params {
  int32_t n_steps;              /* p_steps                                                   */
  const float* labels;          /* [n_steps] network label for step i (t*(N-1) or sigma(t))  */
  const float* std_x;           /* [n_steps] sigma_x(t): score = net / std_x                 */
  const float* G;               /* [n_steps] reverse-diffusion G_i                            */
  const float* std_y;           /* [n_steps] sigma_y(t) or NULL (SR3 / unconditional)        */
  float snr;                    /* Langevin target snr                                       */
  int32_t denoise;              /* return x_mean of the last predictor step                  */
  /* noise source: a tape (parity mode) or on-device Philox (throughput mode) */
  const float* noise_tape;      /* draws in reference order, or NULL         */
  uint64_t seed;                /* Philox key when noise_tape == NULL                        */
  float* record;                /* optional [n_steps, B, C, S, S]: x after every step, or NULL*/
  /* the other registered update rules on the same device loop (zero-initialised = the pair above).  They are all affine in
   * (x, score, z) with per-step host scalars: x_mean = p*x + a*score, x = x_mean + b*z  (sampling/predictors.py:52-76 Euler-
   * Maruyama, :105-135 ancestral sampling; sampling/correctors.py:111-142 annealed Langevin dynamics). */
  int32_t predictor;            /* 0 reverse diffusion (G); 1 affine table pred_coef; 2 none (no evaluation, no draw) */
  int32_t corrector;            /* 0 Langevin (snr, batch-mean norms); 1 affine table corr_coef; 2 none               */
  const float* pred_coef;       /* [n_steps][3] = (p, a, b) when predictor == 1                                      */
  const float* corr_coef;       /* [n_steps][3] when corrector == 1                                                  */
  /* `use_path` of the two-SDE samplers (sampling/conditional.py:85-100,124-178; sde_lib.py:323-339): y_t is not redrawn from the
   * marginal for every evaluation but follows the bridge p(y_t | y_0, y_{t+tau}): once per step y_t = w0*y + w1*y_{t+tau} + s*z,
   * the PREDICTOR runs first, then the corrector, both on that y_t.  path_coef != NULL selects it (std_y must be NULL);
   * y_{T+tau} = y + path_std0 * z.  Draw order: prior | z_y0 | per step: z_y, z_predictor, z_corrector (existing phases only). */
  const float* path_coef;       /* [n_steps][3] = (w0, w1, s), or NULL                                                */
  float path_std0;              /* sigma_y(T + tau)                                                                    */
  /* Langevin corrector of the VP / subVP SDEs (sampling/correctors.py:63-65,94-96): step size times alphas[timestep_i]     */
  const float* corr_alpha;      /* [n_steps] or NULL (= 1: the VE SDEs)                                                */
} csd_pc_params;

/* x: [B, x_channels, S, S] in: prior sample (already scaled by sigma_max); out: result.
 * y: [B, y_channels, S, S] or NULL.  scratch: csd_pc_scratch_bytes() device bytes.
 * Finiteness contract: the Langevin corrector's norms (which see every element of the score and of the noise) and one pass over the
 * returned state set a device flag; csd_pc_sample reads it back behind the stream before it returns - its ONE synchronisation - and
 * fails with CSD_ERR_NONFINITE rather than hand back NaN images (csd_pc_step_end does the same after the last step). */
size_t csd_pc_scratch_bytes(const csd_unet* net, int B);
int csd_pc_sample(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes,
                  void* scratch, size_t scratch_bytes, float* x, const float* y, int B,
                  const csd_pc_params* p, void* stream);
/* The same loop one PC step at a time, for the GLOBAL-NORM exactness mode of batch-sharded sampling (identical to
 * ONE reference process holding the global batch; the Langevin step size uses batch-mean norms, sampling/correctors.py:100-106).
 *   csd_pc_step_begin(step): corrector network evaluation + its noise draw; norm_sums[0] = sum_b ||score_b||_2 and
 *                            norm_sums[1] = sum_b ||z_b||_2 over THIS rank's B samples (two fp32 on the device).
 *   -- the caller all-reduces (SUM) the two floats over the ranks: one 8-byte RCCL collective, no host synchronisation --
 *   csd_pc_step_end(step):   Langevin update with gbar = norm_sums[0] / global_batch, nbar = norm_sums[1] / global_batch,
 *                            then the predictor half of the step; after the last step x holds x_mean when p->denoise.
 * Same params / scratch / noise order as csd_pc_sample (with global_batch == B and no all-reduce the two calls per step
 * reproduce it); p->record is honoured. */
int csd_pc_step_begin(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes, void* scratch,
                      size_t scratch_bytes, float* x, const float* y, int B, const csd_pc_params* p, int step,
                      float* norm_sums, void* stream);
int csd_pc_step_end(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes, void* scratch,
                    size_t scratch_bytes, float* x, const float* y, int B, const csd_pc_params* p, int step,
                    const float* norm_sums, int global_batch, void* stream);

/* Stand-alone update kernels (the "noise-add" steps), usable with any score source:
 *   csd_langevin_step: sampling/correctors.py:51-78,88-108: step = (snr * mean||z|| / mean||score||)^2 * 2 * alpha with
 *                      alpha = sde.alphas[timestep] for the VP / subVP SDEs (:63-65,94-96) and 1 for the VE SDEs
 *   csd_reverse_diffusion_step: sampling/predictors.py:84-89,97-102 with f = 0
 * net: raw network output; score = net / std.  x is updated in place, x_mean written.
 * scratch: >= csd_update_scratch_bytes(B) bytes. */
size_t csd_update_scratch_bytes(int B);
int csd_langevin_step(float* x, float* x_mean, const float* net, const float* z, float std,
                      float snr, float alpha, int B, int64_t per_sample, void* scratch, void* stream);
int csd_reverse_diffusion_step(float* x, float* x_mean, const float* net, const float* z, float std,
                               float G, int B, int64_t per_sample, void* stream);
/* General one-step update  x_mean = p*x + a*score,  x = x_mean + c*z  (n elements, scalars per call): the
 * Euler-Maruyama and ancestral-sampling predictors (sampling/predictors.py:52-76,105-179) and the annealed
 * Langevin corrector (sampling/correctors.py:111-142); `score` is the score itself (already divided by std). */
int csd_affine_noise_step(float* x, float* x_mean, const float* score, const float* z, float p, float a,
                          float c, int64_t n, void* stream);
/* out[b] = ||a_b||_2 over per_sample elements (fp64 accumulation): the per-sample norms behind the Langevin step size
 * (sampling/correctors.py:102-103), for the global-batch exactness mode that all-reduces them across ranks */
int csd_row_norms(const float* a, float* out, int B, int64_t per_sample, void* stream);
/* standard-normal fill (Philox4x32-10 + Box-Muller); counter-based: (seed, stream_id) */
int csd_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);
/* out[b,:] = in[b,:] * scale[b]  or / scale[b] (divide_by_sigmas, models/utils.py:50-74) */
int csd_scale_rows(float* out, const float* in, const float* scale, int divide, int B,
                   int64_t per_sample, void* stream);

/* ------------------------------------------------------------------------------------------
 * Individual operators (NCHW fp32 boundary) - the kernels the network is built from, exported
 * for per-op parity tests and for callers that use the reference's functional ops directly.
 * ---------------------------------------------------------------------------------------- */
/* y = act(GroupNorm(x; G groups, eps) * gamma + beta)  - nn.GroupNorm + get_act
 * (models/layers.py:571,638,646).  scratch: csd_groupnorm_scratch_bytes(B, C, H, W). */
size_t csd_groupnorm_scratch_bytes(int B, int C, int H, int W);
int csd_groupnorm_act(const float* x, const float* gamma, const float* beta, float* y, int B, int C,
                      int H, int W, int groups, float eps, int act, void* scratch, void* stream);

/* 3x3 / 1x1 convolution (models/layers.py:100-132 ddpm_conv1x1/ddpm_conv3x3; NIN :555-564 via
 * ksize=1).  weight OIHW, stride in {1,2}; pad_mode 0: symmetric pad (ksize/2); 1: the
 * reference Downsample's (0,1,0,1) pad + stride 2 (models/layers.py:619-625); up2: nearest x2
 * upsample of x first (models/layers.py:600-604).  scratch: csd_conv_scratch_bytes(). */
size_t csd_conv_scratch_bytes(int B, int Cin, int Cout, int H, int W, int ksize, int up2);
int csd_conv2d(const float* x, const float* weight, const float* bias, float* y, int B, int Cin,
               int Cout, int H, int W, int ksize, int stride, int pad_mode, int up2, int precision,
               void* scratch, void* stream);

/* single-head self-attention core of AttnBlock (models/layers.py:584-588):
 * q,k,v,out [B, C, H, W]; w = softmax(q.k * C^-1/2) over keys; out = w.v */
size_t csd_attention_scratch_bytes(int B, int C, int H, int W);
int csd_attention(const float* q, const float* k, const float* v, float* out, int B, int C, int H,
                  int W, void* scratch, void* stream);

/* upfirdn2d (op/upfirdn2d.py:147-158, op/upfirdn2d_kernel.cu:107-207): x [N, C, H, W],
 * kernel [kh, kw]; out [N, C, (H*up+pad0+pad1-kh)/down+1, ...]. */
int csd_upfirdn2d(const float* x, const float* kernel, float* out, int N, int C, int H, int W, int kh,
                  int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                  int pad_y1, void* stream);
/* fused_bias_act (op/fused_bias_act_kernel.cu:18-49): out = lrelu(x + b[c], alpha) * scale,
 * act: 1 linear, 3 lrelu; grad 0 forward, 1 backward w.r.t. x using `ref` = forward output. */
int csd_fused_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t numel,
                       int C, int64_t inner, int act, int grad, float alpha, float scale, void* stream);
/* nearest-neighbour x2 upsample (F.interpolate in models/layers.py:601) */
int csd_nearest_up2(const float* x, float* out, int N, int C, int H, int W, void* stream);
/* sinusoidal timestep embedding (models/layers.py:524-538): out [B, dim] */
int csd_timestep_embedding(const float* t, float* out, int B, int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small operators for graphs orchestrated above the C ABI (the NCSN++ adapter, models/ncsnpp.py:238-388):
 * ---------------------------------------------------------------------------------------- */
/* nn.Linear on the activated input: out[b] = act_in(in[b]) . weight^T + bias; weight [N, K] (temb MLP
 * models/ncsnpp.py:96-102,257-263; Dense_0 models/layerspp.py:253-255) */
int csd_linear(const float* in, const float* weight, const float* bias, float* out, int B, int K, int N,
               int act_in, void* stream);
/* GaussianFourierProjection (models/layerspp.py:32-41): out [B, 2E] = [sin(2 pi W t), cos(2 pi W t)] */
int csd_fourier_embedding(const float* t, const float* W, float* out, int B, int E, void* stream);
/* out = (alpha*a + beta*b + gamma) * post, b may be NULL: x + h, (x + h)/sqrt(2) (layerspp.py:271-274),
 * Combine 'sum' (:56-57), 2x - 1 (ncsnpp.py:266-268), pyramid sums (:352) */
int csd_axpby(const float* a, const float* b, float* out, float alpha, float beta, float gamma, float post,
              int64_t n, void* stream);
/* out[b,c,:] = act(x[b,c,:] + bias[b*bias_stride + c]): h += Dense_0(act(temb))[:, :, None, None] */
int csd_bias_add_nchw(const float* x, const float* bias, float* out, int B, int C, int64_t inner,
                      int bias_stride, int act, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gradient operators (csrc/backward.hip) - the autograd backward the reference gets from torch for its layers
 * (the training step of run_lib.py:55-73 / losses.py:99-232).  NCHW fp32 like the forward
 * operators.  The DATA gradient of a convolution is csd_conv2d on dy with the flipped, transposed weight (stride 2:
 * dy zero-inserted with csd_upfirdn2d; nearest-x2: a 2x2 sum of the result), so only the weight gradient is new.
 * ---------------------------------------------------------------------------------------- */
/* dw[Cout, Cin, k, k] = sum over batch and pixels of dy (x) x for the convolution csd_conv2d(x; ksize, stride,
 * pad_mode, up2) with x [B, Cin, H, W], dy [B, Cout, OH, OW].  fp32 MFMA, deterministic split-K reduction. */
size_t csd_conv_wgrad_scratch_bytes(int B, int Cin, int Cout, int H, int W, int ksize, int stride, int up2);
int csd_conv2d_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W, int ksize,
                     int stride, int pad_mode, int up2, void* scratch, void* stream);
/* backward of csd_groupnorm_act: dx [B,C,H,W]; dgamma_rows / dbeta_rows [B, C] hold the per-sample sums
 * (dgamma = csd_sum_rows of them) */
int csd_groupnorm_act_backward(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                               float* dgamma_rows, float* dbeta_rows, int B, int C, int H, int W, int groups, float eps,
                               int act, void* stream);
/* backward of csd_attention: dq, dk, dv from dout; scratch csd_attention_backward_scratch_bytes() */
size_t csd_attention_backward_scratch_bytes(int B, int C, int H, int W);
int csd_attention_backward(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                           float* dv, int B, int C, int H, int W, void* scratch, void* stream);
/* strided batched fp32 GEMM  C[z][m*scm + n*scn] = alpha * sum_k A[z][m*sam + k*sak] * B[z][k*sbk + n*sbn]
 * (z-strides za, zb, zc): the Linear / NIN-free contractions of the backward (dIn = dOut.W, dW = dOut^T.act(in)) */
int csd_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk,
              int64_t sbn, int64_t scm, int64_t scn, int batch, int64_t za, int64_t zb, int64_t zc, float alpha,
              void* stream);
/* out[r] = sum of row r ([rows, inner], fp64 accumulation): per-(sample, channel) sums of dy = gradient of a
 * broadcast bias / time-embedding add;  out[c] = sum_r x[r][c]: the batch reduction that follows */
int csd_sum_inner(const float* x, float* out, int64_t rows, int64_t inner, void* stream);
int csd_sum_rows(const float* x, float* out, int R, int C, void* stream);
/* dy == NULL: out = act(x); else out = dy * act'(x) */
int csd_act(const float* x, const float* dy, float* out, int act, int64_t n, void* stream);
int csd_mul(const float* a, const float* b, float* out, int64_t n, void* stream);
/* nn.Dropout(p) in training mode (models/layers.py:647,662): mask = (u >= p)/(1-p) from Philox4x32-10 keyed by
 * (seed, stream_id), out = x * mask; the backward is csd_mul(dy, mask) */
int csd_dropout(const float* x, float* out, float* mask, float p, uint64_t seed, uint64_t stream_id, int64_t n,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * Parameter update of a training step on flat fp32 buffers (csrc/optim.hip): global-norm clipping
 * (torch.nn.utils.clip_grad_norm_, losses.py:48-49; grad_norm = device scalar ||grad||_2 from csd_global_norm, NULL
 * or max_norm < 0: no clipping) + torch.optim.Adam (losses.py:12-23) + the EMA of models/ema.py:61-90 (ema may be
 * NULL) in one pass.  `step` is the 1-based update count (bias correction); lr already carries the warm-up factor.
 * ---------------------------------------------------------------------------------------- */
int csd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                  const float* grad_norm, int64_t n, int step, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float max_norm, float ema_decay, void* stream);
/* out[0] = ||a||_2 of one long vector (fp64 partials of 1024 workgroups added in index order): the total gradient norm */
size_t csd_global_norm_scratch_bytes(void);
int csd_global_norm(const float* a, float* out, int64_t n, void* scratch, void* stream);
/* ema -= (1 - decay) * (ema - param)  (models/ema.py:85-89) */
int csd_ema_update(float* ema, const float* param, int64_t n, float decay, void* stream);

/* ------------------------------------------------------------------------------------------
 * NHWC forms of the training operators (csrc/train_nhwc.hip): the differentiable DDPM-family graph keeps activations in
 * the library's internal layout [B, H, W, C], so no layer pays an NCHW<->NHWC change.  Same kernels as above.
 * ---------------------------------------------------------------------------------------- */
/* The ResnetBlock convolution with its prologue fused (csrc/conv_ff.hip; reference models/layers.py:632-675: Conv(act(GroupNorm(x)))
 * [+ Dense(temb)] [+ x]): 3x3, stride 1, pad 1 on NHWC fp32 tensors.  x0 [B,H,W,C0] (+ x1 [B,H,W,C1] or NULL: virtual concat),
 * H % 16 == 0, W % 16 == 0 (CSD_PREC_F16X3 layers of the Winograd form - an even number >= 4 of 16-channel stages - also H % 8 == 0, W % 8 == 0
 * with tiles at least 65 % full, e.g. 40 x 40: ragged 16 x 16 tiles, csrc/conv_xk.hip), C0 / C1 multiples of 32, Cout a multiple of 96 (three 32-cout tiles per workgroup: the nf = 96 nets) or of
 * 64 (two: the nf = 128 nets); weight OIHW [Cout, C0+C1, 3, 3]; nscale / nshift
 * [B, C0+C1] = the GroupNorm's per-(sample, channel) rstd*gamma and beta - mean*rstd*gamma, applied as SiLU(x*scale + shift)
 * while the operand is staged (both NULL: the convolution reads x as it is); temb [B, temb_stride] or NULL (column c of sample b
 * is added to cout c), res [B,H,W,Cout] or NULL (added), out_scale multiplies the result.  stats (or NULL):
 * [B*ceil(H/16)*ceil(W/16)][Cout][2] doubles = per-tile (sum, sum of squares) of the written tensor (valid pixels only).  precision: CSD_PREC_F16X3, CSD_PREC_F16F8 (the
 * split operands with the two correction products on the fp8 matrix cores; 1.3e-5 / 4.5e-5 network error, see csd_precision) or CSD_PREC_F16.  scratch holds the packed weight: csd_conv3x3_block_scratch_bytes(C0 + C1, Cout). */
size_t csd_conv3x3_block_scratch_bytes(int Cin, int Cout);
int csd_conv3x3_block(const float* x0, const float* x1, const float* weight, const float* bias, const float* nscale,
                      const float* nshift, const float* temb, int temb_stride, const float* res, float out_scale, float* y,
                      double* stats, int B, int C0, int C1, int Cout, int H, int W, int precision, void* scratch, void* stream);

/* csd_conv2d / csd_conv2d_wgrad with layout flags: bit 0 = first tensor operand (x) is NHWC, bit 1 = second (y resp. dy) is
 * NHWC.  An NHWC x needs Cin % 8 == 0.  csd_conv2d_ex bit 2: `weight` is the OIHW weight [Cin, Cout, k, k] of the TRANSPOSED
 * convolution and is applied transposed + spatially flipped (the data gradient, without materialising that weight).
 * csd_conv2d_wgrad_ex bit 2: the stride-1 weight gradient may run with split-bf16
 * operands on the bf16 matrix cores (hi*hi + hi*lo + lo*hi, ~2^-16 relative error per product) instead of exact fp32 MFMA. */
int csd_conv2d_ex(const float* x, const float* weight, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                  int ksize, int stride, int pad_mode, int up2, int precision, int layout, void* scratch, void* stream);
int csd_conv2d_wgrad_ex(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W, int ksize,
                        int stride, int pad_mode, int up2, int layout, void* scratch, void* stream);
/* GroupNorm(+act) on [B, HW, C]; rs / ms [B, C] receive (rstd, -mean*rstd) per channel for the backward */
size_t csd_groupnorm_nhwc_scratch_bytes(int B, int C, int HW);
int csd_groupnorm_act_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* rs, float* ms, int B,
                           int C, int HW, int groups, float eps, int act, void* scratch, void* stream);
int csd_groupnorm_act_backward_nhwc(const float* x, const float* gamma, const float* beta, const float* rs, const float* ms,
                                    const float* dy, float* dx, float* dgamma_rows, float* dbeta_rows, int row_stride, int B, int C,
                                    int HW, int groups, int act, void* scratch, void* stream);
/* out[b,p,c] = x[b,p,c] + bias[b,c];  csd_sum_pixels_nhwc: out[b,c] = sum_p x[b,p,c] (the gradient of that add; a conv bias gradient after
 * csd_sum_rows).  csd_groupnorm_act_backward_nhwc writes row b of dgamma / dbeta at b*row_stride, so both
 * can live in one [B, 2C] buffer that a single csd_sum_rows reduces. */
int csd_bias_add_nhwc(const float* x, const float* bias, float* out, int B, int HW, int C, void* stream);
size_t csd_sum_pixels_scratch_bytes(int B, int HW, int C);
int csd_sum_pixels_nhwc(const float* x, float* out, int B, int HW, int C, void* scratch, void* stream);
/* data-gradient helpers of the resampling convolutions: dy on the odd positions of a 2h x 2w grid; 2x2 block sums */
int csd_zero_insert_odd_nhwc(const float* dy, float* z, int B, int h, int w, int C, void* stream);
int csd_sumpool2_nhwc(const float* in, float* out, int B, int h, int w, int C, void* stream);
/* attention core and its backward on the packed qkv tensor [B, L, 3C] (q | k | v per pixel); out, dout [B, L, C];
 * scratch of the backward: csd_attention_backward_scratch_bytes(B, C, L, 1) */
int csd_attention_nhwc(const float* qkv, float* out, int B, int L, int C, void* stream);
/* the same forward core in the arithmetic csd_unet_forward uses in precision mode `precision` (CSD_PREC_*): F32 = the call above;
 * F16X3 / F16F8 = operands split hi + lo on the fp16 matrix cores (fp32-class); F16 = plain fp16 operands */
int csd_attention_nhwc_prec(const float* qkv, float* out, int B, int L, int C, int precision, void* stream);
int csd_attention_backward_nhwc(const float* qkv, const float* dout, float* dqkv, int B, int L, int C, void* scratch,
                                void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step of the DDPM-family network as ONE planned graph (replaces torch autograd over
 * models/ddpm.py:149-213 + models/layers.py:524-675 in `model.train()` mode, run_lib.py:55-73).
 *   csd_unet_train_forward: the forward with nn.Dropout(dropout_p) active (mask = Philox keyed by (dropout_seed,
 *     (call_index << 16) + running dropout index), models/layers.py:647,662); every tensor a gradient needs stays in `workspace`.
 *   csd_unet_backward: given d loss / d out ([B, out_channels, S, S]), writes d loss / d parameter i to grads[i] (overwrite).
 * params[i] / grads[i]: device pointers of parameter i in csd_unet_param_info order and layout (fp32, 16-byte aligned).
 * One backward per forward, same handle / workspace / B / call_index; the workspace must not be touched in between: a second
 * forward into the same workspace before the backward makes that backward fail with CSD_ERR_STATE (it names the first forward's
 * call_index) - two forwards of one network may be alive at once when each has its own workspace.
 * csd_unet_train_workspace_bytes depends on B and on whether dropout_p > 0.  Both architectures: arch 0 (models/ddpm.py:149-213) and
 * arch 1 (NCSN++, models/ncsnpp.py:238-388: BigGAN blocks with FIR up / down sampling, Combine 'sum', input / output pyramids).
 * ---------------------------------------------------------------------------------------- */
size_t csd_unet_train_workspace_bytes(csd_unet* net, int B, float dropout_p);
int csd_unet_train_forward(csd_unet* net, const float* const* params, void* workspace, size_t workspace_bytes, const float* x,
                           const float* y, const float* labels, float* out, int B, float dropout_p, uint64_t dropout_seed,
                           uint64_t call_index, void* stream);
int csd_unet_backward(csd_unet* net, const float* const* params, float* const* grads, void* workspace, size_t workspace_bytes,
                      const float* d_out, int B, uint64_t call_index, void* stream);
/* The library keeps one recorded training graph per (handle, workspace).  A caller that frees a workspace (a monitoring forward's
 * private one) tells the library so: the record is dropped (no error if there is none).  csd_unet_destroy drops all of a handle's. */
int csd_unet_train_release(csd_unet* net, const void* workspace);
/* The same for the record of ONE forward: dropped only if it still is the record of csd_unet_train_forward call `call_index` (a late
 * release must not erase the record of a newer forward that was given the same workspace address). */
int csd_unet_train_release_call(csd_unet* net, const void* workspace, uint64_t call_index);
/* Gradient-ready marks - what Lightning-DDP's bucketed all-reduce hooks into autograd for (run_lib.py:55-73), for a backward that is
 * ONE call: while csd_unet_backward enqueues its kernels on `stream`, it records events[k] on that stream as soon as every gradient
 * of the modules with all_modules index >= first_module[k] is final (the backward walks all_modules back to front; the embedding
 * MLP's gradients come last).  A data-parallel caller makes its communication stream wait for events[k] and launches the
 * all-reduce of that bucket there: the reduction of the late layers' gradients overlaps the backward of the early ones.
 * The marks stay registered on the handle until replaced or cleared (n = 0); the events are the caller's.
 * csd_unet_backward_marks_epoch: number of csd_unet_backward calls of this handle that recorded every mark (a caller checks that it
 * advanced before trusting the events). */
int csd_unet_backward_marks(csd_unet* net, const int* first_module, void* const* events, int n);
uint64_t csd_unet_backward_marks_epoch(csd_unet* net);
void* csd_event_create(void);                         /* hipEventDisableTiming; NULL on failure */
int csd_event_destroy(void* event);
int csd_stream_wait_event(void* stream, void* event);
int csd_event_query(void* event);                     /* 1: complete, 0: not yet, < 0: error */

#ifdef __cplusplus
}
#endif
#endif /* CSD_H_ */
