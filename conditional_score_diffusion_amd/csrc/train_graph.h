// train_graph.h - the training step of BOTH network families (arch 0: DDPM family, arch 1: NCSN++) as ONE planned graph behind the C
// ABI: csd_unet_train_forward / csd_unet_backward.  Included at the end of unet.hip (shares Net /
// Module / Param).  NCSN++ (models/ncsnpp.py:238-388, models/layerspp.py:44-91,212-274): BigGAN blocks with FIR up / down sampling of
// both branches (backward = the transposed FIR: the other direction's geometry with the flipped taps, op/upfirdn2d.py:20-87), the
// Conv_2 shortcut, (x + h) / sqrt(2), AttnBlockpp, Combine 'sum' of the input pyramid, the output pyramid, Fourier or positional
// embedding, GroupNorm(min(C / 4, 32)).
//
// What it replaces: torch autograd over models/ddpm.py:149-213 + models/layers.py:524-675 in training mode (run_lib.py:55-73).
// The forward runs the reference's layer sequence on the NHWC operators of this library with nn.Dropout active and keeps
// exactly the tensors a gradient needs in the caller's workspace; the backward walks the recorded steps in reverse and
// writes every parameter gradient straight to the caller's pointers (one per parameter, csd_unet_param_info order) - no
// autograd graph, no per-layer host round trip, no transposed / concatenated weight copies (NIN weights are applied through
// the transposed-weight flag of csd_conv2d_ex, their gradients come from the weight-gradient kernel with swapped operands),
// Conv_0's bias gradient and the time-embedding gradient of a block share one reduction.
//
// Memory: [saved activations (forward, bump)] [gradients + temporaries (backward, bump with per-module release)]; the size
// is found by a dry run of the same code (csd_unet_train_workspace_bytes).
#pragma once

namespace csd {

struct TT { float* p = nullptr; int H = 0, C = 0; float* g = nullptr; };          // tensor [B, H, H, C]: data, gradient
enum TSKind { TS_STEM, TS_RES, TS_ATTN, TS_DOWN, TS_UP, TS_CAT, TS_HEAD, TS_COMBINE, TS_PYR };
struct TStep {
  TSKind kind;
  int mod = -1, in0 = -1, in1 = -1, out = -1;
  float* sv[10] = {};
  uint64_t drop_id = 0;
  int flag = 0;                 // TS_PYR: 1 = the pyramid level above adds FIR-up of the previous level (in1)
};

struct TrainState {
  bool valid = false;
  int B = 0;
  void* ws = nullptr;
  size_t fwd_top = 0;
  float p_drop = 0.f;
  uint64_t call = 0;            // the forward's call_index: csd_unet_backward must name it
  std::vector<TT> t;
  std::vector<TStep> steps;
  float *xin = nullptr, *emb = nullptr, *temb1 = nullptr, *temb2 = nullptr, *temb2_act = nullptr;
  int lin0 = 0, emb_dim = 0, fourier_mod = -1;      // module index of the embedding MLP's first Linear; width of the embedding; NCSN++ Fourier module
};

// one recorded forward per (network handle, workspace): two forwards of one network may be alive at once (a monitoring forward
// between a training forward and its backward) as long as each has its own workspace - and a backward names the call_index of the
// forward it belongs to, so a workspace that was re-used in between is an error, never a silently wrong gradient
static std::map<std::pair<const Net*, const void*>, TrainState> g_train;
// gradient-ready marks of a handle (csd_unet_backward_marks): event k is recorded on the backward's stream as soon as every gradient
// of the modules with all_modules index >= first_module[k] is final; `epoch` counts the backward calls that recorded all of them
struct GradMarks {
  std::vector<std::pair<int, hipEvent_t>> marks;     // sorted by first_module, descending (the order they become ready)
  uint64_t epoch = 0;
};
static std::map<const Net*, GradMarks> g_marks;
static std::mutex g_train_mu;                         // (the map only; a handle itself is driven by one thread at a time)
static TrainState& train_state_of(const Net* n, const void* ws) {
  std::lock_guard<std::mutex> lk(g_train_mu);
  return g_train[std::make_pair(n, ws)];
}
static TrainState* train_state_find(const Net* n, const void* ws) {
  std::lock_guard<std::mutex> lk(g_train_mu);
  auto it = g_train.find(std::make_pair(n, ws));
  return it == g_train.end() ? nullptr : &it->second;
}
static void train_state_erase_one(const Net* n, const void* ws, const uint64_t* call = nullptr) {
  std::lock_guard<std::mutex> lk(g_train_mu);
  auto it = g_train.find(std::make_pair(n, ws));
  // (a release that names its forward's call index drops the record of THAT forward only: a finalizer that runs late - its context sat
  // in a reference cycle - may find the allocator has handed the same address to a newer forward whose backward is still pending)
  if (it != g_train.end() && (!call || it->second.call == *call)) g_train.erase(it);
}
// records whose forward has been consumed (or failed) and whose workspace is not `keep`: a long run with monitoring forwards in
// private workspaces must not grow the map without bound (each record holds pointers into a workspace that may be gone)
static void train_state_purge_stale(const Net* n, const void* keep) {
  std::lock_guard<std::mutex> lk(g_train_mu);
  for (auto it = g_train.begin(); it != g_train.end();)
    it = (it->first.first == n && it->first.second != keep && !it->second.valid) ? g_train.erase(it) : std::next(it);
}
static void train_state_erase(const Net* n) {
  std::lock_guard<std::mutex> lk(g_train_mu);
  for (auto it = g_train.begin(); it != g_train.end();) it = it->first.first == n ? g_train.erase(it) : std::next(it);
  g_marks.erase(n);
}

__global__ void concat_c_kernel(const float4* __restrict__ a, int ca4, const float4* __restrict__ b, int cb4,
                                float4* __restrict__ out, size_t npix) {
  const int c4 = ca4 + cb4;
  const size_t n = npix * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i / c4;
    const int c = (int)(i - px * c4);
    out[i] = c < ca4 ? a[px * ca4 + c] : b[px * cb4 + (c - ca4)];
  }
}

__global__ void split_c_kernel(const float4* __restrict__ in, int ca4, int cb4, float4* __restrict__ a,
                               float4* __restrict__ b, size_t npix) {
  const int c4 = ca4 + cb4;
  const size_t n = npix * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t px = i / c4;
    const int c = (int)(i - px * c4);
    const float4 v = in[i];
    if (c < ca4) a[px * ca4 + c] = v; else b[px * cb4 + (c - ca4)] = v;
  }
}

// rows of `width` floats: dst[r*dpitch + j] = src[r*spitch + j]   (q|k|v weight concatenation and its inverse)
__global__ void copy2d_kernel(const float* __restrict__ src, int spitch, float* __restrict__ dst, int dpitch, int width, int rows) {
  const size_t n = (size_t)width * rows;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / width;
    const int j = (int)(i - r * width);
    dst[r * dpitch + j] = src[r * spitch + j];
  }
}

struct TG {
  Net& n;
  TrainState& st;
  int B;
  hipStream_t s;
  bool dry;
  const float* const* P;
  float* const* G;
  float* base;
  size_t top = 0, peak = 0;
  int prec, act;
  float p_drop = 0.f;
  uint64_t seed = 0, call = 0;
  int drop_count = 0;
  GradMarks* gm = nullptr;      // backward only
  size_t gm_next = 0;
  // every gradient of the modules with index >= `from` has been enqueued: record the marks that this completes
  int marks_reached(int from) {
    if (!gm || dry) return CSD_OK;
    while (gm_next < gm->marks.size() && gm->marks[gm_next].first >= from) {
      CSD_CHECK_HIP(hipEventRecord(gm->marks[gm_next].second, s));
      ++gm_next;
    }
    return CSD_OK;
  }

  TG(Net& net, TrainState& state, int B_, hipStream_t s_, bool dry_, const float* const* P_, float* const* G_, float* ws)
      : n(net), st(state), B(B_), s(s_), dry(dry_), P(P_), G(G_), base(ws), prec(net.cfg.precision), act(net.cfg.act) {}

  float* alloc(size_t nfl) {
    nfl = (nfl + 63) / 64 * 64;
    float* p = base + top;
    top += nfl;
    peak = std::max(peak, top);
    return p;
  }
  float* alloc_bytes(size_t bytes) { return alloc((bytes + 3) / 4); }
  int G_of(int C) const { return n.cfg.arch == 1 ? std::min(C / 4, 32) : 32; }      // GroupNorm groups (layerspp.py:67,219,231)
  float skip_scale() const { return (n.cfg.arch == 1 && n.cfg.skip_rescale) ? 0.70710678118654752440f : 1.f; }
  size_t act_n(int H, int C) const { return (size_t)B * H * H * C; }
  const float* W(int mod, const char* sub) const { const int i = n.P(mname(mod, sub)); return (dry || i < 0) ? nullptr : P[i]; }
  float* DW(int mod, const char* sub) const { const int i = n.P(mname(mod, sub)); return (dry || i < 0) ? nullptr : G[i]; }

#define TG_RUN(expr)                       \
  do {                                     \
    if (!dry) {                            \
      const int _rc = (expr);              \
      if (_rc) return _rc;                 \
    }                                      \
  } while (0)

  // ---- operator wrappers (scratch is a temporary above `top`) ---------------------------------------------------------
  int conv(const float* x, const float* w, const float* b, float* y, int Cin, int Cout, int H, int k, int stride, int dpad,
           int up2, int layout, const float* res = nullptr, const float* temb = nullptr, const void* planes = nullptr) {
    const size_t m = top;
    float* sc = alloc_bytes(csd_conv_scratch_bytes(B, Cin, Cout, H, H, k, up2));
    TG_RUN(conv2d_impl(x, w, b, res, temb, y, B, Cin, Cout, H, H, k, stride, dpad, up2, prec, layout, sc, s, planes));
    top = m;
    return CSD_OK;
  }
  int wgrad(const float* x, const float* dy, float* dw, int Cin, int Cout, int H, int k, int stride, int dpad, int up2, int layout) {
    const size_t m = top;
    float* sc = alloc_bytes(csd_conv_wgrad_scratch_bytes(B, Cin, Cout, H, H, k, stride, up2));
    TG_RUN(csd_conv2d_wgrad_ex(x, dy, dw, B, Cin, Cout, H, H, k, stride, dpad, up2, layout | (prec == CSD_PREC_F32 ? 0 : 4), sc, s));
    top = m;
    return CSD_OK;
  }
  // (mask != null: nn.Dropout of the activated tensor in the same pass; the mask csd_dropout would draw for (seed, drop_id))
  // (planes != null: the fp16 operand planes of y for the conv that follows, written by the same pass - conv2d_operand_planes())
  int gn(const float* x, const float* gamma, const float* beta, float* y, float* rs, float* ms, int C, int H, int a, float* mask = nullptr,
         uint64_t drop_id = 0, void* planes = nullptr, int plane_count = 0) {
    const size_t m = top;
    float* sc = alloc_bytes(csd_groupnorm_nhwc_scratch_bytes(B, C, H * H));
    TG_RUN(groupnorm_act_dropout_nhwc(x, gamma, beta, y, rs, ms, mask, p_drop, seed, drop_id, B, C, H * H, G_of(C), 1e-6f, a, sc, s, planes,
                                      plane_count));
    top = m;
    return CSD_OK;
  }
  // dx, dgamma / dbeta written to their parameter slots
  int gn_bwd(const float* x, const float* gamma, const float* beta, const float* rs, const float* ms, const float* dy, float* dx,
             float* dgamma, float* dbeta, int C, int H, int a, const float* add = nullptr) {
    const size_t m = top;
    float* grow = alloc((size_t)B * C);
    float* brow = alloc((size_t)B * C);
    float* sc = alloc_bytes(csd_groupnorm_nhwc_scratch_bytes(B, C, H * H));
    TG_RUN(groupnorm_act_backward_nhwc_add(x, gamma, beta, rs, ms, dy, add, dx, grow, brow, C, B, C, H * H, G_of(C), a, sc, s));
    TG_RUN(sum_rows2(grow, dgamma, brow, dbeta, B, C, s));
    top = m;
    return CSD_OK;
  }
  int copy(const float* src, float* dst, size_t nfl) {
    CSD_CHECK_HIP(hipMemcpyAsync(dst, src, nfl * sizeof(float), hipMemcpyDeviceToDevice, s));
    return CSD_OK;
  }
  // out[b, c] = sum over pixels of dy [B, H*H, C]
  int sum_pixels(const float* dy, float* out, int C, int H) {
    const size_t m = top;
    float* sc = alloc_bytes(csd_sum_pixels_scratch_bytes(B, H * H, C));
    TG_RUN(csd_sum_pixels_nhwc(dy, out, B, H * H, C, sc, s));
    top = m;
    return CSD_OK;
  }
  int bias_grad(const float* dy, float* db, int C, int H) {          // conv bias: batch + pixel sum of an NHWC gradient
    const size_t m = top;
    float* bc = alloc((size_t)B * C);
    int rc = sum_pixels(dy, bc, C, H);
    if (rc) return rc;
    TG_RUN(csd_sum_rows(bc, db, B, C, s));
    top = m;
    return CSD_OK;
  }
  int add_into(float* dst, const float* src, size_t nfl) { TG_RUN(csd_axpby(dst, src, dst, 1.f, 1.f, 0.f, 1.f, (int64_t)nfl, s)); return CSD_OK; }
  // FIR resampling by 2 of [Bn, H, H, C] (upsample_2d / downsample_2d, up_or_down_sampling.py:196-257).  adjoint: the gradient of
  // the OTHER direction's forward = this geometry with the flipped taps and the other gain (op/upfirdn2d.py:20-87: up and down
  // swapped, kernel flipped, pads g_pad): d upsample = down geometry x 4, d downsample = up geometry / 4.
  int fir(const float* x, float* y, int Bn, int H, int C, bool up_geometry, bool adjoint) {
    TG_RUN(fir_resample_nhwc_launch(x, y, Bn, H, H, C, n.cfg.fir_kernel, up_geometry ? 1 : 0, s,
                                    adjoint ? (up_geometry ? 0.25f : 4.f) : 1.f, adjoint ? 1 : 0));
    return CSD_OK;
  }
  int scale_into(const float* src, float* dst, float f, size_t nfl) { TG_RUN(csd_axpby(src, nullptr, dst, f, 0.f, 0.f, 1.f, (int64_t)nfl, s)); return CSD_OK; }
  int launch_blocks(size_t n) const { return (int)std::min<size_t>((n + 255) / 256, 256 * 16); }

  // gradient routing: the first contribution becomes the tensor's gradient buffer, later ones are added
  int contribute(int tid, float* g) {
    TT& t = st.t[tid];
    if (!t.g) { t.g = g; return CSD_OK; }
    return add_into(t.g, g, act_n(t.H, t.C));
  }
  int new_tensor(int H, int C, bool allocate = true) {
    TT t;
    t.H = H; t.C = C;
    if (allocate) t.p = alloc(act_n(H, C));
    st.t.push_back(t);
    return (int)st.t.size() - 1;
  }

  // =====================================================================================================================
  // forward (models/ddpm.py:149-213 with model.train())
  // =====================================================================================================================
  int res_fwd(const Module& m, int in, int* out_tid) {
    const csd_unet_config& c = n.cfg;
    const int H = st.t[in].H, cin = m.cin, cout = m.cout;
    const float* h = st.t[in].p;
    TStep sp;
    sp.kind = TS_RES; sp.mod = m.idx; sp.in0 = in;
    // persistent tensors first (the backward reads them), temporaries - the convs' operand planes, the shortcut - above them
    float* a0 = alloc(act_n(H, cin));                // act(GroupNorm_0(h)): Conv_0's operand
    float* rs0 = alloc((size_t)B * cin); float* ms0 = alloc((size_t)B * cin);
    float* c0 = alloc(act_n(H, cout));               // Conv_0(.) + Dense_0(act(temb))[:, None, None, :]: GroupNorm_1's input
    float* a1 = alloc(act_n(H, cout));               // dropout(act(GroupNorm_1(.))): Conv_1's operand
    float* rs1 = alloc((size_t)B * cout); float* ms1 = alloc((size_t)B * cout);
    float* mask = nullptr;
    ++drop_count;
    sp.drop_id = (call << 16) + (uint64_t)drop_count;
    if (p_drop > 0.f) mask = alloc(act_n(H, cout));
    const int out = new_tensor(H, cout);
    float* o = st.t[out].p;
    int rc;
    {
      const size_t mk = top;
      // the GroupNorm's apply pass also writes the fp16 operand planes Conv_0 would otherwise split a0 into in a pass of its own
      const int np0 = conv2d_operand_planes(B, cin, cout, H, H, 3, 1, 0, 0, prec);
      float* pl0 = np0 ? alloc_bytes((size_t)np0 * B * H * H * cin * 2) : nullptr;
      rc = gn(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), a0, rs0, ms0, cin, H, act, nullptr, 0, pl0, np0);
      if (rc) return rc;
      float* d = nullptr;
      if (c.conditional) {                           // the time-embedding row rides in Conv_0's epilogue
        d = alloc((size_t)B * cout);
        // (act(temb) is computed once per forward: every block's Dense_0 used to redo it inside its own latency-bound launch)
        TG_RUN(csd_linear(st.temb2_act, W(m.idx, "Dense_0.weight"), W(m.idx, "Dense_0.bias"), d, B, 4 * c.nf, cout, CSD_ACT_NONE, s));
      }
      rc = conv(a0, W(m.idx, "Conv_0.weight"), W(m.idx, "Conv_0.bias"), c0, cin, cout, H, 3, 1, 0, 0, 3, nullptr, d, pl0);
      if (rc) return rc;
      top = mk;
    }
    {
      const size_t mk = top;
      const int np1 = conv2d_operand_planes(B, cout, cout, H, H, 3, 1, 0, 0, prec);
      float* pl1 = np1 ? alloc_bytes((size_t)np1 * B * H * H * cout * 2) : nullptr;
      rc = gn(c0, W(m.idx, "GroupNorm_1.weight"), W(m.idx, "GroupNorm_1.bias"), a1, rs1, ms1, cout, H, act, mask, sp.drop_id, pl1, np1);   // (dropout in the apply pass)
      if (rc) return rc;
      if (cin != cout) {                             // NIN shortcut: h . W + b through the transposed-weight flag (no W^T copy)
        float* sc = alloc(act_n(H, cout));
        rc = conv(h, W(m.idx, "NIN_0.W"), W(m.idx, "NIN_0.b"), sc, cin, cout, H, 1, 1, 0, 0, 3 | 4);
        if (rc) return rc;
        rc = conv(a1, W(m.idx, "Conv_1.weight"), W(m.idx, "Conv_1.bias"), o, cout, cout, H, 3, 1, 0, 0, 3, sc, nullptr, pl1);   // + shortcut in the epilogue
        if (rc) return rc;
      } else {
        rc = conv(a1, W(m.idx, "Conv_1.weight"), W(m.idx, "Conv_1.bias"), o, cout, cout, H, 3, 1, 0, 0, 3, h, nullptr, pl1);
        if (rc) return rc;
      }
      top = mk;
    }
    sp.out = out;
    sp.sv[0] = a0; sp.sv[1] = rs0; sp.sv[2] = ms0; sp.sv[3] = c0; sp.sv[4] = a1; sp.sv[5] = rs1; sp.sv[6] = ms1; sp.sv[7] = mask;
    st.steps.push_back(sp);
    *out_tid = out;
    return CSD_OK;
  }

  // ResnetBlockBigGANpp.forward (models/layerspp.py:242-274): h = act(GroupNorm_0(x)); up / down: h, x = FIR(h), FIR(x); h = Conv_0(h)
  // + Dense_0(act(temb)); h = Conv_1(dropout(act(GroupNorm_1(h)))); x = Conv_2(x) if the shape changes; (x + h) / sqrt(2)
  int respp_fwd(const Module& m, int in, int* out_tid) {
    const csd_unet_config& c = n.cfg;
    const int H = st.t[in].H, cin = m.cin, cout = m.cout;
    const int Ho = m.up ? 2 * H : (m.down ? H / 2 : H);
    const bool resample = m.up || m.down, has_sc = cin != cout || resample;
    const float* h = st.t[in].p;
    TStep sp;
    sp.kind = TS_RES; sp.mod = m.idx; sp.in0 = in;
    float* a0 = alloc(act_n(H, cin));                // act(GroupNorm_0(x))
    float* rs0 = alloc((size_t)B * cin); float* ms0 = alloc((size_t)B * cin);
    float *hr = a0, *xr = nullptr;
    if (resample) { hr = alloc(act_n(Ho, cin)); xr = alloc(act_n(Ho, cin)); }
    float* c0 = alloc(act_n(Ho, cout));
    float* a1 = alloc(act_n(Ho, cout));
    float* rs1 = alloc((size_t)B * cout); float* ms1 = alloc((size_t)B * cout);
    float* mask = nullptr;
    ++drop_count;
    sp.drop_id = (call << 16) + (uint64_t)drop_count;
    if (p_drop > 0.f) mask = alloc(act_n(Ho, cout));
    const int out = new_tensor(Ho, cout);
    float* o = st.t[out].p;
    int rc;
    {
      const size_t mk = top;
      const int np0 = resample ? 0 : conv2d_operand_planes(B, cin, cout, H, H, 3, 1, 0, 0, prec);
      float* pl0 = np0 ? alloc_bytes((size_t)np0 * B * H * H * cin * 2) : nullptr;
      rc = gn(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), a0, rs0, ms0, cin, H, act, nullptr, 0, pl0, np0);
      if (rc) return rc;
      if (resample) {
        rc = fir(a0, hr, B, H, cin, m.up != 0, false); if (rc) return rc;
        rc = fir(h, xr, B, H, cin, m.up != 0, false); if (rc) return rc;
      }
      float* d = alloc((size_t)B * cout);            // (time-conditional networks only: build_modules_ncsnpp)
      TG_RUN(csd_linear(st.temb2_act, W(m.idx, "Dense_0.weight"), W(m.idx, "Dense_0.bias"), d, B, 4 * c.nf, cout, CSD_ACT_NONE, s));
      rc = conv(hr, W(m.idx, "Conv_0.weight"), W(m.idx, "Conv_0.bias"), c0, cin, cout, Ho, 3, 1, 0, 0, 3, nullptr, d, pl0);
      if (rc) return rc;
      top = mk;
    }
    {
      const size_t mk = top;
      const int np1 = conv2d_operand_planes(B, cout, cout, Ho, Ho, 3, 1, 0, 0, prec);
      float* pl1 = np1 ? alloc_bytes((size_t)np1 * B * Ho * Ho * cout * 2) : nullptr;
      rc = gn(c0, W(m.idx, "GroupNorm_1.weight"), W(m.idx, "GroupNorm_1.bias"), a1, rs1, ms1, cout, Ho, act, mask, sp.drop_id, pl1, np1);
      if (rc) return rc;
      const float* shortcut = h;
      if (has_sc) {                                  // Conv_2: 1x1 on the (resampled) block input
        float* sc = alloc(act_n(Ho, cout));
        rc = conv(resample ? xr : h, W(m.idx, "Conv_2.weight"), W(m.idx, "Conv_2.bias"), sc, cin, cout, Ho, 1, 1, 0, 0, 3);
        if (rc) return rc;
        shortcut = sc;
      }
      rc = conv(a1, W(m.idx, "Conv_1.weight"), W(m.idx, "Conv_1.bias"), o, cout, cout, Ho, 3, 1, 0, 0, 3, shortcut, nullptr, pl1);
      if (rc) return rc;
      if (skip_scale() != 1.f) { rc = scale_into(o, o, skip_scale(), act_n(Ho, cout)); if (rc) return rc; }
      top = mk;
    }
    sp.out = out;
    sp.sv[0] = hr; sp.sv[1] = rs0; sp.sv[2] = ms0; sp.sv[3] = c0; sp.sv[4] = a1; sp.sv[5] = rs1; sp.sv[6] = ms1; sp.sv[7] = mask; sp.sv[8] = xr;
    st.steps.push_back(sp);
    *out_tid = out;
    return CSD_OK;
  }

  int attn_fwd(const Module& m, int in, int* out_tid) {
    const int H = st.t[in].H, C = m.cin;
    const float* h = st.t[in].p;
    TStep sp;
    sp.kind = TS_ATTN; sp.mod = m.idx; sp.in0 = in;
    float* t = alloc(act_n(H, C));
    float* rs = alloc((size_t)B * C); float* ms = alloc((size_t)B * C);
    int rc = gn(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), t, rs, ms, C, H, CSD_ACT_NONE);
    if (rc) return rc;
    float* qkv = alloc(act_n(H, 3 * C));
    {                                                // q | k | v from ONE contraction: W = [NIN_0.W | NIN_1.W | NIN_2.W] ([C, 3C])
      const size_t mk = top;
      float* wc = alloc((size_t)C * 3 * C);
      float* bc = alloc(3 * C);
      for (int j = 0; j < 3 && !dry; ++j) {
        char wn[16], bn[16];
        snprintf(wn, sizeof(wn), "NIN_%d.W", j); snprintf(bn, sizeof(bn), "NIN_%d.b", j);
        hipLaunchKernelGGL(copy2d_kernel, dim3(launch_blocks((size_t)C * C)), dim3(256), 0, s, W(m.idx, wn), C, wc + j * C, 3 * C, C, C);
        CSD_LAUNCH_CHECK();
        TG_RUN(copy(W(m.idx, bn), bc + j * C, C));
      }
      rc = conv(t, wc, bc, qkv, C, 3 * C, H, 1, 1, 0, 0, 3 | 4);
      if (rc) return rc;
      top = mk;
    }
    float* a = alloc(act_n(H, C));
    TG_RUN(csd_attention_nhwc_prec(qkv, a, B, H * H, C, prec, s));      // (the arithmetic of the mode, as csd_unet_forward)
    const int out = new_tensor(H, C);
    float* o = st.t[out].p;
    rc = conv(a, W(m.idx, "NIN_3.W"), W(m.idx, "NIN_3.b"), o, C, C, H, 1, 1, 0, 0, 3 | 4, h);      // + h in the epilogue
    if (rc) return rc;
    if (skip_scale() != 1.f) { rc = scale_into(o, o, skip_scale(), act_n(H, C)); if (rc) return rc; }      // AttnBlockpp: (x + h) / sqrt(2)
    sp.out = out;
    sp.sv[0] = t; sp.sv[1] = rs; sp.sv[2] = ms; sp.sv[3] = qkv; sp.sv[4] = a;
    st.steps.push_back(sp);
    *out_tid = out;
    return CSD_OK;
  }

  int cat_fwd(int a, int b, int* out_tid) {
    const int H = st.t[a].H, Ca = st.t[a].C, Cb = st.t[b].C;
    const int out = new_tensor(H, Ca + Cb);
    if (!dry) {
      const size_t npix = (size_t)B * H * H;
      hipLaunchKernelGGL(concat_c_kernel, dim3(launch_blocks(npix * (Ca + Cb) / 4)), dim3(256), 0, s,
                         reinterpret_cast<const float4*>(st.t[a].p), Ca / 4, reinterpret_cast<const float4*>(st.t[b].p), Cb / 4,
                         reinterpret_cast<float4*>(st.t[out].p), npix);
      CSD_LAUNCH_CHECK();
    }
    TStep sp;
    sp.kind = TS_CAT; sp.in0 = a; sp.in1 = b; sp.out = out;
    st.steps.push_back(sp);
    *out_tid = out;
    return CSD_OK;
  }

  int forward(const float* x, const float* y, const float* labels, float* out) {
    const csd_unet_config& c = n.cfg;
    if (c.arch == 1) return forward_ncsnpp(x, y, labels, out);
    const int S = c.image_size, cx = c.x_channels, cy = c.y_channels, cio = cx + cy, nf = c.nf;
    st.t.clear(); st.steps.clear();
    drop_count = 0;
    st.lin0 = 0; st.emb_dim = nf; st.fourier_mod = -1;
    // network input: cat(x, y) NCHW (models/ddpm.py:275-298 wrappers), 2h - 1 for data in [0, 1] (:163-168)
    const size_t hw = (size_t)S * S;
    st.xin = alloc((size_t)B * cio * hw);
    if (!dry) {
      CSD_CHECK_HIP(hipMemcpy2DAsync(st.xin, cio * hw * 4, x, cx * hw * 4, cx * hw * 4, B, hipMemcpyDeviceToDevice, s));
      if (cy) CSD_CHECK_HIP(hipMemcpy2DAsync(st.xin + cx * hw, cio * hw * 4, y, cy * hw * 4, cy * hw * 4, B, hipMemcpyDeviceToDevice, s));
      if (!c.centered) TG_RUN(csd_axpby(st.xin, nullptr, st.xin, 2.f, 0.f, -1.f, 1.f, (int64_t)((size_t)B * cio * hw), s));
    }
    size_t mi = 0;
    if (c.conditional) {                             // timestep embedding + 2-layer MLP (models/ddpm.py:153-160)
      st.emb = alloc((size_t)B * nf); st.temb1 = alloc((size_t)B * 4 * nf); st.temb2 = alloc((size_t)B * 4 * nf);
      TG_RUN(csd_timestep_embedding(labels, st.emb, B, nf, s));
      TG_RUN(csd_linear(st.emb, W(0, "weight"), W(0, "bias"), st.temb1, B, nf, 4 * nf, CSD_ACT_NONE, s));
      TG_RUN(csd_linear(st.temb1, W(1, "weight"), W(1, "bias"), st.temb2, B, 4 * nf, 4 * nf, act, s));
      st.temb2_act = alloc((size_t)B * 4 * nf);
      TG_RUN(csd_act(st.temb2, nullptr, st.temb2_act, act, (int64_t)B * 4 * nf, s));
      mi = 2;
    }
    int rc;
    std::vector<int> hs;
    {                                                // stem conv: NCHW in, NHWC out
      const Module& m = n.mods[mi++];
      const int t0 = new_tensor(S, nf);
      rc = conv(st.xin, W(m.idx, "weight"), W(m.idx, "bias"), st.t[t0].p, cio, nf, S, 3, 1, 0, 0, 2);
      if (rc) return rc;
      TStep sp;
      sp.kind = TS_STEM; sp.mod = m.idx; sp.out = t0;
      st.steps.push_back(sp);
      hs.push_back(t0);
    }
    auto is_attn = [&](int res) {
      for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
      return false;
    };
    int h = -1;
    for (int l = 0; l < c.n_levels; ++l) {
      for (int b = 0; b < c.num_res_blocks; ++b) {
        rc = res_fwd(n.mods[mi++], hs.back(), &h);
        if (rc) return rc;
        if (is_attn(st.t[h].H)) { rc = attn_fwd(n.mods[mi++], h, &h); if (rc) return rc; }
        hs.push_back(h);
      }
      if (l != c.n_levels - 1) {                     // Downsample: pad (0,1,0,1) + stride-2 conv (models/layers.py:619-625)
        const Module& m = n.mods[mi++];
        const int in = hs.back(), H = st.t[in].H, C = m.cin;
        const int o = new_tensor(H / 2, C);
        rc = conv(st.t[in].p, W(m.idx, "Conv_0.weight"), W(m.idx, "Conv_0.bias"), st.t[o].p, C, C, H, 3, 2, 1, 0, 3);
        if (rc) return rc;
        TStep sp;
        sp.kind = TS_DOWN; sp.mod = m.idx; sp.in0 = in; sp.out = o;
        st.steps.push_back(sp);
        hs.push_back(o);
      }
    }
    h = hs.back();
    rc = res_fwd(n.mods[mi++], h, &h); if (rc) return rc;
    rc = attn_fwd(n.mods[mi++], h, &h); if (rc) return rc;
    rc = res_fwd(n.mods[mi++], h, &h); if (rc) return rc;
    for (int l = c.n_levels - 1; l >= 0; --l) {
      for (int b = 0; b < c.num_res_blocks + 1; ++b) {
        int cat;
        rc = cat_fwd(h, hs.back(), &cat); if (rc) return rc;
        hs.pop_back();
        rc = res_fwd(n.mods[mi++], cat, &h); if (rc) return rc;
      }
      if (is_attn(st.t[h].H)) { rc = attn_fwd(n.mods[mi++], h, &h); if (rc) return rc; }
      if (l != 0) {                                  // Upsample: nearest x2 + conv (models/layers.py:600-604)
        const Module& m = n.mods[mi++];
        const int H = st.t[h].H, C = m.cin;
        const int o = new_tensor(2 * H, C);
        rc = conv(st.t[h].p, W(m.idx, "Conv_0.weight"), W(m.idx, "Conv_0.bias"), st.t[o].p, C, C, H, 3, 1, 0, 1, 3);
        if (rc) return rc;
        TStep sp;
        sp.kind = TS_UP; sp.mod = m.idx; sp.in0 = h; sp.out = o;
        st.steps.push_back(sp);
        h = o;
      }
    }
    CSD_REQUIRE(hs.empty() && mi + 2 == n.mods.size(), "train_forward: module walk out of step (%zu of %zu)", mi, n.mods.size());
    {                                                // head: act(GroupNorm) + conv, NHWC in, NCHW out
      const Module& mg = n.mods[mi];
      const Module& mc = n.mods[mi + 1];
      const int H = st.t[h].H, C = mg.cin;
      float* g = alloc(act_n(H, C));
      float* rs = alloc((size_t)B * C); float* ms = alloc((size_t)B * C);
      rc = gn(st.t[h].p, W(mg.idx, "weight"), W(mg.idx, "bias"), g, rs, ms, C, H, act);
      if (rc) return rc;
      rc = conv(g, W(mc.idx, "weight"), W(mc.idx, "bias"), out, C, c.out_channels, H, 3, 1, 0, 0, 1);
      if (rc) return rc;
      TStep sp;
      sp.kind = TS_HEAD; sp.mod = mg.idx; sp.in0 = h;
      sp.sv[0] = g; sp.sv[1] = rs; sp.sv[2] = ms;
      st.steps.push_back(sp);
    }
    st.fwd_top = top;
    return CSD_OK;
  }

  // =====================================================================================================================
  // forward, NCSN++ (models/ncsnpp.py:238-388 with model.train(); the module walk of build_modules_ncsnpp)
  // =====================================================================================================================
  int forward_ncsnpp(const float* x, const float* y, const float* labels, float* out) {
    const csd_unet_config& c = n.cfg;
    const int S = c.image_size, cx = c.x_channels, cy = c.y_channels, cio = cx + cy, nf = c.nf;
    st.t.clear(); st.steps.clear();
    drop_count = 0;
    const size_t hw = (size_t)S * S;
    st.xin = alloc((size_t)B * cio * hw);            // cat(x, y) NCHW, 2h - 1 for data in [0, 1] (ncsnpp.py:266-268)
    if (!dry) {
      CSD_CHECK_HIP(hipMemcpy2DAsync(st.xin, cio * hw * 4, x, cx * hw * 4, cx * hw * 4, B, hipMemcpyDeviceToDevice, s));
      if (cy) CSD_CHECK_HIP(hipMemcpy2DAsync(st.xin + cx * hw, cio * hw * 4, y, cy * hw * 4, cy * hw * 4, B, hipMemcpyDeviceToDevice, s));
      if (!c.centered) TG_RUN(csd_axpby(st.xin, nullptr, st.xin, 2.f, 0.f, -1.f, 1.f, (int64_t)((size_t)B * cio * hw), s));
    }
    size_t mi = 0;
    st.fourier_mod = -1;
    st.emb_dim = nf;
    if (c.embedding_type == 1) {                     // Gaussian Fourier features of the label (layerspp.py:32-41; W is a fixed buffer)
      const Module& fm = n.mods[mi++];
      st.fourier_mod = fm.idx;
      st.emb_dim = 2 * nf;
      st.emb = alloc((size_t)B * 2 * nf);
      TG_RUN(csd_fourier_embedding(labels, W(fm.idx, "W"), st.emb, B, nf, s));
    } else {
      st.emb = alloc((size_t)B * nf);
      TG_RUN(csd_timestep_embedding(labels, st.emb, B, nf, s));
    }
    st.lin0 = (int)mi;
    st.temb1 = alloc((size_t)B * 4 * nf); st.temb2 = alloc((size_t)B * 4 * nf);
    TG_RUN(csd_linear(st.emb, W((int)mi, "weight"), W((int)mi, "bias"), st.temb1, B, st.emb_dim, 4 * nf, CSD_ACT_NONE, s));
    TG_RUN(csd_linear(st.temb1, W((int)mi + 1, "weight"), W((int)mi + 1, "bias"), st.temb2, B, 4 * nf, 4 * nf, act, s));
    st.temb2_act = alloc((size_t)B * 4 * nf);
    TG_RUN(csd_act(st.temb2, nullptr, st.temb2_act, act, (int64_t)B * 4 * nf, s));
    mi += 2;
    int rc;
    std::vector<int> hs;
    {                                                // first conv: NCHW in, NHWC out
      const Module& m = n.mods[mi++];
      const int t0 = new_tensor(S, nf);
      rc = conv(st.xin, W(m.idx, "weight"), W(m.idx, "bias"), st.t[t0].p, cio, nf, S, 3, 1, 0, 0, 2);
      if (rc) return rc;
      TStep sp;
      sp.kind = TS_STEM; sp.mod = m.idx; sp.out = t0;
      st.steps.push_back(sp);
      hs.push_back(t0);
    }
    auto is_attn = [&](int res) {
      for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
      return false;
    };
    const float* pyr_in = st.xin;                    // input pyramid (NCHW = [B * cio] single-channel images for the FIR pass)
    int pyr_side = S;
    int h = -1;
    for (int l = 0; l < c.n_levels; ++l) {
      for (int b = 0; b < c.num_res_blocks; ++b) {
        rc = respp_fwd(n.mods[mi++], hs.back(), &h); if (rc) return rc;
        if (is_attn(st.t[h].H)) { rc = attn_fwd(n.mods[mi++], h, &h); if (rc) return rc; }
        hs.push_back(h);
      }
      if (l != c.n_levels - 1) {
        rc = respp_fwd(n.mods[mi++], hs.back(), &h); if (rc) return rc;      // the down block
        if (c.progressive_input == 1) {              // pyramid_downsample + Combine 'sum': Conv_0(pyramid) + h (layerspp.py:44-59)
          const Module& cm = n.mods[mi++];
          const int side = st.t[h].H, C = cm.cout;
          float* pn = alloc((size_t)B * cio * side * side);
          rc = fir(pyr_in, pn, B * cio, pyr_side, 1, false, false); if (rc) return rc;
          pyr_in = pn; pyr_side = side;
          const int o = new_tensor(side, C);
          rc = conv(pn, W(cm.idx, "Conv_0.weight"), W(cm.idx, "Conv_0.bias"), st.t[o].p, cio, C, side, 1, 1, 0, 0, 2); if (rc) return rc;
          rc = add_into(st.t[o].p, st.t[h].p, act_n(side, C)); if (rc) return rc;
          TStep sp;
          sp.kind = TS_COMBINE; sp.mod = cm.idx; sp.in0 = h; sp.out = o; sp.sv[0] = pn;
          st.steps.push_back(sp);
          h = o;
        }
        hs.push_back(h);
      }
    }
    h = hs.back();
    rc = respp_fwd(n.mods[mi++], h, &h); if (rc) return rc;
    rc = attn_fwd(n.mods[mi++], h, &h); if (rc) return rc;
    rc = respp_fwd(n.mods[mi++], h, &h); if (rc) return rc;
    int pyr = -1;                                    // output pyramid: tensor id of the previous (coarser) level, NCHW data
    for (int l = c.n_levels - 1; l >= 0; --l) {
      for (int b = 0; b < c.num_res_blocks + 1; ++b) {
        int cat;
        rc = cat_fwd(h, hs.back(), &cat); if (rc) return rc;
        hs.pop_back();
        rc = respp_fwd(n.mods[mi++], cat, &h); if (rc) return rc;
      }
      if (is_attn(st.t[h].H)) { rc = attn_fwd(n.mods[mi++], h, &h); if (rc) return rc; }
      if (c.progressive == 1) {                      // output_skip: pyramid = Conv(act(GroupNorm(h))) + pyramid_upsample(pyramid) (ncsnpp.py:340-352)
        const Module& mg = n.mods[mi++];
        const Module& mc = n.mods[mi++];
        const int H = st.t[h].H, C = mg.cin;
        float* g = alloc(act_n(H, C));
        float* rs = alloc((size_t)B * C); float* ms = alloc((size_t)B * C);
        rc = gn(st.t[h].p, W(mg.idx, "weight"), W(mg.idx, "bias"), g, rs, ms, C, H, act); if (rc) return rc;
        const int pt = new_tensor(H, cio, false);
        st.t[pt].p = l == 0 ? out : alloc((size_t)B * cio * H * H);
        rc = conv(g, W(mc.idx, "weight"), W(mc.idx, "bias"), st.t[pt].p, C, cio, H, 3, 1, 0, 0, 1); if (rc) return rc;
        TStep sp;
        sp.kind = TS_PYR; sp.mod = mg.idx; sp.in0 = h; sp.in1 = pyr; sp.out = pt;
        sp.sv[0] = g; sp.sv[1] = rs; sp.sv[2] = ms;
        sp.flag = (pyr >= 0 ? 1 : 0) | (l == 0 ? 2 : 0);
        if (pyr >= 0) {
          const size_t mk = top;
          float* up = alloc((size_t)B * cio * H * H);
          rc = fir(st.t[pyr].p, up, B * cio, H / 2, 1, true, false); if (rc) return rc;
          rc = add_into(st.t[pt].p, up, (size_t)B * cio * H * H); if (rc) return rc;
          top = mk;
        }
        st.steps.push_back(sp);
        pyr = pt;
      }
      if (l != 0) { rc = respp_fwd(n.mods[mi++], h, &h); if (rc) return rc; }      // the up block
    }
    if (c.progressive != 1) {                        // act(GroupNorm) + conv, NHWC in, NCHW out
      const Module& mg = n.mods[mi];
      const Module& mc = n.mods[mi + 1];
      mi += 2;
      const int H = st.t[h].H, C = mg.cin;
      float* g = alloc(act_n(H, C));
      float* rs = alloc((size_t)B * C); float* ms = alloc((size_t)B * C);
      rc = gn(st.t[h].p, W(mg.idx, "weight"), W(mg.idx, "bias"), g, rs, ms, C, H, act); if (rc) return rc;
      rc = conv(g, W(mc.idx, "weight"), W(mc.idx, "bias"), out, C, c.out_channels, H, 3, 1, 0, 0, 1); if (rc) return rc;
      TStep sp;
      sp.kind = TS_HEAD; sp.mod = mg.idx; sp.in0 = h;
      sp.sv[0] = g; sp.sv[1] = rs; sp.sv[2] = ms;
      st.steps.push_back(sp);
    }
    CSD_REQUIRE(hs.empty() && mi == n.mods.size(), "train_forward (ncsnpp): module walk out of step (%zu of %zu)", mi, n.mods.size());
    st.fwd_top = top;
    return CSD_OK;
  }

  // =====================================================================================================================
  // backward
  // =====================================================================================================================
  // Linear y = act_in(x) W^T + b with x [B, K], W [N, K]: dW, db from dy; dact (gradient w.r.t. act_in(x)) is ADDED to dact_acc
  int linear_bwd(const float* x, int act_in, const float* w, const float* dy, float* dw, float* db, float* dact_acc, int K, int N) {
    const size_t m = top;
    const float* a = x;
    if (act_in != CSD_ACT_NONE) {
      float* ax = alloc((size_t)B * K);
      TG_RUN(csd_act(x, nullptr, ax, act_in, (int64_t)B * K, s));
      a = ax;
    }
    TG_RUN(csd_bgemm(dy, a, dw, N, K, B, 1, N, K, 1, K, 1, 1, 0, 0, 0, 1.f, s));         // dW[n,k] = sum_b dy[b,n] a[b,k]
    TG_RUN(csd_sum_rows(dy, db, B, N, s));
    if (dact_acc) {
      float* da = alloc((size_t)B * K);
      TG_RUN(csd_bgemm(dy, w, da, B, K, N, N, 1, K, 1, K, 1, 1, 0, 0, 0, 1.f, s));       // da[b,k] = sum_n dy[b,n] W[n,k]
      int rc = add_into(dact_acc, da, (size_t)B * K);
      if (rc) return rc;
    }
    top = m;
    return CSD_OK;
  }

  int res_bwd(const TStep& sp, float* dtemb_act) {
    const csd_unet_config& c = n.cfg;
    const Module& m = n.mods[sp.mod];
    const TT& tin = st.t[sp.in0];
    const int H = tin.H, cin = m.cin, cout = m.cout;
    const float* h = tin.p;
    float* dout = st.t[sp.out].g;
    float *a0 = sp.sv[0], *rs0 = sp.sv[1], *ms0 = sp.sv[2], *c0 = sp.sv[3], *a1 = sp.sv[4], *rs1 = sp.sv[5], *ms1 = sp.sv[6], *mask = sp.sv[7];
    float* dh = alloc(act_n(H, cin));                // gradient w.r.t. the block input (kept until its producer has consumed it)
    const size_t mk = top;
    int rc;
    // Conv_1
    rc = wgrad(a1, dout, DW(m.idx, "Conv_1.weight"), cout, cout, H, 3, 1, 0, 0, 3); if (rc) return rc;
    rc = bias_grad(dout, DW(m.idx, "Conv_1.bias"), cout, H); if (rc) return rc;
    float* d1 = alloc(act_n(H, cout));
    rc = conv(dout, W(m.idx, "Conv_1.weight"), nullptr, d1, cout, cout, H, 3, 1, 0, 0, 3 | 4); if (rc) return rc;
    if (mask) TG_RUN(csd_mul(d1, mask, d1, (int64_t)act_n(H, cout), s));
    // GroupNorm_1 + act
    float* d1b = alloc(act_n(H, cout));
    rc = gn_bwd(c0, W(m.idx, "GroupNorm_1.weight"), W(m.idx, "GroupNorm_1.bias"), rs1, ms1, d1, d1b, DW(m.idx, "GroupNorm_1.weight"),
                DW(m.idx, "GroupNorm_1.bias"), cout, H, act);
    if (rc) return rc;
    d1 = d1b;
    // Conv_0 bias and the time-embedding row share the per-(sample, channel) pixel sums of d1
    {
      float* dd = alloc((size_t)B * cout);
      rc = sum_pixels(d1, dd, cout, H); if (rc) return rc;
      TG_RUN(csd_sum_rows(dd, DW(m.idx, "Conv_0.bias"), B, cout, s));
      if (c.conditional) {
        rc = linear_bwd(st.temb2_act, CSD_ACT_NONE, W(m.idx, "Dense_0.weight"), dd, DW(m.idx, "Dense_0.weight"), DW(m.idx, "Dense_0.bias"), dtemb_act,
                        4 * c.nf, cout);
        if (rc) return rc;
      }
    }
    rc = wgrad(a0, d1, DW(m.idx, "Conv_0.weight"), cin, cout, H, 3, 1, 0, 0, 3); if (rc) return rc;
    float* d0 = alloc(act_n(H, cin));
    rc = conv(d1, W(m.idx, "Conv_0.weight"), nullptr, d0, cout, cin, H, 3, 1, 0, 0, 3 | 4); if (rc) return rc;
    if (cin != cout) {
      // NIN_0: y = h.W + b, W [cin, cout]: dW[i,o] = sum_p h[p,i] dout[p,o] = the 1x1 weight gradient with the operands swapped;
      // the shortcut's data gradient dout . W^T = the OIHW 1x1 conv of dout with W read as [O = cin, I = cout]; it joins dh inside
      // the GroupNorm_0 backward
      rc = wgrad(dout, h, DW(m.idx, "NIN_0.W"), cout, cin, H, 1, 1, 0, 0, 3); if (rc) return rc;
      rc = bias_grad(dout, DW(m.idx, "NIN_0.b"), cout, H); if (rc) return rc;
      float* dsc = alloc(act_n(H, cin));
      rc = conv(dout, W(m.idx, "NIN_0.W"), nullptr, dsc, cout, cin, H, 1, 1, 0, 0, 3); if (rc) return rc;
      rc = gn_bwd(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), rs0, ms0, d0, dh, DW(m.idx, "GroupNorm_0.weight"),
                  DW(m.idx, "GroupNorm_0.bias"), cin, H, act, dsc);
      if (rc) return rc;
    } else {
      rc = gn_bwd(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), rs0, ms0, d0, dh, DW(m.idx, "GroupNorm_0.weight"),
                  DW(m.idx, "GroupNorm_0.bias"), cin, H, act, dout);
      if (rc) return rc;
    }
    top = mk;
    return contribute(sp.in0, dh);
  }

  int respp_bwd(const TStep& sp, float* dtemb_act) {
    const csd_unet_config& c = n.cfg;
    const Module& m = n.mods[sp.mod];
    const TT& tin = st.t[sp.in0];
    const int H = tin.H, cin = m.cin, cout = m.cout;
    const int Ho = m.up ? 2 * H : (m.down ? H / 2 : H);
    const bool resample = m.up || m.down, has_sc = cin != cout || resample;
    const float* h = tin.p;
    float *op0 = sp.sv[0], *rs0 = sp.sv[1], *ms0 = sp.sv[2], *c0 = sp.sv[3], *a1 = sp.sv[4], *rs1 = sp.sv[5], *ms1 = sp.sv[6], *mask = sp.sv[7],
          *xr = sp.sv[8];
    float* dh = alloc(act_n(H, cin));
    const size_t mk = top;
    int rc;
    float* dout = st.t[sp.out].g;
    if (skip_scale() != 1.f) {                       // out = (x + h) / sqrt(2)
      float* ds = alloc(act_n(Ho, cout));
      rc = scale_into(dout, ds, skip_scale(), act_n(Ho, cout)); if (rc) return rc;
      dout = ds;
    }
    // Conv_1
    rc = wgrad(a1, dout, DW(m.idx, "Conv_1.weight"), cout, cout, Ho, 3, 1, 0, 0, 3); if (rc) return rc;
    rc = bias_grad(dout, DW(m.idx, "Conv_1.bias"), cout, Ho); if (rc) return rc;
    float* d1 = alloc(act_n(Ho, cout));
    rc = conv(dout, W(m.idx, "Conv_1.weight"), nullptr, d1, cout, cout, Ho, 3, 1, 0, 0, 3 | 4); if (rc) return rc;
    if (mask) TG_RUN(csd_mul(d1, mask, d1, (int64_t)act_n(Ho, cout), s));
    float* d1b = alloc(act_n(Ho, cout));
    rc = gn_bwd(c0, W(m.idx, "GroupNorm_1.weight"), W(m.idx, "GroupNorm_1.bias"), rs1, ms1, d1, d1b, DW(m.idx, "GroupNorm_1.weight"),
                DW(m.idx, "GroupNorm_1.bias"), cout, Ho, act);
    if (rc) return rc;
    d1 = d1b;
    {
      float* dd = alloc((size_t)B * cout);
      rc = sum_pixels(d1, dd, cout, Ho); if (rc) return rc;
      TG_RUN(csd_sum_rows(dd, DW(m.idx, "Conv_0.bias"), B, cout, s));
      rc = linear_bwd(st.temb2_act, CSD_ACT_NONE, W(m.idx, "Dense_0.weight"), dd, DW(m.idx, "Dense_0.weight"), DW(m.idx, "Dense_0.bias"), dtemb_act,
                      4 * c.nf, cout);
      if (rc) return rc;
    }
    rc = wgrad(op0, d1, DW(m.idx, "Conv_0.weight"), cin, cout, Ho, 3, 1, 0, 0, 3); if (rc) return rc;
    float* d0 = alloc(act_n(Ho, cin));
    rc = conv(d1, W(m.idx, "Conv_0.weight"), nullptr, d0, cout, cin, Ho, 3, 1, 0, 0, 3 | 4); if (rc) return rc;
    if (resample) {                                  // through the FIR of the h branch: the other direction's geometry, flipped taps
      float* d0l = alloc(act_n(H, cin));
      rc = fir(d0, d0l, B, Ho, cin, m.up == 0, true); if (rc) return rc;
      d0 = d0l;
    }
    const float* add = dout;                         // the identity shortcut's gradient
    if (has_sc) {                                    // Conv_2 (1x1 OIHW) on the (resampled) block input
      rc = wgrad(resample ? xr : h, dout, DW(m.idx, "Conv_2.weight"), cin, cout, Ho, 1, 1, 0, 0, 3); if (rc) return rc;
      rc = bias_grad(dout, DW(m.idx, "Conv_2.bias"), cout, Ho); if (rc) return rc;
      float* dsc = alloc(act_n(Ho, cin));
      rc = conv(dout, W(m.idx, "Conv_2.weight"), nullptr, dsc, cout, cin, Ho, 1, 1, 0, 0, 3 | 4); if (rc) return rc;
      if (resample) {
        float* dscl = alloc(act_n(H, cin));
        rc = fir(dsc, dscl, B, Ho, cin, m.up == 0, true); if (rc) return rc;
        dsc = dscl;
      }
      add = dsc;
    }
    rc = gn_bwd(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), rs0, ms0, d0, dh, DW(m.idx, "GroupNorm_0.weight"),
                DW(m.idx, "GroupNorm_0.bias"), cin, H, act, add);
    if (rc) return rc;
    top = mk;
    return contribute(sp.in0, dh);
  }

  int attn_bwd(const TStep& sp) {
    const Module& m = n.mods[sp.mod];
    const TT& tin = st.t[sp.in0];
    const int H = tin.H, C = m.cin;
    const float* h = tin.p;
    float* dout = st.t[sp.out].g;
    float *t = sp.sv[0], *rs = sp.sv[1], *ms = sp.sv[2], *qkv = sp.sv[3], *a = sp.sv[4];
    float* dh = alloc(act_n(H, C));
    const size_t mk = top;
    int rc;
    if (skip_scale() != 1.f) {                       // AttnBlockpp: out = (x + h) / sqrt(2)
      float* ds = alloc(act_n(H, C));
      rc = scale_into(dout, ds, skip_scale(), act_n(H, C)); if (rc) return rc;
      dout = ds;
    }
    // NIN_3
    rc = wgrad(dout, a, DW(m.idx, "NIN_3.W"), C, C, H, 1, 1, 0, 0, 3); if (rc) return rc;
    rc = bias_grad(dout, DW(m.idx, "NIN_3.b"), C, H); if (rc) return rc;
    float* da = alloc(act_n(H, C));
    rc = conv(dout, W(m.idx, "NIN_3.W"), nullptr, da, C, C, H, 1, 1, 0, 0, 3); if (rc) return rc;
    // attention core
    float* dqkv = alloc(act_n(H, 3 * C));
    {
      const size_t m2 = top;
      float* sc = alloc_bytes(csd_attention_backward_scratch_bytes(B, C, H * H, 1));
      TG_RUN(csd_attention_backward_nhwc(qkv, da, dqkv, B, H * H, C, sc, s));
      top = m2;
    }
    // q | k | v contraction: gradients of the concatenated weight, scattered back to NIN_0..2
    float* wc = alloc((size_t)C * 3 * C);
    float* dwc = alloc((size_t)C * 3 * C);
    float* dbc = alloc(3 * C);
    rc = wgrad(dqkv, t, dwc, 3 * C, C, H, 1, 1, 0, 0, 3); if (rc) return rc;           // dW[i, o] with o over 3C
    rc = bias_grad(dqkv, dbc, 3 * C, H); if (rc) return rc;
    for (int j = 0; j < 3 && !dry; ++j) {
      char wn[16], bn[16];
      snprintf(wn, sizeof(wn), "NIN_%d.W", j); snprintf(bn, sizeof(bn), "NIN_%d.b", j);
      hipLaunchKernelGGL(copy2d_kernel, dim3(launch_blocks((size_t)C * C)), dim3(256), 0, s, W(m.idx, wn), C, wc + j * C, 3 * C, C, C);
      CSD_LAUNCH_CHECK();
      hipLaunchKernelGGL(copy2d_kernel, dim3(launch_blocks((size_t)C * C)), dim3(256), 0, s, dwc + j * C, 3 * C, DW(m.idx, wn), C, C, C);
      CSD_LAUNCH_CHECK();
      TG_RUN(copy(dbc + j * C, DW(m.idx, bn), C));
    }
    rc = conv(dqkv, wc, nullptr, da, 3 * C, C, H, 1, 1, 0, 0, 3); if (rc) return rc;    // dt = dqkv . Wcat^T (Wcat read as OIHW [C, 3C])
    rc = gn_bwd(h, W(m.idx, "GroupNorm_0.weight"), W(m.idx, "GroupNorm_0.bias"), rs, ms, da, dh, DW(m.idx, "GroupNorm_0.weight"),
                DW(m.idx, "GroupNorm_0.bias"), C, H, CSD_ACT_NONE, dout);
    if (rc) return rc;
    top = mk;
    return contribute(sp.in0, dh);
  }

  int backward(const float* d_out) {
    const csd_unet_config& c = n.cfg;
    const int nf = c.nf;
    top = st.fwd_top;
    int rc;
    float* dtemb_act = nullptr;                      // sum over the blocks of dDense . W: gradient w.r.t. act(temb2)
    if (c.conditional) {
      dtemb_act = alloc((size_t)B * 4 * nf);
      if (!dry) CSD_CHECK_HIP(hipMemsetAsync(dtemb_act, 0, (size_t)B * 4 * nf * sizeof(float), s));
    }
    // the marks may follow the steps only if the recorded steps walk the modules in ascending order (they do: the forward consumes
    // all_modules front to back) - otherwise every mark waits for the end of the backward
    bool ordered = true;
    {
      int last = -1;
      for (const TStep& sp : st.steps) {
        if (sp.mod < 0) continue;
        const int idx = n.mods[sp.mod].idx;
        if (idx < last) ordered = false;
        last = idx;
      }
    }
    for (int i = (int)st.steps.size() - 1; i >= 0; --i) {
      const TStep& sp = st.steps[i];
      if (i + 1 < (int)st.steps.size() && ordered && st.steps[i + 1].mod >= 0) {      // step i + 1 and everything after it is complete
        rc = marks_reached(n.mods[st.steps[i + 1].mod].idx); if (rc) return rc;
      }
      switch (sp.kind) {
        case TS_HEAD: {
          const Module& mg = n.mods[sp.mod];
          const Module& mc = n.mods[sp.mod + 1];
          const TT& tin = st.t[sp.in0];
          const int H = tin.H, C = mg.cin, Co = c.out_channels;
          float* dh = alloc(act_n(H, C));
          const size_t mk = top;
          rc = wgrad(sp.sv[0], d_out, DW(mc.idx, "weight"), C, Co, H, 3, 1, 0, 0, 1); if (rc) return rc;
          {                                          // bias: NCHW gradient -> per-(sample, channel) sums -> batch sum
            float* bc = alloc((size_t)B * Co);
            TG_RUN(csd_sum_inner(d_out, bc, (int64_t)B * Co, (int64_t)H * H, s));
            TG_RUN(csd_sum_rows(bc, DW(mc.idx, "bias"), B, Co, s));
          }
          float* dg = alloc(act_n(H, C));
          rc = conv(d_out, W(mc.idx, "weight"), nullptr, dg, Co, C, H, 3, 1, 0, 0, 2 | 4); if (rc) return rc;
          rc = gn_bwd(tin.p, W(mg.idx, "weight"), W(mg.idx, "bias"), sp.sv[1], sp.sv[2], dg, dh, DW(mg.idx, "weight"), DW(mg.idx, "bias"), C,
                      H, act);
          if (rc) return rc;
          top = mk;
          rc = contribute(sp.in0, dh); if (rc) return rc;
          break;
        }
        case TS_RES: rc = (c.arch == 1 ? respp_bwd(sp, dtemb_act) : res_bwd(sp, dtemb_act)); if (rc) return rc; break;
        case TS_COMBINE: {                           // out = Conv_0(pyramid) + h: the pyramid is data (no gradient), h's gradient is dout
          const Module& m = n.mods[sp.mod];
          const TT& to = st.t[sp.out];
          const int cio = c.x_channels + c.y_channels;
          rc = wgrad(sp.sv[0], to.g, DW(m.idx, "Conv_0.weight"), cio, m.cout, to.H, 1, 1, 0, 0, 2); if (rc) return rc;
          rc = bias_grad(to.g, DW(m.idx, "Conv_0.bias"), m.cout, to.H); if (rc) return rc;
          rc = contribute(sp.in0, to.g); if (rc) return rc;
          break;
        }
        case TS_PYR: {                               // pyramid level: Conv(act(GroupNorm(h))) (+ FIR-up of the coarser level), NCHW
          const Module& mg = n.mods[sp.mod];
          const Module& mc = n.mods[sp.mod + 1];
          const TT& tin = st.t[sp.in0];
          const int H = tin.H, C = mg.cin, Co = c.x_channels + c.y_channels;
          const float* dP = (sp.flag & 2) ? d_out : st.t[sp.out].g;
          float* dh = alloc(act_n(H, C));
          float* dprev = (sp.flag & 1) ? alloc((size_t)B * Co * (H / 2) * (H / 2)) : nullptr;
          const size_t mk = top;
          rc = wgrad(sp.sv[0], dP, DW(mc.idx, "weight"), C, Co, H, 3, 1, 0, 0, 1); if (rc) return rc;
          {
            float* bc = alloc((size_t)B * Co);
            TG_RUN(csd_sum_inner(dP, bc, (int64_t)B * Co, (int64_t)H * H, s));
            TG_RUN(csd_sum_rows(bc, DW(mc.idx, "bias"), B, Co, s));
          }
          float* dg = alloc(act_n(H, C));
          rc = conv(dP, W(mc.idx, "weight"), nullptr, dg, Co, C, H, 3, 1, 0, 0, 2 | 4); if (rc) return rc;
          rc = gn_bwd(tin.p, W(mg.idx, "weight"), W(mg.idx, "bias"), sp.sv[1], sp.sv[2], dg, dh, DW(mg.idx, "weight"), DW(mg.idx, "bias"), C,
                      H, act);
          if (rc) return rc;
          if (dprev) { rc = fir(dP, dprev, B * Co, H, 1, false, true); if (rc) return rc; }      // d pyramid_upsample
          top = mk;
          rc = contribute(sp.in0, dh); if (rc) return rc;
          if (dprev) st.t[sp.in1].g = dprev;         // (the coarser level's only consumer)
          break;
        }
        case TS_ATTN: rc = attn_bwd(sp); if (rc) return rc; break;
        case TS_CAT: {
          const TT& ta = st.t[sp.in0];
          const TT& tb = st.t[sp.in1];
          float* da = alloc(act_n(ta.H, ta.C));
          float* db = alloc(act_n(tb.H, tb.C));
          if (!dry) {
            const size_t npix = (size_t)B * ta.H * ta.H;
            hipLaunchKernelGGL(split_c_kernel, dim3(launch_blocks(npix * (ta.C + tb.C) / 4)), dim3(256), 0, s,
                               reinterpret_cast<const float4*>(st.t[sp.out].g), ta.C / 4, tb.C / 4, reinterpret_cast<float4*>(da),
                               reinterpret_cast<float4*>(db), npix);
            CSD_LAUNCH_CHECK();
          }
          rc = contribute(sp.in0, da); if (rc) return rc;
          rc = contribute(sp.in1, db); if (rc) return rc;
          break;
        }
        case TS_UP: {                                // nearest x2 + conv: dx = 2x2 block sums of the full-resolution data gradient
          const Module& m = n.mods[sp.mod];
          const TT& tin = st.t[sp.in0];
          const int H = tin.H, C = m.cin;
          float* dout = st.t[sp.out].g;
          float* dh = alloc(act_n(H, C));
          const size_t mk = top;
          rc = wgrad(tin.p, dout, DW(m.idx, "Conv_0.weight"), C, C, H, 3, 1, 0, 1, 3); if (rc) return rc;
          rc = bias_grad(dout, DW(m.idx, "Conv_0.bias"), C, 2 * H); if (rc) return rc;
          float* full = alloc(act_n(2 * H, C));
          rc = conv(dout, W(m.idx, "Conv_0.weight"), nullptr, full, C, C, 2 * H, 3, 1, 0, 0, 3 | 4); if (rc) return rc;
          TG_RUN(csd_sumpool2_nhwc(full, dh, B, H, H, C, s));
          top = mk;
          rc = contribute(sp.in0, dh); if (rc) return rc;
          break;
        }
        case TS_DOWN: {                              // stride-2 conv: dy zero-inserted on the odd grid positions, then a pad-1 conv
          const Module& m = n.mods[sp.mod];
          const TT& tin = st.t[sp.in0];
          const int H = tin.H, C = m.cin;
          float* dout = st.t[sp.out].g;
          float* dh = alloc(act_n(H, C));
          const size_t mk = top;
          rc = wgrad(tin.p, dout, DW(m.idx, "Conv_0.weight"), C, C, H, 3, 2, 1, 0, 3); if (rc) return rc;
          rc = bias_grad(dout, DW(m.idx, "Conv_0.bias"), C, H / 2); if (rc) return rc;
          float* z = alloc(act_n(H, C));
          TG_RUN(csd_zero_insert_odd_nhwc(dout, z, B, H / 2, H / 2, C, s));
          rc = conv(z, W(m.idx, "Conv_0.weight"), nullptr, dh, C, C, H, 3, 1, 0, 0, 3 | 4); if (rc) return rc;
          top = mk;
          rc = contribute(sp.in0, dh); if (rc) return rc;
          break;
        }
        case TS_STEM: {
          const Module& m = n.mods[sp.mod];
          const TT& to = st.t[sp.out];
          const int cio = c.x_channels + c.y_channels;
          rc = wgrad(st.xin, to.g, DW(m.idx, "weight"), cio, nf, to.H, 3, 1, 0, 0, 2); if (rc) return rc;
          rc = bias_grad(to.g, DW(m.idx, "bias"), nf, to.H); if (rc) return rc;
          break;
        }
      }
    }
    if (c.conditional) {                             // temb MLP (models/ddpm.py:153-160)
      const size_t mk = top;
      float* d2 = alloc((size_t)B * 4 * nf);         // gradient w.r.t. temb2 = act'(temb2) * dtemb_act
      TG_RUN(csd_act(st.temb2, dtemb_act, d2, act, (int64_t)B * 4 * nf, s));
      float* dact1 = alloc((size_t)B * 4 * nf);
      if (!dry) CSD_CHECK_HIP(hipMemsetAsync(dact1, 0, (size_t)B * 4 * nf * sizeof(float), s));
      const int l0 = st.lin0, l1 = st.lin0 + 1;
      rc = linear_bwd(st.temb1, act, W(l1, "weight"), d2, DW(l1, "weight"), DW(l1, "bias"), dact1, 4 * nf, 4 * nf); if (rc) return rc;
      float* d1 = alloc((size_t)B * 4 * nf);
      TG_RUN(csd_act(st.temb1, dact1, d1, act, (int64_t)B * 4 * nf, s));
      rc = linear_bwd(st.emb, CSD_ACT_NONE, W(l0, "weight"), d1, DW(l0, "weight"), DW(l0, "bias"), nullptr, st.emb_dim, 4 * nf); if (rc) return rc;
      if (st.fourier_mod >= 0 && !dry)               // the Fourier W is a fixed buffer of the reference (requires_grad = False): its slot reads zero
        CSD_CHECK_HIP(hipMemsetAsync(DW(st.fourier_mod, "W"), 0, (size_t)nf * sizeof(float), s));
      top = mk;
    }
    return marks_reached(-1);                        // (the embedding MLP holds the lowest module indices: everything is final now)
  }
#undef TG_RUN
};

static int train_check(const csd_unet* net) {
  CSD_REQUIRE(net, "train: null handle");
  const csd_unet_config& c = net->net.cfg;
  CSD_REQUIRE(c.arch == 0 || c.arch == 1, "train graph: arch %d has no planned training graph", c.arch);
  if (c.arch == 0) CSD_REQUIRE(c.resamp_with_conv, "train graph: resamp_with_conv = False is not provided");
  CSD_REQUIRE((c.x_channels + c.y_channels) >= 1, "train graph: no input channels");
  return CSD_OK;
}

}  // namespace csd

extern "C" size_t csd_unet_train_workspace_bytes(csd_unet* net, int B, float dropout_p) {
  if (!net || train_check(net) || B < 1) return 0;
  TrainState st;
  TG g(net->net, st, B, nullptr, true, nullptr, nullptr, reinterpret_cast<float*>(uintptr_t(256)));
  g.p_drop = dropout_p;
  if (g.forward(nullptr, nullptr, nullptr, nullptr)) return 0;
  if (g.backward(nullptr)) return 0;
  return g.peak * sizeof(float) + 256;
}

extern "C" int csd_unet_train_forward(csd_unet* net, const float* const* params, void* workspace, size_t workspace_bytes,
                                      const float* x, const float* y, const float* labels, float* out, int B, float dropout_p,
                                      uint64_t dropout_seed, uint64_t call_index, void* stream) {
  int rc = train_check(net);
  if (rc) return rc;
  CSD_REQUIRE(params && workspace && x && out && B >= 1, "train_forward: null argument");
  CSD_REQUIRE((net->net.cfg.y_channels == 0) == (y == nullptr), "train_forward: y must be given iff y_channels > 0");
  CSD_REQUIRE(!net->net.cfg.conditional || labels, "train_forward: labels required for a conditional network");
  CSD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "train_forward: workspace must be 256-byte aligned");
  CSD_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "train_forward: dropout_p %g out of range", dropout_p);
  const size_t need = csd_unet_train_workspace_bytes(net, B, dropout_p);
  if (workspace_bytes < need) {
    set_error("train_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    return CSD_ERR_WORKSPACE;
  }
  for (size_t i = 0; i < net->net.params.size(); ++i)
    CSD_REQUIRE(params[i], "train_forward: parameter %zu (%s) is null", i, net->net.params[i].name.c_str());
  train_state_purge_stale(&net->net, workspace);
  TrainState& st = train_state_of(&net->net, workspace);
  st.valid = false;
  TG g(net->net, st, B, (hipStream_t)stream, false, params, nullptr, static_cast<float*>(workspace));
  g.p_drop = dropout_p; g.seed = dropout_seed; g.call = call_index;
  rc = g.forward(x, y, labels, out);
  if (rc) return rc;
  st.valid = true; st.B = B; st.ws = workspace; st.p_drop = dropout_p; st.call = call_index;
  return CSD_OK;
}

extern "C" int csd_unet_train_release(csd_unet* net, const void* workspace) {
  int rc = train_check(net);
  if (rc) return rc;
  train_state_erase_one(&net->net, workspace);
  return CSD_OK;
}

extern "C" int csd_unet_train_release_call(csd_unet* net, const void* workspace, uint64_t call_index) {
  int rc = train_check(net);
  if (rc) return rc;
  train_state_erase_one(&net->net, workspace, &call_index);
  return CSD_OK;
}

extern "C" int csd_unet_backward(csd_unet* net, const float* const* params, float* const* grads, void* workspace,
                                 size_t workspace_bytes, const float* d_out, int B, uint64_t call_index, void* stream) {
  int rc = train_check(net);
  if (rc) return rc;
  CSD_REQUIRE(params && grads && workspace && d_out, "backward: null argument");
  TrainState* sp = train_state_find(&net->net, workspace);
  if (!sp || !sp->valid || sp->B != B || sp->ws != workspace) {
    set_error("backward: no matching csd_unet_train_forward (same handle, workspace and batch) precedes this call");
    return CSD_ERR_STATE;
  }
  TrainState& st = *sp;
  if (st.call != call_index) {
    set_error("backward: the workspace holds the activations of csd_unet_train_forward call %llu, not of call %llu (a later forward "
              "re-used it before this backward ran)", (unsigned long long)st.call, (unsigned long long)call_index);
    return CSD_ERR_STATE;
  }
  const size_t need = csd_unet_train_workspace_bytes(net, B, st.p_drop);
  if (workspace_bytes < need) {
    set_error("backward: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    return CSD_ERR_WORKSPACE;
  }
  for (size_t i = 0; i < net->net.params.size(); ++i)
    CSD_REQUIRE(params[i] && grads[i], "backward: parameter / gradient pointer %zu (%s) is null", i, net->net.params[i].name.c_str());
  for (auto& t : st.t) t.g = nullptr;
  TG g(net->net, st, B, (hipStream_t)stream, false, params, grads, static_cast<float*>(workspace));
  {
    std::lock_guard<std::mutex> lk(g_train_mu);
    auto it = g_marks.find(&net->net);
    g.gm = it == g_marks.end() ? nullptr : &it->second;
  }
  rc = g.backward(d_out);
  if (!rc && g.gm) ++g.gm->epoch;
  st.valid = false;                                  // the saved activations are consumed (gradient buffers overwrote nothing, but one backward per forward)
  return rc;
}

// ---- gradient-ready marks + the event / stream helpers a data-parallel caller needs to overlap its all-reduce with the backward ----
extern "C" int csd_unet_backward_marks(csd_unet* net, const int* first_module, void* const* events, int n) {
  int rc = csd::train_check(net);
  if (rc) return rc;
  CSD_REQUIRE(n >= 0 && (n == 0 || (first_module && events)), "backward_marks: null argument");
  std::lock_guard<std::mutex> lk(csd::g_train_mu);
  if (n == 0) { csd::g_marks.erase(&net->net); return CSD_OK; }
  csd::GradMarks& gm = csd::g_marks[&net->net];
  gm.marks.clear();
  for (int k = 0; k < n; ++k) {
    CSD_REQUIRE(events[k], "backward_marks: event %d is null", k);
    gm.marks.emplace_back(first_module[k], static_cast<hipEvent_t>(events[k]));
  }
  std::stable_sort(gm.marks.begin(), gm.marks.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  return CSD_OK;
}
extern "C" uint64_t csd_unet_backward_marks_epoch(csd_unet* net) {
  if (!net) return 0;
  std::lock_guard<std::mutex> lk(csd::g_train_mu);
  auto it = csd::g_marks.find(&net->net);
  return it == csd::g_marks.end() ? 0 : it->second.epoch;
}
extern "C" void* csd_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}
extern "C" int csd_event_destroy(void* event) {
  if (event) CSD_CHECK_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return CSD_OK;
}
extern "C" int csd_stream_wait_event(void* stream, void* event) {
  CSD_REQUIRE(event, "stream_wait_event: null event");
  CSD_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
  return CSD_OK;
}
extern "C" int csd_event_query(void* event) {      /* 1: complete, 0: not yet, < 0: error */
  CSD_REQUIRE(event, "event_query: null event");
  const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  csd::set_error("event_query: %s", hipGetErrorString(e));
  return CSD_ERR_HIP;
}
