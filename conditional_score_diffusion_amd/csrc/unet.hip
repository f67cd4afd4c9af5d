// unet.hip - graph executor for the DDPM-family score network + the fused PC sampling loop,
// and the C-ABI entry points around them.
//
// Replaces (reference, behaviour only): models/ddpm.py:80-213 (DDPM.__init__/forward),
// :275-298 (paired wrappers), the per-step glue of sampling/conditional.py:180-226 and
// sampling/unconditional.py:194-226.  The module list is rebuilt from the config values exactly
// as DDPM.__init__ does, so parameter names/indices equal the reference state_dict
// ("all_modules.{i}.Conv_0.weight" ...).  Execution is a flat, pre-planned list of kernel
// launches on ONE stream: no per-step host objects, no allocation, no synchronisation.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <memory>
#include <vector>

#include "common.h"

namespace csd {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

// ---------------------------------------------------------------------------------------------
// in-library profiler: HIP events around every launch of the network / sampler, recorded on the
// SAME stream the kernels run on (bench.py's roofline numbers come from here)
// ---------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes, abytes; };
struct Profiler {
  bool on = false;
  unsigned mask = ~0u;        // launch classes that get events (bit = CSD_PROF_* id)
  int step_stride = 1;        // csd_pc_sample: only every step_stride-th PC step is bracketed
  bool step_on = true;
  std::vector<ProfRec> recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
};
// per calling thread: a host thread that drives its own handle / stream profiles its own launches (csd_profile_* carry no handle)
static thread_local Profiler g_prof;

struct ProfScope {
  hipStream_t s;
  hipEvent_t b = nullptr;
  ProfScope(int cls, double flops, double bytes, hipStream_t s_, double abytes = -1.0) : s(s_) {
    if (!g_prof.on || !g_prof.step_on || !((g_prof.mask >> cls) & 1u)) return;
    if (g_prof.used == g_prof.pool.size()) {
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
      g_prof.pool.push_back({e0, e1});
    }
    auto& pr = g_prof.pool[g_prof.used++];
    (void)hipEventRecord(pr.first, s);
    b = pr.second;
    g_prof.recs.push_back({pr.first, pr.second, cls, flops, bytes, abytes < 0 ? bytes : abytes});
  }
  ~ProfScope() {
    if (b) (void)hipEventRecord(b, s);
  }
};

// ---------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------
struct Param {
  std::string name;
  int ndim;
  int64_t shape[4];
  int64_t numel;
  const float* ptr = nullptr;
};

enum ModKind { M_LINEAR, M_CONV3, M_RES, M_ATTN, M_DOWN, M_UP, M_GN, M_FOURIER, M_COMBINE };
struct Module {
  ModKind kind;
  int idx;
  int cin = 0, cout = 0;   // res / conv3 / linear ; attn/down/up/gn use cin as "channels"
  int up = 0, down = 0;    // NCSN++ ResnetBlockBigGANpp: FIR resampling of h and x inside the block
};

// one packed convolution weight (+bias): where it lives inside the packed buffer
struct PackedConv {
  ConvPlan proto;          // channel-level fields only (C0,C1,Cout,taps,KC,NT,CoutPad)
  size_t w_off = 0, b_off = 0;   // float offsets in the packed buffer
  int ns = 0;                    // 0: fp32 kernel layout; 1/2: fp16 kernel layout with ns planes
  bool pw = false;               // ns != 0 and the layer runs on the pointwise fp16 kernel (conv_pw16.hip)
  bool q = false;                // ns != 0 and the layer runs on the quad-wave fp16 kernel (conv_f16_q.hip)
  bool ff = false;               // ns != 0 and the layer runs on the fused-prologue kernel (conv_ff.hip): fp32 sources, no gn_apply16
  int tap_cout = 0;              // pw and the layer is a 3x3 convolution with tap_cout (<= 6) output channels in its tap-partial form (conv_pw16.hip):
                                 // proto is the POINTWISE contraction to 9 * tap_cout partial channels, a 9-tap gather finishes it
  bool stem = false;             // the DDPM-family first layer fused with the input assembly (stem.hip): x, y (NCHW) -> nf channels NHWC
  bool up4 = false;              // q and the layer is the nearest-x2 Upsample conv in its phase-decomposed form (4 x 2x2 taps; conv_f16_q.hip UP4)
  struct Src { int param_w, param_b, layout, cout_src, cout_off, cin_src; };
  std::vector<Src> srcs;
};

struct Net;

// ---------------------------------------------------------------------------------------------
// execution plan for one batch size
// ---------------------------------------------------------------------------------------------
enum OpKind { OP_ASSEMBLE, OP_STEM, OP_TEMB, OP_FOURIER, OP_FIR, OP_FIR2, OP_GN_APPLY32, OP_LINEAR, OP_GN_STATS, OP_GN_FINAL, OP_GN_FINAL_TILES, OP_GN_APPLY16, OP_GN_FUSED16, OP_GN_STATFIN, OP_CONV, OP_ATTN, OP_AVGPOOL,
              OP_UPNEAR, OP_TO_NCHW, OP_TAPSUM,
              OP_FORK, OP_JOIN };     // batch-chunk region (build_plan): the chunk streams start behind / the main stream resumes behind them

static const size_t NONE = (size_t)-1;

struct Op {
  OpKind kind;
  // generic offsets (floats) into workspace unless stated
  size_t a = NONE, b = NONE, c = NONE, d = NONE, e = NONE, out = NONE;
  size_t stats = NONE;                // conv: per-tile GroupNorm partials it writes; GN_FINAL_TILES: a = source 0's, b = source 1's
  size_t temb_base = NONE;            // offset of dense_all (conv epilogue time-embedding source)
  size_t pk0 = NONE, pk1 = NONE;      // offsets into the packed buffer
  int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0;
  ConvPlan cp;
  GNPlan gp;
  int act = 0;
  int out_external = 0;               // conv writes to the caller's NCHW output
  int side = 0;                       // 1: launched on the side stream (after the main stream's work so far); 2: the main stream waits for it first
  int nb = 0;                         // batch of THIS launch (0: the plan's) - the ops of a batch chunk carry the chunk's size
  int stream = 0;                     // 0: the caller's stream; k > 0: chunk stream k - 1 (between an OP_FORK and its OP_JOIN)
  int chunk = 0;                      // k + 1 for the ops of batch chunk k (chunk 0 runs on the caller's stream)
  float fscale = 1.f;                 // conv: epilogue out_scale
  size_t temb_col = NONE;             // column offset inside dense_all
  int temb_stride = 0;
  int cls = CSD_PROF_OTHER;           // profiling class
  double flops = 0, bytes = 0;        // algorithmic flops of this launch; bytes THIS kernel has to move
  double abytes = -1;                 // SURVEY 8(d) bytes of the layer (input + output tensor, fp32) where they differ from `bytes`
};

struct Plan {
  int B = 0;
  std::vector<Op> ops;
  size_t ws_floats = 0;
  int64_t launches = 0;
  double flops = 0, bytes = 0;
};

class Arena {
 public:
  Arena() {}
  explicit Arena(size_t base) : base_(base) {}       // a sub-arena: offsets start at `base` (a block reserved in the parent)
  size_t alloc(size_t nfloats) { return base_ + alloc_local(nfloats); }
  void release(size_t off) {
    if (off == NONE || off < base_ || off - base_ >= top_) return;      // (not ours: e.g. a slice of a parent tensor handed into a chunk)
    release_local(off - base_);
  }
  size_t peak() const { return peak_; }

 private:
  size_t base_ = 0;
  size_t alloc_local(size_t nfloats) {
    nfloats = (nfloats + 63) / 64 * 64;   // 256-byte granules
    // best fit in the free list
    int best = -1;
    for (int i = 0; i < (int)free_.size(); ++i)
      if (free_[i].second >= nfloats && (best < 0 || free_[i].second < free_[best].second)) best = i;
    size_t off;
    if (best >= 0) {
      off = free_[best].first;
      if (free_[best].second == nfloats) free_.erase(free_.begin() + best);
      else { free_[best].first += nfloats; free_[best].second -= nfloats; }
    } else {
      // extend the top (merge with a trailing free block if there is one)
      off = top_;
      for (int i = 0; i < (int)free_.size(); ++i)
        if (free_[i].first + free_[i].second == top_) { off = free_[i].first; free_.erase(free_.begin() + i); break; }
      top_ = off + nfloats;
    }
    live_[off] = nfloats;
    peak_ = std::max(peak_, top_);
    return off;
  }
  void release_local(size_t off) {
    auto it = live_.find(off);
    if (it == live_.end()) return;
    size_t n = it->second;
    live_.erase(it);
    // coalesce
    for (int i = 0; i < (int)free_.size();) {
      if (free_[i].first + free_[i].second == off) { off = free_[i].first; n += free_[i].second; free_.erase(free_.begin() + i); }
      else if (off + n == free_[i].first) { n += free_[i].second; free_.erase(free_.begin() + i); }
      else ++i;
    }
    free_.push_back({off, n});
  }
  std::vector<std::pair<size_t, size_t>> free_;
  std::map<size_t, size_t> live_;
  size_t top_ = 0, peak_ = 0;
};

struct Net {
  csd_unet_config cfg;
  std::vector<Module> mods;
  std::vector<Param> params;
  std::map<std::string, int> pindex;
  // packed layout
  std::vector<PackedConv> pconvs;
  std::map<std::string, int> pconv_by_name;          // "3.Conv_0", "13.qkv", "13.NIN_3", "2" ...
  struct Copy { int param; size_t off; };            // raw fp32 copies (linear, GN affine, dense)
  std::vector<Copy> copies;
  std::map<std::string, size_t> copy_off;
  size_t packed_floats = 0;
  size_t dense_all_off = 0, dense_all_bias_off = 0;  // concatenated Dense_0 of every res block
  int dense_total = 0;
  std::map<int, int> dense_col;                      // module idx -> first column
  bool packed_once = false;
  struct CopyDesc* copy_tab = nullptr;      // device table of pack_all's raw copies (+ its host copy and the event behind the upload)
  size_t copy_tab_cap = 0;
  std::vector<struct CopyDesc> copy_host;
  hipEvent_t copy_ev = nullptr;
  std::map<int, std::unique_ptr<Plan>> plans;
  int in_cpad = 8;
  // second stream for the ResnetBlock shortcut contraction (independent of the block's GroupNorm -> conv chain until the second conv
  // adds it): an HBM-bound pointwise kernel that fills the CUs a 3x3 launch leaves idle in its last round
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_ready() {
    if (side) return true;
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; return false; }
    if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(side);
      side = nullptr;
      return false;
    }
    return true;
  }
  // batch-chunk streams (build_plan: the <= 20^2 levels of a big batch run as CHUNKS concurrent sub-batches - their kernels are bound by
  // per-launch latency, not by throughput, and a sample's bits do not depend on the batch it runs in)
  static constexpr int MAX_CHUNKS = 4;
  hipStream_t cstream[MAX_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_cfork = nullptr, ev_cjoin[MAX_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
  bool chunks_ready(int n_extra) {      // the first n_extra chunk streams exist (created on demand: a stream occupies a hardware queue)
    for (int k = 0; k < n_extra && k < MAX_CHUNKS; ++k) {
      if (cstream[k]) continue;
      if (hipStreamCreateWithFlags(&cstream[k], hipStreamNonBlocking) != hipSuccess) { cstream[k] = nullptr; return false; }
      if (hipEventCreateWithFlags(&ev_cjoin[k], hipEventDisableTiming) != hipSuccess) return false;
    }
    return ev_cfork != nullptr || hipEventCreateWithFlags(&ev_cfork, hipEventDisableTiming) == hipSuccess;
  }
  ~Net() {
    if (copy_ev) { (void)hipEventSynchronize(copy_ev); (void)hipEventDestroy(copy_ev); }
    if (copy_tab) (void)hipFree(copy_tab);
    for (int k = 0; k < MAX_CHUNKS; ++k) {
      if (cstream[k]) { (void)hipStreamSynchronize(cstream[k]); (void)hipStreamDestroy(cstream[k]); }
      if (ev_cjoin[k]) (void)hipEventDestroy(ev_cjoin[k]);
    }
    if (ev_cfork) (void)hipEventDestroy(ev_cfork);
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }

  int add_param(const std::string& name, std::initializer_list<int64_t> shape) {
    Param p;
    p.name = name;
    p.ndim = (int)shape.size();
    p.numel = 1;
    int i = 0;
    for (auto s : shape) { p.shape[i++] = s; p.numel *= s; }
    for (; i < 4; ++i) p.shape[i] = 1;
    params.push_back(p);
    pindex[name] = (int)params.size() - 1;
    return (int)params.size() - 1;
  }
  int P(const std::string& name) const {
    auto it = pindex.find(name);
    return it == pindex.end() ? -1 : it->second;
  }
};

static std::string mname(int idx, const char* sub) {
  char buf[96];
  if (sub && sub[0]) snprintf(buf, sizeof(buf), "all_modules.%d.%s", idx, sub);
  else snprintf(buf, sizeof(buf), "all_modules.%d", idx);
  return buf;
}

// ---- module list: mirrors DDPM.__init__ (models/ddpm.py:96-147) -------------------------------
static int build_modules_ncsnpp(Net& n);

static int build_modules(Net& n) {
  const csd_unet_config& c = n.cfg;
  CSD_REQUIRE(c.arch == 0 || c.arch == 1, "unet: arch %d not supported (0 = DDPM family, 1 = NCSN++)", c.arch);
  if (c.arch == 1) return build_modules_ncsnpp(n);
  CSD_REQUIRE(c.n_levels >= 1 && c.n_levels <= CSD_MAX_LEVELS, "unet: bad n_levels %d", c.n_levels);
  CSD_REQUIRE(c.nf % 32 == 0, "unet: nf=%d must be a multiple of 32 (GroupNorm(32) + 32-wide MFMA tiles)", c.nf);
  CSD_REQUIRE(c.image_size % (1 << (c.n_levels - 1)) == 0, "unet: image_size %d not divisible by 2^%d",
              c.image_size, c.n_levels - 1);
  CSD_REQUIRE(c.x_channels >= 1 && c.x_channels + c.y_channels <= 8, "unet: x+y channels must be <= 8");
  CSD_REQUIRE(c.act >= CSD_ACT_SWISH && c.act <= CSD_ACT_ELU, "unet: bad activation id %d", c.act);
  CSD_REQUIRE(c.precision >= CSD_PREC_F32 && c.precision <= CSD_PREC_F16F8, "unet: bad precision id %d", c.precision);
  auto add = [&](ModKind k, int cin, int cout) {
    Module m;
    m.kind = k; m.idx = (int)n.mods.size(); m.cin = cin; m.cout = cout;
    n.mods.push_back(m);
  };
  auto is_attn = [&](int res) {
    for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
    return false;
  };
  const int nf = c.nf;
  if (c.conditional) { add(M_LINEAR, nf, 4 * nf); add(M_LINEAR, 4 * nf, 4 * nf); }
  add(M_CONV3, c.x_channels + c.y_channels, nf);
  std::vector<int> hs_c{nf};
  int in_ch = nf;
  for (int l = 0; l < c.n_levels; ++l) {
    const int res = c.image_size >> l;
    for (int b = 0; b < c.num_res_blocks; ++b) {
      const int out_ch = nf * c.ch_mult[l];
      add(M_RES, in_ch, out_ch);
      in_ch = out_ch;
      if (is_attn(res)) add(M_ATTN, in_ch, in_ch);
      hs_c.push_back(in_ch);
    }
    if (l != c.n_levels - 1) { add(M_DOWN, in_ch, in_ch); hs_c.push_back(in_ch); }
  }
  add(M_RES, in_ch, in_ch);
  add(M_ATTN, in_ch, in_ch);
  add(M_RES, in_ch, in_ch);
  for (int l = c.n_levels - 1; l >= 0; --l) {
    const int res = c.image_size >> l;
    for (int b = 0; b < c.num_res_blocks + 1; ++b) {
      const int out_ch = nf * c.ch_mult[l];
      add(M_RES, in_ch + hs_c.back(), out_ch);
      hs_c.pop_back();
      in_ch = out_ch;
    }
    if (is_attn(res)) add(M_ATTN, in_ch, in_ch);
    if (l != 0) add(M_UP, in_ch, in_ch);
  }
  add(M_GN, in_ch, in_ch);
  add(M_CONV3, in_ch, c.out_channels);

  // parameter table in state_dict order
  const int temb = 4 * nf;
  for (auto& m : n.mods) {
    switch (m.kind) {
      case M_LINEAR:
        n.add_param(mname(m.idx, "weight"), {m.cout, m.cin});
        n.add_param(mname(m.idx, "bias"), {m.cout});
        break;
      case M_CONV3:
        n.add_param(mname(m.idx, "weight"), {m.cout, m.cin, 3, 3});
        n.add_param(mname(m.idx, "bias"), {m.cout});
        break;
      case M_GN:
        n.add_param(mname(m.idx, "weight"), {m.cin});
        n.add_param(mname(m.idx, "bias"), {m.cin});
        break;
      case M_DOWN:
      case M_UP:
        if (c.resamp_with_conv) {
          n.add_param(mname(m.idx, "Conv_0.weight"), {m.cin, m.cin, 3, 3});
          n.add_param(mname(m.idx, "Conv_0.bias"), {m.cin});
        }
        break;
      case M_ATTN:
        n.add_param(mname(m.idx, "GroupNorm_0.weight"), {m.cin});
        n.add_param(mname(m.idx, "GroupNorm_0.bias"), {m.cin});
        for (int j = 0; j < 4; ++j) {
          char w[32], b[32];
          snprintf(w, sizeof(w), "NIN_%d.W", j);
          snprintf(b, sizeof(b), "NIN_%d.b", j);
          n.add_param(mname(m.idx, w), {m.cin, m.cin});
          n.add_param(mname(m.idx, b), {m.cin});
        }
        break;
      case M_RES:
        n.add_param(mname(m.idx, "GroupNorm_0.weight"), {m.cin});
        n.add_param(mname(m.idx, "GroupNorm_0.bias"), {m.cin});
        n.add_param(mname(m.idx, "Conv_0.weight"), {m.cout, m.cin, 3, 3});
        n.add_param(mname(m.idx, "Conv_0.bias"), {m.cout});
        if (c.conditional) {
          n.add_param(mname(m.idx, "Dense_0.weight"), {m.cout, temb});
          n.add_param(mname(m.idx, "Dense_0.bias"), {m.cout});
        }
        n.add_param(mname(m.idx, "GroupNorm_1.weight"), {m.cout});
        n.add_param(mname(m.idx, "GroupNorm_1.bias"), {m.cout});
        n.add_param(mname(m.idx, "Conv_1.weight"), {m.cout, m.cout, 3, 3});
        n.add_param(mname(m.idx, "Conv_1.bias"), {m.cout});
        if (m.cin != m.cout) {
          n.add_param(mname(m.idx, "NIN_0.W"), {m.cin, m.cout});
          n.add_param(mname(m.idx, "NIN_0.b"), {m.cout});
        }
        break;
    }
  }
  return CSD_OK;
}

// ---- NCSN++ module list: mirrors NCSNpp.__init__ (models/ncsnpp.py:44-236) for resblock_type 'biggan', fir = True,
// progressive in {none, output_skip}, progressive_input in {none, input_skip}, progressive_combine 'sum' ----
static int ncsnpp_groups(int c) { return std::min(c / 4, 32); }     // layerspp.py:67,219,231; ncsnpp.py:200-233

static int build_modules_ncsnpp(Net& n) {
  const csd_unet_config& c = n.cfg;
  CSD_REQUIRE(c.n_levels >= 1 && c.n_levels <= CSD_MAX_LEVELS, "ncsnpp: bad n_levels %d", c.n_levels);
  CSD_REQUIRE(c.nf % 8 == 0, "ncsnpp: nf=%d must be a multiple of 8", c.nf);
  CSD_REQUIRE(c.image_size % (1 << (c.n_levels - 1)) == 0, "ncsnpp: image_size %d not divisible by 2^%d", c.image_size,
              c.n_levels - 1);
  CSD_REQUIRE(c.x_channels >= 1 && c.x_channels + c.y_channels <= 8, "ncsnpp: x+y channels must be <= 8");
  CSD_REQUIRE(c.out_channels == c.x_channels + c.y_channels, "ncsnpp: the network maps its %d input channels to as many outputs",
              c.x_channels + c.y_channels);
  CSD_REQUIRE(c.act >= CSD_ACT_SWISH && c.act <= CSD_ACT_ELU, "ncsnpp: bad activation id %d", c.act);
  CSD_REQUIRE(c.precision >= CSD_PREC_F32 && c.precision <= CSD_PREC_F16F8, "ncsnpp: bad precision id %d", c.precision);
  CSD_REQUIRE(c.conditional, "ncsnpp: only time-conditional networks are supported");
  CSD_REQUIRE(c.progressive >= 0 && c.progressive <= 1 && c.progressive_input >= 0 && c.progressive_input <= 1,
              "ncsnpp: 'residual' progressive growing is not supported");
  CSD_REQUIRE(c.n_fir == 4, "ncsnpp: a 4-tap FIR kernel is required (got %d taps)", c.n_fir);
  auto add = [&](ModKind k, int cin, int cout, int up = 0, int down = 0) {
    Module m;
    m.kind = k; m.idx = (int)n.mods.size(); m.cin = cin; m.cout = cout; m.up = up; m.down = down;
    n.mods.push_back(m);
  };
  auto is_attn = [&](int res) {
    for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
    return false;
  };
  const int nf = c.nf, channels = c.x_channels + c.y_channels;
  int embed_dim = nf;
  if (c.embedding_type == 1) { add(M_FOURIER, nf, 2 * nf); embed_dim = 2 * nf; }
  add(M_LINEAR, embed_dim, 4 * nf);
  add(M_LINEAR, 4 * nf, 4 * nf);
  add(M_CONV3, channels, nf);
  std::vector<int> hs_c{nf};
  int in_ch = nf;
  for (int l = 0; l < c.n_levels; ++l) {
    const int res = c.image_size >> l;
    for (int b = 0; b < c.num_res_blocks; ++b) {
      const int out_ch = nf * c.ch_mult[l];
      add(M_RES, in_ch, out_ch);
      in_ch = out_ch;
      if (is_attn(res)) add(M_ATTN, in_ch, in_ch);
      hs_c.push_back(in_ch);
    }
    if (l != c.n_levels - 1) {
      add(M_RES, in_ch, in_ch, 0, 1);
      if (c.progressive_input == 1) add(M_COMBINE, channels, in_ch);
      hs_c.push_back(in_ch);
    }
  }
  add(M_RES, in_ch, in_ch);
  add(M_ATTN, in_ch, in_ch);
  add(M_RES, in_ch, in_ch);
  for (int l = c.n_levels - 1; l >= 0; --l) {
    const int res = c.image_size >> l;
    for (int b = 0; b < c.num_res_blocks + 1; ++b) {
      const int out_ch = nf * c.ch_mult[l];
      add(M_RES, in_ch + hs_c.back(), out_ch);
      hs_c.pop_back();
      in_ch = out_ch;
    }
    if (is_attn(res)) add(M_ATTN, in_ch, in_ch);
    if (c.progressive == 1) { add(M_GN, in_ch, in_ch); add(M_CONV3, in_ch, channels); }
    if (l != 0) add(M_RES, in_ch, in_ch, 1, 0);
  }
  if (c.progressive != 1) { add(M_GN, in_ch, in_ch); add(M_CONV3, in_ch, channels); }

  const int temb = 4 * nf;
  for (auto& m : n.mods) {
    switch (m.kind) {
      case M_FOURIER:
        n.add_param(mname(m.idx, "W"), {m.cin});
        break;
      case M_LINEAR:
        n.add_param(mname(m.idx, "weight"), {m.cout, m.cin});
        n.add_param(mname(m.idx, "bias"), {m.cout});
        break;
      case M_CONV3:
        n.add_param(mname(m.idx, "weight"), {m.cout, m.cin, 3, 3});
        n.add_param(mname(m.idx, "bias"), {m.cout});
        break;
      case M_GN:
        n.add_param(mname(m.idx, "weight"), {m.cin});
        n.add_param(mname(m.idx, "bias"), {m.cin});
        break;
      case M_COMBINE:
        n.add_param(mname(m.idx, "Conv_0.weight"), {m.cout, m.cin, 1, 1});
        n.add_param(mname(m.idx, "Conv_0.bias"), {m.cout});
        break;
      case M_ATTN:
        n.add_param(mname(m.idx, "GroupNorm_0.weight"), {m.cin});
        n.add_param(mname(m.idx, "GroupNorm_0.bias"), {m.cin});
        for (int j = 0; j < 4; ++j) {
          char w[32], b[32];
          snprintf(w, sizeof(w), "NIN_%d.W", j);
          snprintf(b, sizeof(b), "NIN_%d.b", j);
          n.add_param(mname(m.idx, w), {m.cin, m.cin});
          n.add_param(mname(m.idx, b), {m.cin});
        }
        break;
      case M_RES:
        n.add_param(mname(m.idx, "GroupNorm_0.weight"), {m.cin});
        n.add_param(mname(m.idx, "GroupNorm_0.bias"), {m.cin});
        n.add_param(mname(m.idx, "Conv_0.weight"), {m.cout, m.cin, 3, 3});
        n.add_param(mname(m.idx, "Conv_0.bias"), {m.cout});
        n.add_param(mname(m.idx, "Dense_0.weight"), {m.cout, temb});
        n.add_param(mname(m.idx, "Dense_0.bias"), {m.cout});
        n.add_param(mname(m.idx, "GroupNorm_1.weight"), {m.cout});
        n.add_param(mname(m.idx, "GroupNorm_1.bias"), {m.cout});
        n.add_param(mname(m.idx, "Conv_1.weight"), {m.cout, m.cout, 3, 3});
        n.add_param(mname(m.idx, "Conv_1.bias"), {m.cout});
        if (m.cin != m.cout || m.up || m.down) {
          n.add_param(mname(m.idx, "Conv_2.weight"), {m.cout, m.cin, 1, 1});
          n.add_param(mname(m.idx, "Conv_2.bias"), {m.cout});
        }
        break;
      default:
        break;
    }
  }
  return CSD_OK;
}

// ---- packed layout ------------------------------------------------------------------------------
static int proto_conv(ConvPlan* p, int c0, int c1, int cout, int taps) {
  memset(p, 0, sizeof(*p));
  p->B = 1; p->IH = p->IW = p->OH = p->OW = 32;   // placeholder geometry: only channel fields matter
  p->C0 = c0; p->C1 = c1; p->Cout = cout; p->taps = taps;
  p->stride = 1; p->pad = taps == 9 ? 1 : 0; p->up = 0;
  return conv_plan_tiles(p);
}

static int build_packed_layout(Net& n) {
  size_t off = 0;
  auto take = [&](size_t nfl) { size_t o = off; off += (nfl + 63) / 64 * 64; return o; };
  auto add_copy = [&](const std::string& pname) {
    const int pi = n.P(pname);
    Net::Copy cpy{pi, take((size_t)n.params[pi].numel)};
    n.copies.push_back(cpy);
    n.copy_off[pname] = cpy.off;
  };
  const int net_ns = precision_ns(n.cfg.precision);
  // the 6 -> nf stem: with the input padded to 16 channels (zeros) it runs on the fp16 matrix cores like every other 3x3 (one K step
  // per tap) instead of the fp32 ones (36 MFMAs of 64 cycles per 32 pixels: 380 us per evaluation at 160^2, B = 64, MFMA-bound)
  if (n.cfg.arch == 0 && net_ns && !CSD_TUNE_ENV("CSD_STEM_F32")) n.in_cpad = 16;
  // quad schedule: measured faster than the loader/consumer one in the split (3-MFMA) mode only
  const bool use_q = net_ns == 2 && !CSD_TUNE_ENV("CSD_NO_Q");
  int cur_res = n.cfg.image_size;    // resolution of the layer being laid out
  // opt-in experiment (measured slower: one loader wave cannot convert a patch as fast as three waves consume it)
  const bool fused_norm = CSD_TUNE_ENV("CSD_FUSED_NORM") != nullptr;
  auto add_conv = [&](const std::string& key, int c0, int c1, int cout, int taps,
                      std::vector<PackedConv::Src> srcs, bool stride1 = true, bool normed = false,
                      bool resample = false, bool upsample = false, bool last = false) -> int {
    PackedConv pc;
    int rc;
    if (last && net_ns && taps == 9 && c1 == 0 && srcs.size() == 1 && srcs[0].layout == 0 && srcs[0].cout_off == 0 &&
        (srcs[0].cin_src <= 0 || srcs[0].cin_src == c0) && pw16_taps_supported(c0, cout, net_ns)) {
      // the network's last layer (nf -> 3 channels): one pointwise contraction to 27 tap-partial channels + a gather (conv_pw16.hip)
      if ((rc = proto_conv(&pc.proto, c0, 0, pw16_taps_cout(cout), 1))) return rc;
      pc.ns = net_ns;
      pc.pw = true;
      pc.tap_cout = cout;
      pc.w_off = take(pw16_packed_bytes(pc.proto, pc.ns) / sizeof(float) + 1);
      pc.b_off = take((size_t)pc.proto.CoutPad + 96);
      pc.srcs = srcs;
      n.pconv_by_name[key] = (int)n.pconvs.size();
      n.pconvs.push_back(pc);
      return CSD_OK;
    }
    rc = proto_conv(&pc.proto, c0, c1, cout, taps);
    if (rc) return rc;
    // the 5 x 5 level on the quad schedule: 32-cout groups (conv_f16_q.hip: q_ntq)
    static const int nt1_res = CSD_TUNE_ENV("CSD_Q_NT1_RES") ? atoi(CSD_TUNE_ENV("CSD_Q_NT1_RES")) : 5;      // tuning aid (0 disables)
    if (net_ns == 2 && normed && stride1 && !resample && cur_res <= nt1_res) pc.proto.qnt = 1;
    ConvPlan one = pc.proto;       // a GroupNorm-ed conv reads ONE fp16 tensor of c0 + c1 channels
    one.C0 = c0 + c1; one.C1 = 0;
    // high-resolution GroupNorm-ed convs run the loader/consumer schedule with the norm fused into its loader
    // (standard fragment layout); the quad schedule serves the lower levels, whose tiles straddle samples
    // (fp16 mode: the loader/consumer schedule wins on the GroupNorm-ed convs, the quad schedule on the resampling ones,
    // which would otherwise convert fp32 -> fp16 inside the old kernel's staging loop)
    const bool q_here = use_q || (net_ns == 1 && resample && !CSD_TUNE_ENV("CSD_NO_Q"));
    // fused-prologue schedule: GroupNorm-ed stride-1 convs whose 16 x 16 tiles lie inside one sample (reads the fp32 residual
    // stream itself: the gn_apply16 pass and its fp16 planes disappear)
    ConvPlan ffp = pc.proto;
    ffp.IH = ffp.IW = ffp.OH = ffp.OW = cur_res;
    // (NCSN++ up / down blocks: Conv_0 reads FIR(act(GroupNorm(x))) - an fp32 tensor that is already normalised and activated, so the same
    // kernel takes it with its NORM = false prologue (plain split; the values stay far inside e4m3): no split pass, no fp16 planes)
    const bool fir_act_input = resample && !upsample && normed && n.cfg.arch == 1;
    const int ff_ns = n.cfg.precision == CSD_PREC_F16F8 ? 3 : net_ns;      // 3: fp16 hi*hi + fp8 corrections (conv_ff.hip)
    if (net_ns && normed && stride1 && (!resample || fir_act_input) && n.cfg.act == CSD_ACT_SWISH && convff_supported(ffp, ff_ns)) {
      pc.ns = ff_ns;
      pc.ff = true;
      pc.proto.KC = 16;
      pc.w_off = take(convff_packed_bytes(pc.proto, pc.ns) / sizeof(float) + 1);
    } else if (q_here && normed && stride1 && !(fused_norm && cur_res >= 64) && conv16q_supported(one, net_ns)) {
      // 3: fp16 hi*hi + fp8 corrections - for GroupNorm-ed operands only: e4m3 saturates at 448, which activated, normalised values never
      // reach, while the raw residual stream that the resampling convs read does at the large-sigma end of a sampling run (measured: the
      // 1000-step trajectory error went from 6e-7 to 1e-4 with them included)
      pc.ns = (n.cfg.precision == CSD_PREC_F16F8 && net_ns == 2 && !resample && conv16q_supported(one, 3)) ? 3 : net_ns;
      pc.q = true;
      pc.proto.KC = 16;
      pc.up4 = upsample && srcs.size() == 1 && conv16q_up4_supported(one, pc.ns);
      pc.w_off = take((pc.up4 ? conv16q_up4_packed_bytes(one, pc.ns) : conv16q_packed_bytes(one, pc.ns)) / sizeof(float) + 1);
    } else if (net_ns && stride1 && conv16_supported(pc.proto)) {
      pc.ns = net_ns;
      if ((rc = conv16_plan_tiles(&pc.proto, pc.ns))) return rc;
      pc.w_off = take(conv16_packed_bytes(pc.proto, pc.ns) / sizeof(float) + 1);
    } else if (net_ns && stride1 && pw16_supported(pc.proto, net_ns)) {
      bool aligned = true;
      for (const auto& sr : srcs) aligned = aligned && (sr.cout_off % 16 == 0);
      if (aligned) {
        pc.ns = net_ns;
        pc.pw = true;
        pc.w_off = take(pw16_packed_bytes(pc.proto, pc.ns) / sizeof(float) + 1);
      } else {
        pc.w_off = take(conv_packed_floats(pc.proto));
      }
    } else {
      pc.w_off = take(conv_packed_floats(pc.proto));
    }
    pc.b_off = take((size_t)pc.proto.CoutPad + 96);
    pc.srcs = srcs;
    n.pconv_by_name[key] = (int)n.pconvs.size();
    n.pconvs.push_back(pc);
    return CSD_OK;
  };
  const csd_unet_config& c = n.cfg;
  // skip-connection split: replay the hs_c stack to know (C0, C1) of every up-path res block
  std::vector<int> hs_c{c.nf};
  int in_ch = c.nf;
  size_t mi = 0;
  auto next_mod = [&]() -> Module& { return n.mods[mi++]; };
  if (c.arch == 0 && c.conditional) {
    for (int j = 0; j < 2; ++j) {
      Module& m = next_mod();
      add_copy(mname(m.idx, "weight"));
      add_copy(mname(m.idx, "bias"));
    }
  }
  // the first layer of both families in the fp16 modes: input assembly + 3x3 conv + the next GroupNorm's partials as one launch (stem.hip)
  auto stem_layout = [&](Module& m) -> bool {
    if (!(net_ns && m.cin == c.x_channels + c.y_channels && stem_supported(c.x_channels, c.y_channels, m.cout, c.image_size, net_ns)))
      return false;
    PackedConv pc;
    if (proto_conv(&pc.proto, n.in_cpad, 0, m.cout, 9)) return false;
    pc.ns = net_ns;
    pc.stem = true;
    pc.w_off = take(stem_packed_bytes(m.cout, net_ns) / sizeof(float) + 1);
    pc.b_off = take((size_t)pc.proto.CoutPad + 96);
    pc.srcs = {{n.P(mname(m.idx, "weight")), n.P(mname(m.idx, "bias")), 0, m.cout, 0, m.cin}};
    n.pconv_by_name[std::to_string(m.idx)] = (int)n.pconvs.size();
    n.pconvs.push_back(pc);
    return true;
  };
  int rc;
  if (c.arch == 0) {
    Module& m = next_mod();   // stem: Cin padded to 8 (zero weights for the padding channels)
    if (!stem_layout(m)) {
      rc = add_conv(std::to_string(m.idx), n.in_cpad, 0, m.cout, 9,
                    {{n.P(mname(m.idx, "weight")), n.P(mname(m.idx, "bias")), 0, m.cout, 0, m.cin}});
      if (rc) return rc;
    }
  }
  auto res_layout = [&](Module& m, int c0, int c1) -> int {
    const std::string k = std::to_string(m.idx);
    add_copy(mname(m.idx, "GroupNorm_0.weight"));
    add_copy(mname(m.idx, "GroupNorm_0.bias"));
    int r = add_conv(k + ".Conv_0", c0, c1, m.cout, 9,
                     {{n.P(mname(m.idx, "Conv_0.weight")), n.P(mname(m.idx, "Conv_0.bias")), 0, m.cout, 0}}, true, true);
    if (r) return r;
    add_copy(mname(m.idx, "GroupNorm_1.weight"));
    add_copy(mname(m.idx, "GroupNorm_1.bias"));
    r = add_conv(k + ".Conv_1", m.cout, 0, m.cout, 9,
                 {{n.P(mname(m.idx, "Conv_1.weight")), n.P(mname(m.idx, "Conv_1.bias")), 0, m.cout, 0}}, true, true);
    if (r) return r;
    if (m.cin != m.cout) {
      r = add_conv(k + ".NIN_0", c0, c1, m.cout, 1,
                   {{n.P(mname(m.idx, "NIN_0.W")), n.P(mname(m.idx, "NIN_0.b")), 1, m.cout, 0}});
      if (r) return r;
    }
    if (c.conditional) {
      n.dense_col[m.idx] = n.dense_total;
      n.dense_total += m.cout;
    }
    return CSD_OK;
  };
  auto attn_layout = [&](Module& m) -> int {
    const std::string k = std::to_string(m.idx);
    const int C = m.cin;
    add_copy(mname(m.idx, "GroupNorm_0.weight"));
    add_copy(mname(m.idx, "GroupNorm_0.bias"));
    int r = add_conv(k + ".qkv", C, 0, 3 * C, 1,
                     {{n.P(mname(m.idx, "NIN_0.W")), n.P(mname(m.idx, "NIN_0.b")), 1, C, 0},
                      {n.P(mname(m.idx, "NIN_1.W")), n.P(mname(m.idx, "NIN_1.b")), 1, C, C},
                      {n.P(mname(m.idx, "NIN_2.W")), n.P(mname(m.idx, "NIN_2.b")), 1, C, 2 * C}});
    if (r) return r;
    return add_conv(k + ".NIN_3", C, 0, C, 1,
                    {{n.P(mname(m.idx, "NIN_3.W")), n.P(mname(m.idx, "NIN_3.b")), 1, C, 0}});
  };
  auto resample_layout = [&](Module& m) -> int {
    if (!c.resamp_with_conv) return CSD_OK;
    return add_conv(std::to_string(m.idx) + ".Conv_0", m.cin, 0, m.cin, 9,
                    {{n.P(mname(m.idx, "Conv_0.weight")), n.P(mname(m.idx, "Conv_0.bias")), 0, m.cin, 0}},
                    /*stride1=*/true,    // (the fp16 kernel also covers the stride-2 Downsample)
                    /*quad-eligible=*/true, /*resample=*/true,    // plain fp16 split of the source tensor + quad schedule (x2 addressing / stride 2)
                    /*upsample=*/m.kind == M_UP);
  };
  auto is_attn = [&](int res) {
    for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
    return false;
  };
  if (c.arch == 1) {
    // ---- NCSN++ (module list of build_modules_ncsnpp): same packed-conv machinery, BigGAN blocks ----
    auto res_layout_pp = [&](Module& m, int c0, int c1) -> int {
      const std::string k = std::to_string(m.idx);
      const bool plain = !m.up && !m.down;
      add_copy(mname(m.idx, "GroupNorm_0.weight"));
      add_copy(mname(m.idx, "GroupNorm_0.bias"));
      // Conv_0 reads the GroupNorm-ed input directly only in plain blocks; up/down blocks feed it the FIR-resampled
      // activation (an fp32 tensor without a norm)
      int r = add_conv(k + ".Conv_0", c0, c1, m.cout, 9,
                       {{n.P(mname(m.idx, "Conv_0.weight")), n.P(mname(m.idx, "Conv_0.bias")), 0, m.cout, 0}}, true,
                       /*quad-eligible (plain split of the resampled tensor in up/down blocks)=*/true, /*resample=*/!plain);
      if (r) return r;
      add_copy(mname(m.idx, "GroupNorm_1.weight"));
      add_copy(mname(m.idx, "GroupNorm_1.bias"));
      r = add_conv(k + ".Conv_1", m.cout, 0, m.cout, 9,
                   {{n.P(mname(m.idx, "Conv_1.weight")), n.P(mname(m.idx, "Conv_1.bias")), 0, m.cout, 0}}, true, true);
      if (r) return r;
      if (m.cin != m.cout || m.up || m.down) {
        r = add_conv(k + ".Conv_2", c0, c1, m.cout, 1,
                     {{n.P(mname(m.idx, "Conv_2.weight")), n.P(mname(m.idx, "Conv_2.bias")), 0, m.cout, 0}});
        if (r) return r;
      }
      n.dense_col[m.idx] = n.dense_total;
      n.dense_total += m.cout;
      return CSD_OK;
    };
    auto conv3_layout = [&](Module& m, int cin_pad, bool normed, bool last = false) -> int {
      return add_conv(std::to_string(m.idx), cin_pad, 0, m.cout, 9,
                      {{n.P(mname(m.idx, "weight")), n.P(mname(m.idx, "bias")), 0, m.cout, 0, m.cin}}, true, normed, false, false, last);
    };
    auto gn_layout = [&](Module& m) {
      add_copy(mname(m.idx, "weight"));
      add_copy(mname(m.idx, "bias"));
    };
    size_t mj = 0;
    auto nextm = [&]() -> Module& { return n.mods[mj++]; };
    if (c.embedding_type == 1) add_copy(mname(nextm().idx, "W"));
    for (int j = 0; j < 2; ++j) {
      Module& m = nextm();
      add_copy(mname(m.idx, "weight"));
      add_copy(mname(m.idx, "bias"));
    }
    {
      Module& ms = nextm();                                              // stem: Cin padded to 8
      if (!stem_layout(ms) && (rc = conv3_layout(ms, n.in_cpad, false))) return rc;
    }
    std::vector<int> hc{c.nf};
    int ich = c.nf;
    for (int l = 0; l < c.n_levels; ++l) {
      const int res = c.image_size >> l;
      cur_res = res;
      for (int b = 0; b < c.num_res_blocks; ++b) {
        Module& m = nextm();
        if ((rc = res_layout_pp(m, ich, 0))) return rc;
        ich = m.cout;
        if (is_attn(res)) { if ((rc = attn_layout(nextm()))) return rc; }
        hc.push_back(ich);
      }
      if (l != c.n_levels - 1) {
        cur_res = res >> 1;
        if ((rc = res_layout_pp(nextm(), ich, 0))) return rc;
        if (c.progressive_input == 1) {
          Module& m = nextm();      // Combine: 1x1 conv of the (8-channel padded) input pyramid
          rc = add_conv(std::to_string(m.idx) + ".Conv_0", n.in_cpad, 0, m.cout, 1,
                        {{n.P(mname(m.idx, "Conv_0.weight")), n.P(mname(m.idx, "Conv_0.bias")), 0, m.cout, 0, m.cin}});
          if (rc) return rc;
        }
        hc.push_back(ich);
      }
    }
    if ((rc = res_layout_pp(nextm(), ich, 0))) return rc;
    if ((rc = attn_layout(nextm()))) return rc;
    if ((rc = res_layout_pp(nextm(), ich, 0))) return rc;
    for (int l = c.n_levels - 1; l >= 0; --l) {
      const int res = c.image_size >> l;
      cur_res = res;
      for (int b = 0; b < c.num_res_blocks + 1; ++b) {
        Module& m = nextm();
        const int skip = hc.back();
        hc.pop_back();
        if ((rc = res_layout_pp(m, ich, skip))) return rc;
        ich = m.cout;
      }
      if (is_attn(res)) { if ((rc = attn_layout(nextm()))) return rc; }
      if (c.progressive == 1) {                  // (the output pyramid's convs: a handful of couts each, like the last layer)
        gn_layout(nextm());
        if ((rc = conv3_layout(nextm(), ich, true, /*last=*/true))) return rc;
      }
      if (l != 0) {
        cur_res = res << 1;
        if ((rc = res_layout_pp(nextm(), ich, 0))) return rc;
      }
    }
    if (c.progressive != 1) {
      gn_layout(nextm());
      if ((rc = conv3_layout(nextm(), ich, true, /*last=*/true))) return rc;
    }
    CSD_REQUIRE(mj == n.mods.size(), "ncsnpp: internal module walk mismatch");
    n.dense_all_off = take((size_t)n.dense_total * 4 * c.nf);
    n.dense_all_bias_off = take((size_t)n.dense_total);
    n.packed_floats = off;
    return CSD_OK;
  }
  for (int l = 0; l < c.n_levels; ++l) {
    const int res = c.image_size >> l;
    cur_res = res;
    for (int b = 0; b < c.num_res_blocks; ++b) {
      Module& m = next_mod();
      if ((rc = res_layout(m, in_ch, 0))) return rc;
      in_ch = m.cout;
      if (is_attn(res)) { if ((rc = attn_layout(next_mod()))) return rc; }
      hs_c.push_back(in_ch);
    }
    if (l != c.n_levels - 1) { if ((rc = resample_layout(next_mod()))) return rc; hs_c.push_back(in_ch); }
  }
  if ((rc = res_layout(next_mod(), in_ch, 0))) return rc;
  if ((rc = attn_layout(next_mod()))) return rc;
  if ((rc = res_layout(next_mod(), in_ch, 0))) return rc;
  for (int l = c.n_levels - 1; l >= 0; --l) {
    const int res = c.image_size >> l;
    cur_res = res;
    for (int b = 0; b < c.num_res_blocks + 1; ++b) {
      Module& m = next_mod();
      const int skip = hs_c.back();
      hs_c.pop_back();
      if ((rc = res_layout(m, in_ch, skip))) return rc;
      in_ch = m.cout;
    }
    if (is_attn(res)) { if ((rc = attn_layout(next_mod()))) return rc; }
    if (l != 0) { if ((rc = resample_layout(next_mod()))) return rc; }
  }
  {
    Module& m = next_mod();
    add_copy(mname(m.idx, "weight"));
    add_copy(mname(m.idx, "bias"));
  }
  {
    Module& m = next_mod();
    rc = add_conv(std::to_string(m.idx), m.cin, 0, m.cout, 9,
                  {{n.P(mname(m.idx, "weight")), n.P(mname(m.idx, "bias")), 0, m.cout, 0}}, true, false, false, false, /*last=*/true);
    if (rc) return rc;
  }
  CSD_REQUIRE(mi == n.mods.size(), "unet: internal module walk mismatch");
  if (c.conditional) {
    n.dense_all_off = take((size_t)n.dense_total * 4 * c.nf);
    n.dense_all_bias_off = take((size_t)n.dense_total);
  }
  n.packed_floats = off;
  return CSD_OK;
}

// ---- plan for batch B ---------------------------------------------------------------------------
struct Builder {
  Net& n;
  Plan& pl;
  Arena ar;
  int B;
  size_t gn_partial = NONE, nscale = NONE, nshift = NONE;   // shared scratch
  size_t dense_all = NONE;
  float pending_scale = 1.f;          // out_scale of the NEXT conv() (NCSN++ skip_rescale: (x + h)/sqrt(2))
  size_t next_out = NONE;             // the NEXT conv() writes here instead of allocating (a batch chunk's slice of the region's output tensor)
  int rc = CSD_OK;
  // tensors whose producing conv left per-tile GroupNorm partials behind: workspace offset -> (partials, tiles per sample)
  struct TileStats { size_t off; int tpi; };
  std::map<size_t, TileStats> tile_stats;

  Builder(Net& n_, Plan& p_, int B_) : n(n_), pl(p_), B(B_) {}

  // every workspace allocation goes through here: a new tensor at an offset invalidates what was known about the old one
  size_t alloc_(size_t nfloats) {
    const size_t off = ar.alloc(nfloats);
    tile_stats.erase(off);
    return off;
  }

  void count(double flops, double bytes) {
    pl.flops += flops; pl.bytes += bytes; pl.launches += 1;
    pl.ops.back().flops = flops; pl.ops.back().bytes = bytes;
  }

  // GroupNorm statistics of (src0|src1) -> nscale/nshift
  void gn(size_t src0, size_t src1, int c0, int c1, int hw, const std::string& gname, const std::string& bname) {
    const int G = n.cfg.arch == 1 ? ncsnpp_groups(c0 + c1) : 32;
    if (src1 != NONE && (tile_stats.count(src0) != 0) != (tile_stats.count(src1) != 0)) {
      // a concatenated input of which ONE half carries epilogue partials (the skip tensor next to an Upsample output): the streaming pass
      // reads only the other half and leaves per-channel partials in the same layout (it used to re-read both: 1.26 GB at 160^2 x 192)
      const bool first = tile_stats.count(src0) == 0;
      const size_t src = first ? src0 : src1;
      const int cs = first ? c0 : c1;
      Op s;
      s.kind = OP_GN_STATS;
      if (gn_plan(&s.gp, B, hw, cs, 0, 1)) { rc = CSD_ERR_INVALID; return; }
      s.a = src; s.b = NONE; s.i0 = 1;      // per-channel partials
      s.out = alloc_((size_t)B * s.gp.nchunk * cs * 2 * 2);      // doubles; never released (small)
      s.cls = CSD_PROF_GN_STATS; s.bytes = (double)B * hw * cs * 4;
      pl.ops.push_back(s);
      pl.launches += 1;
      tile_stats[src] = TileStats{s.out, s.gp.nchunk};
    }
    const auto t0 = tile_stats.find(src0);
    const auto t1 = src1 == NONE ? tile_stats.end() : tile_stats.find(src1);
    if (t0 != tile_stats.end() && (src1 == NONE || t1 != tile_stats.end())) {
      // both sources were written by convs that accumulated the statistics in their epilogue: no pass over the tensor
      Op f;
      f.kind = OP_GN_FINAL_TILES;
      f.a = t0->second.off; f.i0 = t0->second.tpi; f.i1 = c0;
      f.b = src1 == NONE ? NONE : t1->second.off; f.i2 = src1 == NONE ? 0 : t1->second.tpi; f.i3 = c1;
      f.i4 = hw;
      f.gp.G = G;
      f.pk0 = n.copy_off.at(gname); f.pk1 = n.copy_off.at(bname);
      f.out = nscale; f.c = nshift;
      f.cls = CSD_PROF_GN_FINAL;
      pl.ops.push_back(f);
      pl.launches += 1;
      pl.bytes += 2.0 * B * hw * (c0 + c1) * 4;   // SURVEY 8(d) algorithmic bytes are those of the unfused op
      return;
    }
    Op s;
    s.kind = OP_GN_STATS;
    if (gn_plan(&s.gp, B, hw, c0, c1, G)) { rc = CSD_ERR_INVALID; return; }
    s.a = src0; s.b = src1; s.out = gn_partial;
    s.cls = CSD_PROF_GN_STATS; s.bytes = (double)B * hw * (c0 + c1) * 4;
    pl.ops.push_back(s);
    Op f;
    f.kind = OP_GN_FINAL;
    f.gp = s.gp;
    f.a = gn_partial; f.pk0 = n.copy_off.at(gname); f.pk1 = n.copy_off.at(bname);
    f.out = nscale; f.b = nshift;
    f.cls = CSD_PROF_GN_FINAL;
    pl.ops.push_back(f);
    pl.launches += 2;
    pl.bytes += 2.0 * B * hw * (c0 + c1) * 4;   // GroupNorm reads + writes its tensor
  }

  // convolution; returns output offset (allocated here unless external)
  size_t conv(const std::string& key, size_t src0, size_t src1, int ih, int iw, int stride, int pad, int up,
              bool norm, int act, size_t res, size_t temb_col, bool external_nchw, int real_cin = -1) {
    const PackedConv& pc = n.pconvs[n.pconv_by_name.at(key)];
    if (pc.tap_cout) {
      // tap-partial form (conv_pw16.hip): pointwise contraction of act(GN(x)) to 9 * cout partial channels, then the 9-tap gather
      if (!norm || src1 != NONE || stride != 1 || up || pad != 1 || temb_col != NONE) {
        set_error("tap-partial conv on an unsupported layer"); rc = CSD_ERR_INVALID; return NONE;
      }
      Op o;
      o.kind = OP_CONV;
      o.cp = pc.proto;
      o.cp.B = B; o.cp.IH = o.cp.OH = ih; o.cp.IW = o.cp.OW = iw;
      o.cp.stride = 1; o.cp.pad = 0; o.cp.up = 0;
      o.i4 = pc.ns; o.i2 = 1;
      o.a = src0; o.b = NONE; o.pk0 = pc.w_off; o.pk1 = NONE;          // (the bias joins in the gather)
      o.d = nscale; o.e = nshift;
      o.act = act;
      o.temb_base = dense_all; o.temb_stride = n.dense_total;
      const size_t part = alloc_((size_t)B * ih * iw * o.cp.Cout);
      o.out = part;
      o.cls = CSD_PROF_CONV1X1;                  // (profiler classes follow the kernels: a pointwise contraction + a small gather)
      pl.ops.push_back(o);
      const size_t out_elems = (size_t)B * ih * iw * pc.tap_cout;
      count(2.0 * out_elems * o.cp.C0 * 9, ((double)B * ih * iw * o.cp.C0 + (double)out_elems) * 4);
      pl.ops.back().bytes = (double)B * ih * iw * (o.cp.C0 + o.cp.Cout) * 4.0;
      Op g;
      g.kind = OP_TAPSUM;
      g.a = part; g.pk1 = pc.b_off; g.c = res;
      g.i0 = ih; g.i1 = iw; g.i2 = pc.tap_cout;
      g.fscale = pending_scale;
      pending_scale = 1.f;
      g.out_external = external_nchw ? 1 : 0;
      g.out = external_nchw ? NONE : alloc_(out_elems);
      g.cls = CSD_PROF_OTHER;
      g.bytes = (double)B * ih * iw * (o.cp.Cout + pc.tap_cout) * 4.0;
      pl.ops.push_back(g);
      pl.launches += 1;
      ar.release(part);
      return g.out;
    }
    Op o;
    o.kind = OP_CONV;
    o.cp = pc.proto;
    o.cp.B = B; o.cp.IH = ih; o.cp.IW = iw;
    o.cp.stride = stride; o.cp.pad = pad; o.cp.up = up;
    o.cp.OH = (ih << up) / stride; o.cp.OW = (iw << up) / stride;
    o.i4 = pc.ns;
    o.i2 = pc.ff ? 3 : (pc.pw ? 1 : (pc.q ? 2 : 0));
    const int kcs = (pc.ns && !pc.pw && norm) ? conv16_kcs(pc.ns, o.cp.C0 + o.cp.C1) : 1;    // fp16-source convs stage in bursts
    bool fused = false;     // GroupNorm affine + activation applied by the conv's loader wave (no gn_apply16 pass)
    if (pc.ns && !pc.pw && !pc.q && norm && stride == 1 && !up && o.cp.C0 % 32 == 0 && o.cp.C1 % 32 == 0 &&
        !CSD_TUNE_ENV("CSD_NO_LC") && CSD_TUNE_ENV("CSD_FUSED_NORM")) {
      ConvPlan trial = o.cp;
      if (conv16_plan_tiles(&trial, pc.ns, 2, true) == CSD_OK && trial.LC && trial.OH % trial.TH == 0) {
        fused = true;
        o.cp = trial;
      }
    }
    if (fused) {
    } else if (pc.ff) {
      if (external_nchw || stride != 1 || up) { set_error("fused-prologue conv on an unsupported layer"); rc = CSD_ERR_INVALID; return NONE; }
      if (convff_plan_tiles(&o.cp, pc.ns)) { rc = CSD_ERR_INVALID; return NONE; }
    } else if (pc.q) {
      if (external_nchw || (!norm && o.cp.C1 != 0) || (stride == 2 && (norm || up))) { set_error("quad fp16 conv on an unsupported layer"); rc = CSD_ERR_INVALID; return NONE; }
      o.cp.C0 = o.cp.C0 + o.cp.C1; o.cp.C1 = 0;
      if (pc.up4) {
        if (!up || stride != 1) { set_error("phase-decomposed Upsample packed for a layer that is not one"); rc = CSD_ERR_INVALID; return NONE; }
        o.cp.up = 2;
      }
      if (conv16q_plan_tiles(&o.cp, pc.ns)) { rc = CSD_ERR_INVALID; return NONE; }
    } else if (pc.pw) {
      if (act != CSD_ACT_NONE || temb_col != NONE || external_nchw) { set_error("pointwise fp16 layer with act/temb/NCHW"); rc = CSD_ERR_INVALID; return NONE; }
    } else if (pc.ns ? conv16_plan_tiles(&o.cp, pc.ns, kcs, kcs > 1 && !CSD_TUNE_ENV("CSD_NO_LC")) : conv_plan_tiles(&o.cp)) { rc = CSD_ERR_INVALID; return NONE; }
    // the packed layout depends on KC only (not on NT / tile shape)
    if (o.cp.KC != pc.proto.KC) { set_error("conv plan/pack mismatch"); rc = CSD_ERR_INVALID; return NONE; }
    o.a = src0; o.b = src1; o.pk0 = pc.w_off; o.pk1 = pc.b_off;
    o.c = res;
    o.d = norm ? nscale : NONE;
    o.e = norm ? nshift : NONE;
    size_t hi16 = NONE, lo16 = NONE;
    // a quad-schedule conv WITHOUT a GroupNorm in front (Downsample / Upsample, the FIR-resampled Conv_0 of NCSN++) in the full split reads
    // its fp32 source directly: the split happens in the kernel's staging burst (no plane write / read)
    const bool q_raw = pc.q && !norm && pc.ns == 2 && o.cp.C1 == 0 && (o.cp.NT == 3 || o.cp.NT == 4) && !CSD_TUNE_ENV("CSD_NO_Q_RAW");
    if (q_raw) {
      o.i3 = 2;
    } else if (pc.ns && !pc.pw && !pc.ff && (norm || pc.q) && !fused) {
      // fp16 kernel: normalise + activate + split ONCE per element into fp16 planes, conv copies them
      // (a quad-schedule conv without a GroupNorm - Upsample, the FIR-resampled Conv_0 of NCSN++ - gets a plain split)
      const size_t nh = ((size_t)B * ih * iw * (o.cp.C0 + o.cp.C1) + 1) / 2;      // halves -> floats
      Op ap;
      ap.kind = OP_GN_APPLY16;
      ap.a = src0; ap.b = src1; ap.i0 = pc.proto.C0; ap.i1 = pc.proto.C1; ap.i2 = ih * iw;
      ap.d = norm ? nscale : NONE; ap.e = norm ? nshift : NONE; ap.act = norm ? act : (int)CSD_ACT_NONE;
      hi16 = alloc_(nh);
      if (pc.ns >= 2) lo16 = alloc_(nh);
      ap.out = hi16; ap.c = lo16;
      // small maps: the GroupNorm's statistics + finalize launches (the two ops gn() has just pushed) and this pass become ONE launch
      const size_t nops = pl.ops.size();
      if (norm && nops >= 2 && pl.ops[nops - 2].kind == OP_GN_STATS && pl.ops[nops - 1].kind == OP_GN_FINAL &&
          pl.ops[nops - 2].a == src0 && pl.ops[nops - 2].b == src1 && !CSD_TUNE_ENV("CSD_NO_GN_FUSED") &&
          gn_fused16_groups(ih * iw, pc.proto.C0, pc.proto.C1, pl.ops[nops - 2].gp.G) > 0) {
        ap.kind = OP_GN_FUSED16;
        ap.gp = pl.ops[nops - 2].gp;
        ap.pk0 = pl.ops[nops - 1].pk0; ap.pk1 = pl.ops[nops - 1].pk1;      // gamma, beta
        ap.d = NONE; ap.e = NONE;
        pl.ops.pop_back();
        pl.ops.pop_back();
        pl.launches -= 2;
      }
      ap.cls = CSD_PROF_GN_APPLY;
      ap.bytes = (double)B * ih * iw * (o.cp.C0 + o.cp.C1) * (4 + 2 * (pc.ns >= 2 ? 2 : 1));
      ap.i3 = pc.ns == 3;                       // second plane = e4m3 byte pairs
      pl.ops.push_back(ap);
      pl.launches += 1;
      o.a = hi16; o.b = lo16;
      o.cp.C0 = o.cp.C0 + o.cp.C1; o.cp.C1 = 0;
      o.d = NONE; o.e = NONE;
      o.i3 = 1;                                   // in16
    }
    o.temb_base = dense_all;
    o.fscale = pending_scale;
    pending_scale = 1.f;
    o.act = act;
    o.temb_col = temb_col;
    o.temb_stride = n.dense_total;
    o.out_external = external_nchw ? 1 : 0;
    const size_t out_elems = (size_t)B * o.cp.OH * o.cp.OW * o.cp.Cout;
    o.out = external_nchw ? NONE : (next_out != NONE ? next_out : alloc_(out_elems));
    next_out = NONE;
    const int oh_tiled = o.cp.up == 2 ? o.cp.IH : o.cp.OH;         // (the phase form tiles the SOURCE image, four workgroups per tile)
    // (pc.ff: the fused-prologue kernels tile every sample on its own - ragged tiles, where they run at all, mask their statistics)
    if (!pc.pw && !external_nchw && o.cp.taps == 9 && (pc.ff || oh_tiled % o.cp.TH == 0) && !CSD_TUNE_ENV("CSD_NO_FUSED_STATS") &&
        (o.cp.up != 2 || (conv16q_up4_stats_ok(o.cp) && !CSD_TUNE_ENV("CSD_NO_UP4_STATS")))) {
      // (phase-decomposed Upsample: fp32 partials per (tile, M half, phase) - in fp64 they cost the kernel more than the streaming pass
      // over its output that they replace - and only where its tile shape does not depend on the batch: conv16q_plan_tiles)
      // every tile lies inside one sample: the epilogue also leaves (sum, sumsq) per (tile, cout) for the next
      // GroupNorm (the fp32 kernel: per (tile, wave, cout))
      const int tpi = cdiv(oh_tiled, o.cp.TH) * o.cp.tiles_x * (pc.q ? 2 : (pc.ns ? 1 : 4)) * (o.cp.up == 2 ? 4 : 1);
      o.stats = alloc_((size_t)B * tpi * o.cp.Cout * 2 * 2);      // doubles; never released (small)
      tile_stats[o.out] = TileStats{o.stats, tpi};
    }
    o.cls = o.cp.taps == 1 ? CSD_PROF_CONV1X1 : ((stride == 1 && !up) ? CSD_PROF_CONV3X3 : CSD_PROF_CONV3X3_RESAMPLE);
    pl.ops.push_back(o);
    ar.release(hi16);      // (plan-time lifetimes: the planes die right after this conv)
    ar.release(lo16);
    const int cin = real_cin > 0 ? real_cin : (o.cp.C0 + o.cp.C1);
    count(2.0 * out_elems * cin * o.cp.taps, ((double)B * ih * iw * cin + (double)out_elems) * 4);
    // the profiler's per-launch bytes are what THIS kernel has to move (fp16 planes in, fp32 out, residual in);
    // the plan totals above stay the SURVEY 8(d) algorithmic figures
    pl.ops.back().abytes = pl.ops.back().bytes;
    pl.ops.back().bytes = (double)B * ih * iw * (o.cp.C0 + o.cp.C1) * (o.i3 ? 2.0 * pc.ns : 4.0) +
                          (double)out_elems * 4 * (res != NONE ? 2 : 1);
    return o.out;
  }

  // the fused first layer (stem.hip): x, y (+ sigma z) in the caller's NCHW -> nf channels NHWC + tile partials for the next GroupNorm
  size_t stem(const PackedConv& pc, const Module& m, int S) {
    const csd_unet_config& c = n.cfg;
    Op o;
    o.kind = OP_STEM;
    o.pk0 = pc.w_off; o.pk1 = pc.b_off;
    o.i0 = m.cout; o.i4 = pc.ns;
    const size_t out_elems = (size_t)B * S * S * m.cout;
    o.out = alloc_(out_elems);
    const int tpi = stem_tiles_per_image(S);
    o.stats = alloc_((size_t)B * tpi * m.cout * 2 * 2);      // doubles; never released (small)
    tile_stats[o.out] = TileStats{o.stats, tpi};
    o.cls = CSD_PROF_CONV3X3;
    pl.ops.push_back(o);
    const int cin = c.x_channels + c.y_channels;
    count(2.0 * out_elems * cin * 9, ((double)B * S * S * cin + (double)out_elems) * 4);
    return o.out;
  }

  size_t res_block(const Module& m, size_t x0, size_t x1, int c0, int c1, int hw_side, size_t out_to = NONE) {
    const std::string k = std::to_string(m.idx);
    const int hw = hw_side * hw_side;
    // the shortcut contraction first; with CSD_SIDE_STREAM=1 on the side stream (it only depends on the block's input; Conv_1 joins it).
    // Opt-in: -0.35 ms per PC step in a same-box A/B, but the per-launch durations of the overlapped 3x3 launches (what the bench's
    // roofline object and the rocprofv3 summaries report) then include the time they share the CUs with it
    static const bool side_on = CSD_TUNE_ENV("CSD_SIDE_STREAM") != nullptr && atoi(CSD_TUNE_ENV("CSD_SIDE_STREAM")) != 0;
    size_t shortcut = x0, sc_buf = NONE;
    if (m.cin != m.cout) {
      const size_t op0 = pl.ops.size();
      sc_buf = conv(k + (n.cfg.arch == 1 ? ".Conv_2" : ".NIN_0"), x0, x1, hw_side, hw_side, 1, 0, 0, false, 0, NONE, NONE, false);
      shortcut = sc_buf;
      if (side_on)
        for (size_t i = op0; i < pl.ops.size(); ++i) pl.ops[i].side = 1;
    } else if (x1 != NONE) {
      set_error("res block %d: identity shortcut on a concatenated input is not supported", m.idx);
      rc = CSD_ERR_INVALID;
    }
    gn(x0, x1, c0, c1, hw, mname(m.idx, "GroupNorm_0.weight"), mname(m.idx, "GroupNorm_0.bias"));
    const size_t tcol = n.cfg.conditional ? (size_t)n.dense_col.at(m.idx) : NONE;
    const size_t h1 = conv(k + ".Conv_0", x0, x1, hw_side, hw_side, 1, 1, 0, true, n.cfg.act, NONE, tcol, false);
    gn(h1, NONE, m.cout, 0, hw, mname(m.idx, "GroupNorm_1.weight"), mname(m.idx, "GroupNorm_1.bias"));
    pending_scale = skip_scale();
    next_out = out_to;
    const size_t out = conv(k + ".Conv_1", h1, NONE, hw_side, hw_side, 1, 1, 0, true, n.cfg.act, shortcut, NONE, false);
    if (side_on && sc_buf != NONE && rc == CSD_OK && out != NONE) pl.ops.back().side = 2;
    ar.release(h1);
    ar.release(sc_buf);
    return out;
  }

  float skip_scale() const { return (n.cfg.arch == 1 && n.cfg.skip_rescale) ? 0.70710678118654752440f : 1.f; }

  // ---- NCSN++ pieces (models/layerspp.py, models/up_or_down_sampling.py) ----
  // FIR resampling of an fp32 NHWC tensor: upsample_2d / downsample_2d with the config's 4-tap kernel
  size_t fir(size_t src, int side, int C, bool up) {
    Op o;
    o.kind = OP_FIR;
    o.a = src; o.i0 = side; o.i1 = C; o.i2 = up ? 1 : 0;
    const int os = up ? side * 2 : side / 2;
    o.out = alloc_((size_t)B * os * os * C);
    o.cls = CSD_PROF_OTHER;
    pl.ops.push_back(o);
    pl.launches += 1;
    return o.out;
  }

  // act(GroupNorm(x)) materialised in fp32 (the up/down blocks resample it before Conv_0)
  size_t gn_apply32(size_t src, int C, int hw, int act) {
    Op o;
    o.kind = OP_GN_APPLY32;
    o.a = src; o.d = nscale; o.e = nshift; o.i0 = C; o.i1 = hw; o.act = act;
    o.out = alloc_((size_t)B * hw * C);
    o.cls = CSD_PROF_GN_APPLY;
    o.bytes = (double)B * hw * C * 8;
    pl.ops.push_back(o);
    pl.launches += 1;
    return o.out;
  }

  // ResnetBlockBigGANpp with up / down (layerspp.py:242-274): GN0+act -> FIR(h), FIR(x) -> Conv_0 (+temb) -> GN1+act
  // -> Conv_1 + Conv_2(x') ; (x + h)/sqrt(2)
  size_t res_block_updown(const Module& m, size_t x, int side) {
    const std::string k = std::to_string(m.idx);
    const int hw = side * side, os = m.up ? side * 2 : side / 2;
    gn(x, NONE, m.cin, 0, hw, mname(m.idx, "GroupNorm_0.weight"), mname(m.idx, "GroupNorm_0.bias"));
    size_t hr, xr;
    if (m.cin % 4 == 0) {      // FIR(act(GroupNorm(x))) and FIR(x) in one pass over x (elementwise.hip: fir_resample2)
      Op o;
      o.kind = OP_FIR2;
      o.a = x; o.d = nscale; o.e = nshift; o.i0 = side; o.i1 = m.cin; o.i2 = m.up ? 1 : 0; o.act = n.cfg.act;
      o.out = alloc_((size_t)B * os * os * m.cin);       // FIR(h)
      o.c = alloc_((size_t)B * os * os * m.cin);         // FIR(x)
      o.cls = CSD_PROF_OTHER;
      o.bytes = (double)B * (hw + 2.0 * os * os) * m.cin * 4;
      pl.ops.push_back(o);
      pl.launches += 1;
      hr = o.out; xr = o.c;
    } else {
      const size_t ha = gn_apply32(x, m.cin, hw, n.cfg.act);
      hr = fir(ha, side, m.cin, m.up != 0);
      ar.release(ha);
      xr = fir(x, side, m.cin, m.up != 0);
    }
    const size_t tcol = (size_t)n.dense_col.at(m.idx);
    const size_t h1 = conv(k + ".Conv_0", hr, NONE, os, os, 1, 1, 0, false, 0, NONE, tcol, false);
    ar.release(hr);
    gn(h1, NONE, m.cout, 0, os * os, mname(m.idx, "GroupNorm_1.weight"), mname(m.idx, "GroupNorm_1.bias"));
    const size_t sc = conv(k + ".Conv_2", xr, NONE, os, os, 1, 0, 0, false, 0, NONE, NONE, false);
    ar.release(xr);
    pending_scale = skip_scale();
    const size_t out = conv(k + ".Conv_1", h1, NONE, os, os, 1, 1, 0, true, n.cfg.act, sc, NONE, false);
    ar.release(h1);
    ar.release(sc);
    return out;
  }

  size_t attn_block(const Module& m, size_t x, int hw_side, size_t out_to = NONE) {
    const std::string k = std::to_string(m.idx);
    const int C = m.cin, L = hw_side * hw_side;
    gn(x, NONE, C, 0, L, mname(m.idx, "GroupNorm_0.weight"), mname(m.idx, "GroupNorm_0.bias"));
    const size_t qkv = conv(k + ".qkv", x, NONE, hw_side, hw_side, 1, 0, 0, true, CSD_ACT_NONE, NONE, NONE, false);
    Op a;
    a.kind = OP_ATTN;
    a.a = qkv; a.i0 = L; a.i1 = C;
    a.out = alloc_((size_t)B * L * C);
    a.cls = CSD_PROF_ATTENTION;
    pl.ops.push_back(a);
    count(4.0 * B * (double)L * L * C, 0);
    pending_scale = skip_scale();
    next_out = out_to;
    const size_t o = conv(k + ".NIN_3", a.out, NONE, hw_side, hw_side, 1, 0, 0, false, 0, x, NONE, false);
    ar.release(qkv);
    ar.release(a.out);
    return o;
  }
};

// after the walk: a GroupNorm whose statistics + finalize pair survived (its consumer applies scale / shift itself: the attention
// blocks' q/k/v contraction, the last layer) becomes ONE launch when the map is small enough for the one-sweep kernel
static void fold_small_gn_pairs(Plan& pl) {
  if (CSD_TUNE_ENV("CSD_NO_GN_FUSED")) return;
  std::vector<Op> out;
  out.reserve(pl.ops.size());
  for (size_t i = 0; i < pl.ops.size(); ++i) {
    const Op& a = pl.ops[i];
    if (a.kind == OP_GN_STATS && i + 1 < pl.ops.size() && pl.ops[i + 1].kind == OP_GN_FINAL &&
        gn_fused16_groups(a.gp.HW, a.gp.C0, a.gp.C1, a.gp.G) > 0) {
      const Op& f = pl.ops[i + 1];
      Op o;
      o.kind = OP_GN_STATFIN;
      o.a = a.a; o.b = a.b; o.gp = a.gp;
      o.pk0 = f.pk0; o.pk1 = f.pk1;
      o.out = f.out; o.c = f.b;                       // nscale, nshift
      o.cls = CSD_PROF_GN_STATS; o.bytes = a.bytes;
      o.nb = a.nb; o.stream = a.stream; o.chunk = a.chunk;
      out.push_back(o);
      pl.launches -= 1;
      ++i;
    } else {
      out.push_back(a);
    }
  }
  pl.ops.swap(out);
}

// profiling classes: when the plan runs its big 3x3 stride-1 layers on the fused-prologue kernel (conv_xk / conv_ff: OP_CONV with i2 == 3),
// class CSD_PROF_CONV3X3 is THAT kernel's launches - the bench's roofline object prices the dominant kernel, SURVEY 8(d) - and the
// other 3x3 stride-1 launches (the first layer, the quad kernel on the small maps) report as CSD_PROF_CONV3X3_OTHER
static void split_conv3x3_classes(Plan& pl) {
  bool any_ff = false;
  for (const Op& o : pl.ops) any_ff = any_ff || (o.kind == OP_CONV && o.i2 == 3);
  if (!any_ff) return;
  for (Op& o : pl.ops)
    if (o.cls == CSD_PROF_CONV3X3 && !(o.kind == OP_CONV && o.i2 == 3)) o.cls = CSD_PROF_CONV3X3_OTHER;
}

// the ops of a chunk region are enqueued round robin over the chunks (op j of every chunk, then op j + 1): when the host is the slower
// side, every chunk stream still advances - chunk after chunk, the last stream would start when the first is nearly done
static void interleave_chunks(Plan& pl) {
  std::vector<Op> out;
  out.reserve(pl.ops.size());
  for (size_t i = 0; i < pl.ops.size();) {
    if (pl.ops[i].kind != OP_FORK) { out.push_back(pl.ops[i++]); continue; }
    out.push_back(pl.ops[i++]);
    std::vector<std::vector<Op>> per;
    while (i < pl.ops.size() && pl.ops[i].kind != OP_JOIN) {
      const int k = pl.ops[i].chunk - 1;
      if (k < 0) break;
      if ((int)per.size() <= k) per.resize(k + 1);
      per[k].push_back(pl.ops[i++]);
    }
    size_t longest = 0;
    for (auto& v : per) longest = std::max(longest, v.size());
    for (size_t j = 0; j < longest; ++j)
      for (auto& v : per)
        if (j < v.size()) out.push_back(v[j]);
  }
  pl.ops.swap(out);
}

static int build_plan(Net& n, int B, Plan** out) {
  auto it = n.plans.find(B);
  if (it != n.plans.end()) { *out = it->second.get(); return CSD_OK; }
  CSD_REQUIRE(B >= 1, "unet: batch must be >= 1");
  std::unique_ptr<Plan> plp(new Plan());
  Plan& pl = *plp;
  pl.B = B;
  Builder bd(n, pl, B);
  const csd_unet_config& c = n.cfg;
  const int S = c.image_size, nf = c.nf;
  // shared scratch: worst-case GN partials / scale+shift
  int cmax = nf;
  for (auto& m : n.mods) cmax = std::max(cmax, std::max(m.cin, m.cout));
  {
    GNPlan g;
    size_t worst = 0;
    for (int l = 0; l < c.n_levels; ++l) {
      const int side = S >> l;
      gn_plan(&g, B, side * side, cmax, 0, 32);
      worst = std::max(worst, gn_partial_bytes(g));
    }
    bd.gn_partial = bd.alloc_(worst / sizeof(float) + 64);
    bd.nscale = bd.alloc_((size_t)B * cmax);
    bd.nshift = bd.alloc_((size_t)B * cmax);
  }
  size_t mi = 0;
  auto next_mod = [&]() -> const Module& { return n.mods[mi++]; };

  if (c.arch == 1) {
    // ================= NCSN++: NCSNpp.forward (models/ncsnpp.py:238-388) =================
    const int channels = c.x_channels + c.y_channels;
    auto is_attn1 = [&](int res) {
      for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
      return false;
    };
    const PackedConv& stem_pc = n.pconvs[n.pconv_by_name.at(std::to_string(n.mods[(c.embedding_type == 1 ? 1 : 0) + 2].idx))];
    Op as;
    as.kind = OP_ASSEMBLE;
    as.out = NONE;
    if (!stem_pc.stem || c.progressive_input == 1) {      // (the input pyramid reads the assembled tensor)
      as.out = bd.alloc_((size_t)B * S * S * n.in_cpad);
      pl.ops.push_back(as);
      pl.launches += 1;
    }
    // time embedding: Fourier features of the label (or the sinusoidal embedding), two Linear layers, all Dense_0
    size_t emb;
    int emb_dim = nf;
    if (c.embedding_type == 1) {
      const Module& fm = next_mod();
      Op e;
      e.kind = OP_FOURIER;
      e.pk0 = n.copy_off.at(mname(fm.idx, "W"));
      e.i0 = nf;
      e.out = bd.alloc_((size_t)B * 2 * nf);
      pl.ops.push_back(e);
      emb = e.out;
      emb_dim = 2 * nf;
    } else {
      Op e;
      e.kind = OP_TEMB;
      e.out = bd.alloc_((size_t)B * nf);
      e.i0 = nf;
      pl.ops.push_back(e);
      emb = e.out;
    }
    {
      const Module& l0 = next_mod();
      const Module& l1 = next_mod();
      Op a;
      a.kind = OP_LINEAR;
      a.a = emb; a.out = bd.alloc_((size_t)B * 4 * nf);
      a.pk0 = n.copy_off.at(mname(l0.idx, "weight")); a.pk1 = n.copy_off.at(mname(l0.idx, "bias"));
      a.i0 = emb_dim; a.i1 = 4 * nf; a.act = CSD_ACT_NONE;
      a.i2 = c.act;          // the activation every consumer applies to this output, applied ONCE by the producer (the consumers used
                             // to recompute it per workgroup: 396 x 24.6 k SiLUs in the Dense_0 launch)
      pl.ops.push_back(a);
      Op b2;
      b2.kind = OP_LINEAR;
      b2.a = a.out; b2.out = bd.alloc_((size_t)B * 4 * nf);
      b2.pk0 = n.copy_off.at(mname(l1.idx, "weight")); b2.pk1 = n.copy_off.at(mname(l1.idx, "bias"));
      b2.i0 = 4 * nf; b2.i1 = 4 * nf; b2.act = CSD_ACT_NONE; b2.i2 = c.act;
      pl.ops.push_back(b2);
      Op d;
      d.kind = OP_LINEAR;   // every block's Dense_0(act(temb)) in one launch
      d.a = b2.out; d.out = bd.alloc_((size_t)B * n.dense_total);
      d.pk0 = n.dense_all_off; d.pk1 = n.dense_all_bias_off;
      d.i0 = 4 * nf; d.i1 = n.dense_total; d.act = CSD_ACT_NONE;
      pl.ops.push_back(d);
      bd.dense_all = d.out;
      pl.launches += 4;
      pl.flops += 2.0 * B * ((double)emb_dim * 4 * nf + 16.0 * nf * nf + 4.0 * nf * n.dense_total);
    }
    struct Skip { size_t off; int ch; };
    std::vector<Skip> hs;
    size_t pyr_in = c.progressive_input == 1 ? as.out : NONE;     // input pyramid (8-channel padded NHWC)
    int pyr_side = S;
    {
      const Module& m = next_mod();
      const size_t h0 = stem_pc.stem ? bd.stem(stem_pc, m, S)
                                     : bd.conv(std::to_string(m.idx), as.out, NONE, S, S, 1, 1, 0, false, 0, NONE, NONE, false, channels);
      hs.push_back({h0, m.cout});
    }
    int in_ch = nf;
    for (int l = 0; l < c.n_levels; ++l) {
      const int side = S >> l;
      for (int b = 0; b < c.num_res_blocks; ++b) {
        const Module& m = next_mod();
        size_t h = bd.res_block(m, hs.back().off, NONE, in_ch, 0, side);
        in_ch = m.cout;
        if (is_attn1(side)) {
          const Module& am = next_mod();
          const size_t h2 = bd.attn_block(am, h, side);
          bd.ar.release(h);
          h = h2;
        }
        hs.push_back({h, in_ch});
      }
      if (l != c.n_levels - 1) {
        const Module& m = next_mod();
        size_t h = bd.res_block_updown(m, hs.back().off, side);
        if (c.progressive_input == 1) {
          const Module& cm = next_mod();
          const size_t pn = bd.fir(pyr_in, pyr_side, n.in_cpad, false);      // pyramid_downsample (FIR, no conv)
          if (pyr_in != as.out) bd.ar.release(pyr_in);
          pyr_in = pn;
          pyr_side /= 2;
          // Combine 'sum' (layerspp.py:53-57): Conv_0(input_pyramid) + h
          const size_t hc2 = bd.conv(std::to_string(cm.idx) + ".Conv_0", pyr_in, NONE, side / 2, side / 2, 1, 0, 0, false, 0, h,
                                     NONE, false, channels);
          bd.ar.release(h);
          h = hc2;
        }
        hs.push_back({h, in_ch});
      }
    }
    if (pyr_in != NONE && pyr_in != as.out) bd.ar.release(pyr_in);
    bd.ar.release(as.out);
    size_t h = hs.back().off;
    {
      const int side = S >> (c.n_levels - 1);
      const Module& r0 = next_mod();
      size_t t0 = bd.res_block(r0, h, NONE, in_ch, 0, side);
      const Module& am = next_mod();
      size_t t1 = bd.attn_block(am, t0, side);
      bd.ar.release(t0);
      const Module& r1 = next_mod();
      size_t t2 = bd.res_block(r1, t1, NONE, in_ch, 0, side);
      bd.ar.release(t1);
      h = t2;
    }
    size_t pyr = NONE;      // output pyramid [B, side, side, channels]
    for (int l = c.n_levels - 1; l >= 0; --l) {
      const int side = S >> l;
      for (int b = 0; b < c.num_res_blocks + 1; ++b) {
        const Module& m = next_mod();
        const Skip sk = hs.back();
        hs.pop_back();
        const size_t o = bd.res_block(m, h, sk.off, in_ch, sk.ch, side);
        bd.ar.release(h);
        bd.ar.release(sk.off);
        h = o;
        in_ch = m.cout;
      }
      if (is_attn1(side)) {
        const Module& am = next_mod();
        const size_t o = bd.attn_block(am, h, side);
        bd.ar.release(h);
        h = o;
      }
      if (c.progressive == 1) {
        const Module& g = next_mod();
        const Module& cm = next_mod();
        size_t res = NONE;
        if (pyr != NONE) {
          res = bd.fir(pyr, side / 2, channels, true);                       // pyramid_upsample
          bd.ar.release(pyr);
        }
        bd.gn(h, NONE, in_ch, 0, side * side, mname(g.idx, "weight"), mname(g.idx, "bias"));
        const bool last = (l == 0);
        const size_t po = bd.conv(std::to_string(cm.idx), h, NONE, side, side, 1, 1, 0, true, c.act, res, NONE, last);
        bd.ar.release(res);
        pyr = po;        // (NONE when written straight to the caller's NCHW output)
      }
      if (l != 0) {
        const Module& m = next_mod();
        const size_t o = bd.res_block_updown(m, h, side);
        bd.ar.release(h);
        h = o;
      }
    }
    if (c.progressive != 1) {
      const Module& g = next_mod();
      bd.gn(h, NONE, in_ch, 0, S * S, mname(g.idx, "weight"), mname(g.idx, "bias"));
      const Module& m = next_mod();
      bd.conv(std::to_string(m.idx), h, NONE, S, S, 1, 1, 0, true, c.act, NONE, NONE, true);
    }
    bd.ar.release(h);
    if (bd.rc) return bd.rc;
    CSD_REQUIRE(mi == n.mods.size() && hs.empty(), "ncsnpp: plan walk mismatch");
    double pbytes1 = 0;
    for (auto& p : n.params) pbytes1 += 4.0 * p.numel;
    pl.bytes += pbytes1;
    fold_small_gn_pairs(pl);
    split_conv3x3_classes(pl);
    pl.ws_floats = bd.ar.peak();
    *out = plp.get();
    n.plans[B] = std::move(plp);
    return CSD_OK;
  }

  // ---- input + time embedding ----
  const PackedConv& stem_pc = n.pconvs[n.pconv_by_name.at(std::to_string(n.mods[c.conditional ? 2 : 0].idx))];
  Op as;
  as.kind = OP_ASSEMBLE;
  as.out = NONE;
  if (!stem_pc.stem) {
    as.out = bd.alloc_((size_t)B * S * S * n.in_cpad);
    pl.ops.push_back(as);
    pl.launches += 1;
  }
  if (c.conditional) {
    const Module& l0 = next_mod();
    const Module& l1 = next_mod();
    Op e;
    e.kind = OP_TEMB;
    e.out = bd.alloc_((size_t)B * nf);
    e.i0 = nf;
    pl.ops.push_back(e);
    Op a;
    a.kind = OP_LINEAR;
    a.a = e.out; a.out = bd.alloc_((size_t)B * 4 * nf);
    a.pk0 = n.copy_off.at(mname(l0.idx, "weight")); a.pk1 = n.copy_off.at(mname(l0.idx, "bias"));
    a.i0 = nf; a.i1 = 4 * nf; a.act = CSD_ACT_NONE;
      a.i2 = c.act;          // the activation every consumer applies to this output, applied ONCE by the producer (the consumers used
                             // to recompute it per workgroup: 396 x 24.6 k SiLUs in the Dense_0 launch)
    pl.ops.push_back(a);
    Op b2;
    b2.kind = OP_LINEAR;
    b2.a = a.out; b2.out = bd.alloc_((size_t)B * 4 * nf);
    b2.pk0 = n.copy_off.at(mname(l1.idx, "weight")); b2.pk1 = n.copy_off.at(mname(l1.idx, "bias"));
    b2.i0 = 4 * nf; b2.i1 = 4 * nf; b2.act = CSD_ACT_NONE; b2.i2 = c.act;
    pl.ops.push_back(b2);
    Op d;
    d.kind = OP_LINEAR;   // every ResnetBlock's Dense_0(act(temb)) in one launch
    d.a = b2.out; d.out = bd.alloc_((size_t)B * n.dense_total);
    d.pk0 = n.dense_all_off; d.pk1 = n.dense_all_bias_off;
    d.i0 = 4 * nf; d.i1 = n.dense_total; d.act = CSD_ACT_NONE;
    pl.ops.push_back(d);
    bd.dense_all = d.out;
    pl.launches += 4;
    pl.flops += 2.0 * B * ((double)nf * 4 * nf + 16.0 * nf * nf + 4.0 * nf * n.dense_total);
    pl.bytes += 4.0 * B * (nf + 4 * nf + 4 * nf + 4 * nf + (4.0 * nf + 1) * 0 + 2.0 * n.dense_total);
  }
  auto is_attn = [&](int res) {
    for (int i = 0; i < c.n_attn; ++i) if (c.attn_resolutions[i] == res) return true;
    return false;
  };
  struct Skip { size_t off; int ch; };
  std::vector<Skip> hs;
  {
    const Module& m = next_mod();
    size_t h0;
    if (stem_pc.stem) {
      h0 = bd.stem(stem_pc, m, S);
    } else {
      h0 = bd.conv(std::to_string(m.idx), as.out, NONE, S, S, 1, 1, 0, false, 0, NONE, NONE, false,
                   c.x_channels + c.y_channels);
      bd.ar.release(as.out);
    }
    hs.push_back({h0, m.cout});
  }
  int in_ch = nf;
  const int last = c.n_levels - 1;
  size_t h = NONE;
  // the walk, level by level (models/ddpm.py:149-213): every piece works on the Builder's CURRENT batch (bd.B), skip stack and position
  auto down_blocks = [&](int l) {
    const int side = S >> l;
    for (int b = 0; b < c.num_res_blocks; ++b) {
      const Module& m = next_mod();
      size_t hb = bd.res_block(m, hs.back().off, NONE, in_ch, 0, side);
      in_ch = m.cout;
      if (is_attn(side)) {
        const Module& am = next_mod();
        const size_t h2 = bd.attn_block(am, hb, side);
        bd.ar.release(hb);
        hb = h2;
      }
      hs.push_back({hb, in_ch});
    }
  };
  auto down_sample = [&](int l) {
    const int side = S >> l;
    const Module& m = next_mod();
    size_t d;
    if (c.resamp_with_conv) {
      d = bd.conv(std::to_string(m.idx) + ".Conv_0", hs.back().off, NONE, side, side, 2, 0, 0, false, 0, NONE,
                  NONE, false);
    } else {
      Op p;
      p.kind = OP_AVGPOOL;
      p.a = hs.back().off; p.i0 = side; p.i1 = in_ch;
      p.out = bd.alloc_((size_t)bd.B * (side / 2) * (side / 2) * in_ch);
      pl.ops.push_back(p);
      pl.launches += 1;
      d = p.out;
    }
    hs.push_back({d, in_ch});
  };
  auto middle = [&]() {
    h = hs.back().off;   // stays on the stack: popped by the first up-path block
    const int side = S >> last;
    const Module& r0 = next_mod();
    size_t t0 = bd.res_block(r0, h, NONE, in_ch, 0, side);
    const Module& am = next_mod();
    size_t t1 = bd.attn_block(am, t0, side);
    bd.ar.release(t0);
    const Module& r1 = next_mod();
    size_t t2 = bd.res_block(r1, t1, NONE, in_ch, 0, side);
    bd.ar.release(t1);
    h = t2;
  };
  // (out_to: the level's LAST launch writes there instead of into a tensor of its own - a batch chunk's slice of the region's output)
  auto up_blocks = [&](int l, size_t out_to) {
    const int side = S >> l;
    for (int b = 0; b < c.num_res_blocks + 1; ++b) {
      const Module& m = next_mod();
      const Skip sk = hs.back();
      hs.pop_back();
      const size_t o = bd.res_block(m, h, sk.off, in_ch, sk.ch, side, (b == c.num_res_blocks && !is_attn(side)) ? out_to : NONE);
      bd.ar.release(h);
      bd.ar.release(sk.off);
      h = o;
      in_ch = m.cout;
    }
    if (is_attn(side)) {
      const Module& am = next_mod();
      const size_t o = bd.attn_block(am, h, side, out_to);
      bd.ar.release(h);
      h = o;
    }
  };
  auto up_sample = [&](int l) {
    const int side = S >> l;
    const Module& m = next_mod();
    size_t o;
    if (c.resamp_with_conv) {
      o = bd.conv(std::to_string(m.idx) + ".Conv_0", h, NONE, side, side, 1, 1, 1, false, 0, NONE, NONE, false);
    } else {
      Op p;
      p.kind = OP_UPNEAR;
      p.a = h; p.i0 = side; p.i1 = in_ch;
      p.out = bd.alloc_((size_t)bd.B * side * 2 * side * 2 * in_ch);
      pl.ops.push_back(p);
      pl.launches += 1;
      o = p.out;
    }
    bd.ar.release(h);
    h = o;
  };
  // levels l0 .. last (small maps) down, the middle, and up again to the end of level l0's blocks
  auto small_levels = [&](int l0, size_t out_to) {
    for (int l = l0; l <= last; ++l) {
      down_blocks(l);
      if (l != last) down_sample(l);
    }
    middle();
    for (int l = last; l >= l0; --l) {
      up_blocks(l, l == l0 ? out_to : NONE);
      if (l != l0) up_sample(l);
    }
  };
  // Batch chunks: below CHUNK_SIDE^2 a launch is bound by its own dependent chain (weights -> LDS, K steps of global loads, epilogue), not by
  // throughput - at 5^2 / 10^2 a kernel takes the same ~20 us for 8 samples as for 64 (profiles/r06_*: timeline by batch).  The levels
  // l0 .. last therefore run as K independent sub-batches on K streams between the Downsample conv that enters level l0 and the Upsample
  // conv that leaves it: each chunk walks the same modules with its own temporaries (a private block of the workspace) and reads /
  // writes its slice of the two full-batch tensors at the region's boundary.  A sample's bits do not depend on the batch it runs in
  // (tests: B = 64 == B = 1), so the chunked plan returns the bits of the unchunked one.
  int l0 = c.n_levels, K = 1;
  {
    const char* e_side = CSD_TUNE_ENV("CSD_CHUNK_SIDE");
    const char* e_k = CSD_TUNE_ENV("CSD_CHUNKS");
    const int chunk_side = e_side ? atoi(e_side) : 20;
    for (int l = 1; l <= last; ++l)
      if ((S >> l) <= chunk_side) { l0 = l; break; }
    K = e_k ? atoi(e_k) : (B >= 64 ? 2 : 1);      // measured (NOTEBOOK round 6): B = 64: 2 chunks -0.24 ms per PC step, 3 chunks +0.25; B = 16 .. 48: 2 chunks +0.05 .. +0.23 ms
    K = std::max(1, std::min(K, std::min(B, (int)Net::MAX_CHUNKS)));
    if (l0 > last) K = 1;
  }
  for (int l = 0; l < (K > 1 ? l0 : c.n_levels); ++l) {
    down_blocks(l);
    if (l != last) down_sample(l);
  }
  if (K == 1) {
    middle();
    for (int l = last; l >= 0; --l) {
      up_blocks(l, NONE);
      if (l != 0) up_sample(l);
    }
  } else {
    const int side0 = S >> l0, Bc = cdiv(B, K);
    const Skip in_full = hs.back();                      // [B, side0, side0, ch]: the Downsample output that enters level l0
    const size_t in_ps = (size_t)side0 * side0 * in_full.ch;
    const size_t mi0 = mi;
    const int in_ch0 = in_ch;
    const size_t dense_full = bd.dense_all, gp_full = bd.gn_partial, ns_full = bd.nscale, nh_full = bd.nshift;
    std::vector<Skip> hs_main;
    hs_main.swap(hs);
    auto tile_stats_main = bd.tile_stats;
    Arena ar_main = bd.ar;
    int ch_out = 0;
    // one chunk's walk in the Builder's current arena; returns false on a builder error
    auto chunk_walk = [&](int b0, int nb, size_t out_slice) {
      bd.B = nb;
      bd.tile_stats.clear();
      mi = mi0; in_ch = in_ch0;
      hs.clear();
      hs.push_back({in_full.off + (size_t)b0 * in_ps, in_full.ch});
      bd.dense_all = dense_full == NONE ? NONE : dense_full + (size_t)b0 * n.dense_total;
      {      // the chunk's own GroupNorm scratch
        GNPlan g;
        size_t worst = 0;
        for (int l = l0; l <= last; ++l) {
          gn_plan(&g, nb, (S >> l) * (S >> l), cmax, 0, 32);
          worst = std::max(worst, gn_partial_bytes(g));
        }
        bd.gn_partial = bd.alloc_(worst / sizeof(float) + 64);
        bd.nscale = bd.alloc_((size_t)nb * cmax);
        bd.nshift = bd.alloc_((size_t)nb * cmax);
      }
      small_levels(l0, out_slice);
      ch_out = in_ch;
      return bd.rc == CSD_OK && hs.empty();
    };
    // dry run of one full-size chunk: the size of a chunk's private block
    size_t chunk_floats = 0;
    {
      const size_t nops = pl.ops.size();
      const int64_t launches = pl.launches;
      const double flops = pl.flops, bytes = pl.bytes;
      bd.ar = Arena(0);
      const bool ok = chunk_walk(0, Bc, 0);
      chunk_floats = (bd.ar.peak() + 63) / 64 * 64;
      pl.ops.resize(nops);
      pl.launches = launches; pl.flops = flops; pl.bytes = bytes;
      bd.next_out = NONE;
      if (!ok) { if (bd.rc) return bd.rc; set_error("unet: chunk walk mismatch"); return CSD_ERR_STATE; }
    }
    bd.ar = ar_main;
    const size_t out_ps = (size_t)side0 * side0 * ch_out;
    const size_t out_full = bd.alloc_((size_t)B * out_ps);
    const size_t block = bd.alloc_((size_t)K * chunk_floats);
    ar_main = bd.ar;
    {
      Op f;
      f.kind = OP_FORK;
      f.i0 = K;
      pl.ops.push_back(f);
    }
    for (int k = 0; k < K; ++k) {
      const int b0 = k * Bc, nb = std::min(Bc, B - b0);
      if (nb <= 0) break;
      const size_t i0 = pl.ops.size();
      bd.ar = Arena(block + (size_t)k * chunk_floats);
      const bool ok = chunk_walk(b0, nb, out_full + (size_t)b0 * out_ps);
      if (!ok) { if (bd.rc) return bd.rc; set_error("unet: chunk walk mismatch"); return CSD_ERR_STATE; }
      CSD_REQUIRE(bd.ar.peak() <= chunk_floats, "unet: a chunk outgrew its block");
      for (size_t i = i0; i < pl.ops.size(); ++i) { pl.ops[i].nb = nb; pl.ops[i].stream = k; pl.ops[i].chunk = k + 1; }
    }
    {
      Op j;
      j.kind = OP_JOIN;
      j.i0 = K;
      pl.ops.push_back(j);
    }
    // back in the full batch
    bd.ar = ar_main;
    bd.B = B;
    bd.tile_stats = tile_stats_main;
    bd.tile_stats.erase(out_full);
    bd.dense_all = dense_full; bd.gn_partial = gp_full; bd.nscale = ns_full; bd.nshift = nh_full;
    hs.swap(hs_main);
    bd.ar.release(block);
    bd.ar.release(hs.back().off);      // the region's input: consumed by every chunk's last up block
    hs.pop_back();
    h = out_full;
    up_sample(l0);
    for (int l = l0 - 1; l >= 0; --l) {
      up_blocks(l, NONE);
      if (l != 0) up_sample(l);
    }
  }
  {
    const Module& g = next_mod();
    bd.gn(h, NONE, in_ch, 0, S * S, mname(g.idx, "weight"), mname(g.idx, "bias"));
    const Module& m = next_mod();
    bd.conv(std::to_string(m.idx), h, NONE, S, S, 1, 1, 0, true, c.act, NONE, NONE, true);
    bd.ar.release(h);
  }
  if (bd.rc) return bd.rc;
  CSD_REQUIRE(mi == n.mods.size() && hs.empty(), "unet: plan walk mismatch");
  // parameters are read once per forward (+ param_bytes per NFE)
  double pbytes = 0;
  for (auto& p : n.params) pbytes += 4.0 * p.numel;
  pl.bytes += pbytes;
  fold_small_gn_pairs(pl);
  split_conv3x3_classes(pl);
  if (!CSD_TUNE_ENV("CSD_CHUNK_SEQ")) interleave_chunks(pl);
  pl.ws_floats = bd.ar.peak();
  *out = plp.get();
  n.plans[B] = std::move(plp);
  return CSD_OK;
}

// ---- run ------------------------------------------------------------------------------------------
static int run_plan(Net& n, const Plan& pl, const float* pk, float* ws, const float* x, const float* y,
                    const float* labels, float* out, const float* y_noise, float y_sigma, hipStream_t s) {
  const csd_unet_config& c = n.cfg;
  const int B = pl.B, S = c.image_size;
  auto W = [&](size_t off) -> float* { return off == NONE ? nullptr : ws + off; };
  bool side_pending = false;
  for (const Op& o : pl.ops) {
    int rc = CSD_OK;
    // batch-chunk region: the chunk streams start behind everything enqueued so far; the caller's stream resumes behind all of them
    if (o.kind == OP_FORK) {
      if (!n.chunks_ready(o.i0 - 1)) { set_error("unet: cannot create the chunk streams"); return CSD_ERR_HIP; }
      CSD_CHECK_HIP(hipEventRecord(n.ev_cfork, s));
      for (int k = 0; k + 1 < o.i0; ++k) CSD_CHECK_HIP(hipStreamWaitEvent(n.cstream[k], n.ev_cfork, 0));
      continue;
    }
    if (o.kind == OP_JOIN) {
      for (int k = 0; k + 1 < o.i0; ++k) {
        CSD_CHECK_HIP(hipEventRecord(n.ev_cjoin[k], n.cstream[k]));
        CSD_CHECK_HIP(hipStreamWaitEvent(s, n.ev_cjoin[k], 0));
      }
      continue;
    }
    const int Bo = o.nb ? o.nb : B;
    // side-stream ops: fork after everything enqueued so far, join before the op that consumes the result
    hipStream_t so = o.stream ? n.cstream[o.stream - 1] : s;
    if (o.side == 1 && n.side_ready()) {
      CSD_CHECK_HIP(hipEventRecord(n.ev_fork, s));
      CSD_CHECK_HIP(hipStreamWaitEvent(n.side, n.ev_fork, 0));
      so = n.side;
    } else if (o.side == 2 && side_pending) {
      CSD_CHECK_HIP(hipStreamWaitEvent(s, n.ev_join, 0));
      side_pending = false;
    }
    ProfScope prof(o.cls, o.flops, o.bytes, so, o.abytes);
    switch (o.kind) {
      case OP_ASSEMBLE:
        rc = assemble_input_launch(x, y, y_noise, y_sigma, W(o.out), Bo, c.x_channels, c.y_channels, S * S,
                                   n.in_cpad, c.centered, so);
        break;
      case OP_STEM:
        rc = stem_launch(x, y, y_noise, y_sigma, pk + o.pk0, pk + o.pk1, W(o.out), reinterpret_cast<double*>(W(o.stats)), Bo,
                         c.x_channels, c.y_channels, o.i0, S, c.centered, o.i4, so);
        break;
      case OP_TEMB:
        rc = timestep_embedding_launch(labels, W(o.out), Bo, o.i0, so);
        break;
      case OP_FOURIER:
        rc = fourier_embedding_launch(labels, pk + o.pk0, W(o.out), Bo, o.i0, so);
        break;
      case OP_FIR:
        rc = fir_resample_nhwc_launch(W(o.a), W(o.out), Bo, o.i0, o.i0, o.i1, c.fir_kernel, o.i2, so);
        break;
      case OP_FIR2:
        rc = fir_resample2_nhwc_launch(W(o.a), W(o.d), W(o.e), W(o.c), W(o.out), Bo, o.i0, o.i0, o.i1, c.fir_kernel, o.i2, o.act, so);
        break;
      case OP_GN_APPLY32:
        rc = gn_apply_launch(W(o.a), W(o.d), W(o.e), W(o.out), Bo, o.i1, o.i0, o.act, so);
        break;
      case OP_LINEAR:
        rc = linear_launch(W(o.a), pk + o.pk0, pk + o.pk1, W(o.out), Bo, o.i0, o.i1, o.act, so, o.i2);
        break;
      case OP_GN_STATS:
        rc = gn_stats_launch(o.gp, W(o.a), W(o.b), reinterpret_cast<double*>(W(o.out)), so, o.i0);
        break;
      case OP_GN_FINAL:
        rc = gn_finalize_launch(o.gp, reinterpret_cast<const double*>(W(o.a)), pk + o.pk0, pk + o.pk1, 1e-6f,
                                W(o.out), W(o.b), so);
        break;
      case OP_GN_FINAL_TILES:
        rc = gn_finalize_tiles_launch(reinterpret_cast<const double*>(W(o.a)), o.i0, o.i1,
                                      reinterpret_cast<const double*>(W(o.b)), o.i2, o.i3, Bo, o.i4, o.gp.G, pk + o.pk0,
                                      pk + o.pk1, 1e-6f, W(o.out), W(o.c), so);
        break;
      case OP_GN_STATFIN:
        rc = gn_fused16_launch(W(o.a), W(o.b), o.gp.C0, o.gp.C1, pk + o.pk0, pk + o.pk1, 1e-6f, nullptr, nullptr, Bo, o.gp.HW, o.gp.G,
                               CSD_ACT_NONE, so, 0, W(o.out), W(o.c));
        break;
      case OP_GN_FUSED16:
        rc = gn_fused16_launch(W(o.a), W(o.b), o.i0, o.i1, pk + o.pk0, pk + o.pk1, 1e-6f, W(o.out), W(o.c), Bo, o.i2, o.gp.G, o.act, so, o.i3);
        break;
      case OP_GN_APPLY16:
        rc = gn_apply16_launch(W(o.a), W(o.b), o.i0, o.i1, W(o.d), W(o.e), W(o.out), W(o.c), Bo, o.i2, o.act, so, o.i3);
        break;
      case OP_CONV: {
        ConvArgs a;
        a.src0 = W(o.a); a.src1 = W(o.b);
        a.wpack = pk + o.pk0; a.bias = o.pk1 == NONE ? nullptr : pk + o.pk1;
        a.temb = o.temb_col == NONE ? nullptr : ws + o.temb_base + o.temb_col;
        a.res = W(o.c);
        a.nscale = W(o.d);
        a.nshift = W(o.e);
        a.out = o.out_external ? out : W(o.out);
        a.temb_stride = o.temb_stride;
        a.out_stride = o.cp.Cout; a.out_coff = 0;
        a.out_nchw = o.out_external;
        a.act = o.act;
        a.out_scale = o.fscale;
        a.dbg = nullptr;
        a.stats = reinterpret_cast<double*>(W(o.stats));
        rc = o.i2 == 3 ? convff_launch(o.cp, o.i4, a, so)
           : o.i2 == 2 ? conv16q_launch(o.cp, o.i4, a, so, o.i3 == 2)
           : o.i2 ? pw16_launch(o.cp, o.i4, a, so) : (o.i4 ? conv16_launch(o.cp, o.i4, a, so, o.i3 != 0) : conv_launch(o.cp, a, so));
        if (so == n.side && n.side != nullptr && rc == CSD_OK) {
          CSD_CHECK_HIP(hipEventRecord(n.ev_join, n.side));
          side_pending = true;
        }
        break;
      }
      case OP_ATTN:
        // fp16 arithmetic modes: the split-operand kernel on the fp16 matrix cores (fp32-class in the split modes); fp32 mode: the fp32 MFMA one
        rc = (precision_ns(c.precision) && !CSD_TUNE_ENV("CSD_ATTN_F32"))
                 ? attention16_launch(W(o.a), 3 * o.i1, W(o.out), Bo, o.i0, o.i1, precision_ns(c.precision) >= 2 ? 2 : 1, so)
                 : attention_launch(W(o.a), 3 * o.i1, W(o.out), Bo, o.i0, o.i1, so);
        break;
      case OP_AVGPOOL:
        rc = avgpool2_launch(W(o.a), W(o.out), Bo, o.i0, o.i0, o.i1, so);
        break;
      case OP_UPNEAR:
        rc = nearest_up2_nhwc_launch(W(o.a), W(o.out), Bo, o.i0, o.i0, o.i1, so);
        break;
      case OP_TAPSUM:
        rc = tapsum_launch(W(o.a), pk + o.pk1, W(o.c), o.out_external ? out : W(o.out), Bo, o.i0, o.i1, o.i2, o.out_external, o.fscale, so);
        break;
      default:
        set_error("unet: unknown op");
        rc = CSD_ERR_STATE;
    }
    if (rc) return rc;
  }
  return CSD_OK;
}


// ---- weight packing -------------------------------------------------------------------------------
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void fill_f32_kernel(float* __restrict__ dst, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = v;
}

static int dev_copy(const float* src, float* dst, size_t nfl, hipStream_t s) {
  if (!nfl) return CSD_OK;
  const int grid = (int)std::min<size_t>(cdiv64(nfl, 256), 1024);
  hipLaunchKernelGGL(copy_f32_kernel, dim3(grid), dim3(256), 0, s, src, dst, nfl);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}
static int dev_fill(float* dst, float v, size_t nfl, hipStream_t s) {
  if (!nfl) return CSD_OK;
  const int grid = (int)std::min<size_t>(cdiv64(nfl, 256), 1024);
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid), dim3(256), 0, s, dst, v, nfl);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// every raw fp32 copy of a pack (GroupNorm affine, Linear / Dense weights, conv biases: ~480 of them at the SR3-160 shape) in ONE launch:
// a table of (source, destination offset, count) in device memory, one workgroup row per entry.  As separate launches they were 900
// kernel dispatches per pack - a third of the launches of a four-forward profile of NCSN++-256.
struct CopyDesc { const float* src; float* dst; unsigned long long n; };
__global__ void copy_table_kernel(const CopyDesc* __restrict__ tab) {
  const CopyDesc d = tab[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += (size_t)gridDim.x * blockDim.x) d.dst[i] = d.src[i];
}
// the host copy of the table belongs to the handle: the upload is stream-ordered and may still be in flight when pack returns, so the
// next pack waits for the event behind the previous upload before it rewrites the vector (no stream synchronisation in a pack)
static int run_copy_table(Net& n, hipStream_t s);

static int pack_all(Net& n, float* pk, hipStream_t s) {
  for (auto& p : n.params)
    CSD_REQUIRE(p.ptr != nullptr, "pack: parameter '%s' was never registered (csd_unet_set_param)", p.name.c_str());
  int rc;
  if (n.copy_ev) CSD_CHECK_HIP(hipEventSynchronize(n.copy_ev));      // (the previous pack's table upload has left the host vector)
  n.copy_host.clear();
  struct { Net& n; void add(const float* src, float* dst, size_t nfl) { if (nfl) n.copy_host.push_back(CopyDesc{src, dst, (unsigned long long)nfl}); } } batch{n};
  for (auto& cp : n.copies) batch.add(n.params[cp.param].ptr, pk + cp.off, (size_t)n.params[cp.param].numel);
  for (auto& pc : n.pconvs) {
    if ((rc = dev_fill(pk + pc.b_off, 0.f, (size_t)pc.proto.CoutPad, s))) return rc;
    for (auto& src : pc.srcs) {   // sources are listed with ascending cout_off, first one clears the tensor
      const int cin_src = src.cin_src > 0 ? src.cin_src : pc.proto.C0 + pc.proto.C1;
      ConvPlan one = pc.proto;
      one.C0 = pc.proto.C0 + pc.proto.C1; one.C1 = 0;
      rc = pc.stem ? stem_pack_weight(n.params[src.param_w].ptr, cin_src, pc.proto.Cout, pc.ns, pk + pc.w_off, s)
         : pc.ff ? convff_pack_weight(pc.proto, pc.ns, n.params[src.param_w].ptr, src.layout, cin_src, src.cout_src,
                                      src.cout_off, pk + pc.w_off, s)
         : pc.up4 ? conv16q_pack_weight_up4(one, pc.ns, n.params[src.param_w].ptr, pk + pc.w_off, s)
         : pc.q ? conv16q_pack_weight(one, pc.ns, n.params[src.param_w].ptr, src.layout, cin_src, src.cout_src,
                                      src.cout_off, pk + pc.w_off, s)
         : pc.tap_cout ? pw16_pack_weight_taps(pc.proto, pc.ns, n.params[src.param_w].ptr, pc.tap_cout, pk + pc.w_off, s)
         : pc.pw ? pw16_pack_weight(pc.proto, pc.ns, n.params[src.param_w].ptr, src.layout, cin_src, src.cout_src,
                                    src.cout_off, pk + pc.w_off, s)
         : pc.ns ? conv16_pack_weight(pc.proto, pc.ns, n.params[src.param_w].ptr, src.layout, cin_src, src.cout_src,
                                      src.cout_off, pk + pc.w_off, s)
                 : conv_pack_weight(pc.proto, n.params[src.param_w].ptr, src.layout, cin_src, src.cout_src,
                                    src.cout_off, pk + pc.w_off, s);
      if (rc) return rc;
      batch.add(n.params[src.param_b].ptr, pk + pc.b_off + src.cout_off, (size_t)src.cout_src);      // (behind the fill above: same stream)
    }
  }
  if (n.cfg.conditional) {
    const int K = 4 * n.cfg.nf;
    for (auto& m : n.mods) {
      if (m.kind != M_RES) continue;
      const int col = n.dense_col.at(m.idx);
      batch.add(n.params[n.P(mname(m.idx, "Dense_0.weight"))].ptr, pk + n.dense_all_off + (size_t)col * K, (size_t)m.cout * K);
      batch.add(n.params[n.P(mname(m.idx, "Dense_0.bias"))].ptr, pk + n.dense_all_bias_off + col, (size_t)m.cout);
    }
  }
  if ((rc = run_copy_table(n, s))) return rc;
  n.packed_once = true;
  return CSD_OK;
}

static int run_copy_table(Net& n, hipStream_t s) {
  if (n.copy_host.empty()) return CSD_OK;
  if (n.copy_tab_cap < n.copy_host.size()) {
    if (n.copy_tab) (void)hipFree(n.copy_tab);
    n.copy_tab = nullptr;
    CSD_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&n.copy_tab), n.copy_host.size() * sizeof(CopyDesc)));
    n.copy_tab_cap = n.copy_host.size();
  }
  if (!n.copy_ev) CSD_CHECK_HIP(hipEventCreateWithFlags(&n.copy_ev, hipEventDisableTiming));
  CSD_CHECK_HIP(hipMemcpyAsync(n.copy_tab, n.copy_host.data(), n.copy_host.size() * sizeof(CopyDesc), hipMemcpyHostToDevice, s));
  CSD_CHECK_HIP(hipEventRecord(n.copy_ev, s));
  size_t longest = 0;
  for (auto& d : n.copy_host) longest = std::max<size_t>(longest, d.n);
  const unsigned gx = (unsigned)std::min<size_t>(cdiv64(longest, 256), 64);
  hipLaunchKernelGGL(copy_table_kernel, dim3(gx, (unsigned)n.copy_host.size()), dim3(256), 0, s, n.copy_tab);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

}  // namespace csd

// =====================================================================================================
// C ABI
// =====================================================================================================
using namespace csd;

struct csd_unet {
  Net net;
};

#include "train_graph.h"

extern "C" int csd_profile_select(unsigned class_mask, int step_stride) {
  g_prof.mask = class_mask;
  g_prof.step_stride = step_stride > 0 ? step_stride : 1;
  g_prof.step_on = true;
  return CSD_OK;
}

extern "C" int csd_profile_start(void) {
  g_prof.on = true;
  g_prof.recs.clear();
  g_prof.used = 0;
  return CSD_OK;
}

extern "C" int csd_profile_stop(int n_classes, double* ms, int64_t* launches, double* flops, double* bytes) {
  return csd_profile_stop_ex(n_classes, ms, launches, flops, bytes, nullptr);
}

extern "C" int csd_profile_stop_ex(int n_classes, double* ms, int64_t* launches, double* flops, double* bytes, double* alg_bytes) {
  g_prof.on = false;
  CSD_REQUIRE(n_classes >= CSD_PROF_NUM_CLASSES && ms && launches && flops && bytes, "profile_stop: bad arguments");
  for (int i = 0; i < n_classes; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = 0; bytes[i] = 0; if (alg_bytes) alg_bytes[i] = 0; }
  if (g_prof.recs.empty()) return CSD_OK;
  for (auto& r : g_prof.recs) CSD_CHECK_HIP(hipEventSynchronize(r.b));      // (the records may sit on several streams: batch chunks)
  for (auto& r : g_prof.recs) {
    float t = 0.f;
    CSD_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cls] += t;
    launches[r.cls] += 1;
    flops[r.cls] += r.flops;
    bytes[r.cls] += r.bytes;
    if (alg_bytes) alg_bytes[r.cls] += r.abytes;
  }
  g_prof.recs.clear();
  g_prof.used = 0;
  return CSD_OK;
}

extern "C" const char* csd_version(void) { return "csd-hip 0.1 (gfx950)"; }
extern "C" const char* csd_last_error(void) { return get_error(); }

extern "C" int csd_unet_create(const csd_unet_config* cfg, csd_unet** out) {
  CSD_REQUIRE(cfg && out, "unet_create: null argument");
  std::unique_ptr<csd_unet> h(new csd_unet());
  h->net.cfg = *cfg;
  int rc = build_modules(h->net);
  if (rc) return rc;
  rc = build_packed_layout(h->net);
  if (rc) return rc;
  *out = h.release();
  return CSD_OK;
}

extern "C" void csd_unet_destroy(csd_unet* net) {
  if (net) {
    train_state_erase(&net->net);
  }
  delete net;
}

extern "C" int csd_unet_num_params(const csd_unet* net) { return net ? (int)net->net.params.size() : 0; }

extern "C" int csd_unet_param_info(const csd_unet* net, int index, const char** name, int* ndim, int64_t shape[4]) {
  CSD_REQUIRE(net && index >= 0 && index < (int)net->net.params.size(), "param_info: index %d out of range", index);
  const Param& p = net->net.params[index];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return CSD_OK;
}

extern "C" int csd_unet_set_param(csd_unet* net, const char* name, const void* dev_ptr, int64_t numel) {
  CSD_REQUIRE(net && name && dev_ptr, "set_param: null argument");
  const int i = net->net.P(name);
  if (i < 0) { set_error("set_param: unknown parameter '%s'", name); return CSD_ERR_NOT_FOUND; }
  Param& p = net->net.params[i];
  CSD_REQUIRE(p.numel == numel, "set_param: '%s' expects %lld elements, got %lld", name, (long long)p.numel,
              (long long)numel);
  CSD_REQUIRE((reinterpret_cast<uintptr_t>(dev_ptr) & 3) == 0, "set_param: '%s' is not 4-byte aligned", name);
  p.ptr = static_cast<const float*>(dev_ptr);
  return CSD_OK;
}

extern "C" size_t csd_unet_packed_bytes(const csd_unet* net) { return net ? net->net.packed_floats * sizeof(float) : 0; }

extern "C" int csd_unet_pack(csd_unet* net, void* packed, void* stream) {
  CSD_REQUIRE(net && packed, "pack: null argument");
  CSD_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 255) == 0, "pack: packed buffer must be 256-byte aligned");
  return pack_all(net->net, static_cast<float*>(packed), (hipStream_t)stream);
}

extern "C" size_t csd_unet_workspace_bytes(csd_unet* net, int B) {
  if (!net) return 0;
  Plan* pl = nullptr;
  if (build_plan(net->net, B, &pl)) return 0;
  return pl->ws_floats * sizeof(float);
}

extern "C" int csd_unet_stats(csd_unet* net, int B, int64_t* launches, double* flops, double* bytes) {
  CSD_REQUIRE(net, "stats: null handle");
  Plan* pl = nullptr;
  int rc = build_plan(net->net, B, &pl);
  if (rc) return rc;
  if (launches) *launches = pl->launches;
  if (flops) *flops = pl->flops;
  if (bytes) *bytes = pl->bytes;
  return CSD_OK;
}

static int check_forward_args(csd_unet* net, const void* packed, void* ws, size_t ws_bytes, int B, Plan** pl) {
  CSD_REQUIRE(net && packed && ws, "forward: null argument");
  if (!net->net.packed_once) { set_error("forward: csd_unet_pack has not been called"); return CSD_ERR_STATE; }
  CSD_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "forward: workspace must be 256-byte aligned");
  int rc = build_plan(net->net, B, pl);
  if (rc) return rc;
  if (ws_bytes < (*pl)->ws_floats * sizeof(float)) {
    set_error("forward: workspace too small (%zu < %zu bytes)", ws_bytes, (*pl)->ws_floats * sizeof(float));
    return CSD_ERR_WORKSPACE;
  }
  return CSD_OK;
}

extern "C" int csd_unet_forward(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes,
                                const float* x, const float* y, const float* labels, float* out, int B,
                                const float* y_noise, float y_sigma, void* stream) {
  Plan* pl = nullptr;
  int rc = check_forward_args(net, packed, workspace, workspace_bytes, B, &pl);
  if (rc) return rc;
  CSD_REQUIRE(x && out, "forward: null x/out");
  CSD_REQUIRE((net->net.cfg.y_channels == 0) == (y == nullptr), "forward: y must be given iff y_channels > 0");
  CSD_REQUIRE(!net->net.cfg.conditional || labels, "forward: labels required for a conditional network");
  return run_plan(net->net, *pl, static_cast<const float*>(packed), static_cast<float*>(workspace), x, y, labels,
                  out, y_noise, y_sigma, (hipStream_t)stream);
}

// ---- fused PC sampler -------------------------------------------------------------------------------
// scratch layout (floats): net_out [B*Co*HW] | x_mean [B*Cx*HW] | z [B*Cx*HW] | zy [B*Cy*HW] | labels [B]
//                          | partial (double) [B*64*2]
extern "C" size_t csd_pc_scratch_bytes(const csd_unet* net, int B) {
  if (!net) return 0;
  const csd_unet_config& c = net->net.cfg;
  const size_t hw = (size_t)c.image_size * c.image_size;
  size_t fl = 0;
  fl += align_up((size_t)B * c.out_channels * hw, 64);
  fl += 2 * align_up((size_t)B * c.x_channels * hw, 64);
  fl += align_up((size_t)B * std::max(c.y_channels, 1) * hw, 64);
  fl += align_up((size_t)B, 64);
  fl += align_up((size_t)B * std::max(c.y_channels, 1) * hw, 64);      // y_t of the use_path bridge
  return fl * sizeof(float) + (size_t)B * 64 * 2 * sizeof(double) + 256;
}

__global__ void fill_labels_kernel(float* dst, float v, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) dst[i] = v;
}

// one validated view of a csd_pc_* call: scratch carved up, noise bookkeeping per step
struct PCCtx {
  Net* n; Plan* pl; const float* pk; float* ws;
  float *net_out, *x_mean, *z, *zy, *labels, *ystate;
  double* partial;
  int* nonfinite;                                    // device flag of the finiteness contract (behind the norm partials)
  bool path = false;                                 // use_path: y_t follows the bridge (csd_pc_params.path_coef)
  const csd_pc_params* p;
  float* x; const float* y;
  int B, nchunk;
  size_t nx, ny;
  int64_t per, net_stride;
  bool perturb_y;
  hipStream_t s;
  // a phase whose rule is 'none' (csd_pc_params.corrector / .predictor == 2) evaluates nothing and draws nothing
  bool has_phase(int phase) const { return (phase == 0 ? p->corrector : p->predictor) != 2; }
  int draws_per_phase() const { return perturb_y ? 2 : 1; }
  int draws_per_step() const { return draws_per_phase() * ((has_phase(0) ? 1 : 0) + (has_phase(1) ? 1 : 0)); }
  // draw k (0-based, in the order of the phases that exist) of step i: from the tape (reference order) or Philox stream
  // 1 + i*draws + k (stream 0 is the prior)
  // use_path draws: -1 = z_y0 (before the loop); step i: 0 = z_y, 1 .. = the existing phases in the order predictor, corrector
  const float* noise_path(int i, int k, float* dst, size_t n) const {
    const int nph = (has_phase(0) ? 1 : 0) + (has_phase(1) ? 1 : 0);
    if (p->noise_tape) {
      if (i < 0) return p->noise_tape;
      const size_t off = ny + (size_t)i * (ny + (size_t)nph * nx) + (k == 0 ? 0 : ny + (size_t)(k - 1) * nx);
      return p->noise_tape + off;
    }
    const uint64_t stream = i < 0 ? 1 : (uint64_t)2 + (uint64_t)i * (1 + nph) + k;
    if (randn_launch(dst, (int64_t)n, p->seed, stream, s)) return nullptr;
    return dst;
  }
  const float* noise(int i, int k, float* dst, size_t n) const {
    if (p->noise_tape) {
      // tape layout per step and existing phase: [zy] z
      const size_t per_phase = nx + (perturb_y ? ny : 0);
      size_t off = (size_t)i * per_phase * (draws_per_step() / draws_per_phase());
      off += (size_t)(k / draws_per_phase()) * per_phase;
      if (perturb_y && (k % 2) == 1) off += ny;
      return p->noise_tape + off;
    }
    if (randn_launch(dst, (int64_t)n, p->seed, (uint64_t)1 + (uint64_t)i * draws_per_step() + k, s)) return nullptr;
    return dst;
  }
};

static int pc_setup(PCCtx* c, csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes, void* scratch,
                    size_t scratch_bytes, float* x, const float* y, int B, const csd_pc_params* p, void* stream) {
  Plan* pl = nullptr;
  int rc = check_forward_args(net, packed, workspace, workspace_bytes, B, &pl);
  if (rc) return rc;
  CSD_REQUIRE(p && x && scratch, "pc_sample: null argument");
  CSD_REQUIRE(p->n_steps >= 1 && p->labels && p->std_x, "pc_sample: per-step scalar arrays missing");
  CSD_REQUIRE(p->predictor >= 0 && p->predictor <= 2 && p->corrector >= 0 && p->corrector <= 2, "pc_sample: bad predictor / corrector id");
  CSD_REQUIRE(p->predictor != 0 || p->G, "pc_sample: the reverse-diffusion predictor needs G");
  CSD_REQUIRE(p->predictor != 1 || p->pred_coef, "pc_sample: predictor table missing");
  CSD_REQUIRE(p->corrector != 1 || p->corr_coef, "pc_sample: corrector table missing");
  CSD_REQUIRE(p->predictor != 2 || p->corrector != 2, "pc_sample: predictor and corrector are both 'none'");
  const csd_unet_config& cf = net->net.cfg;
  CSD_REQUIRE((cf.y_channels == 0) == (y == nullptr), "pc_sample: y must be given iff y_channels > 0");
  CSD_REQUIRE(!(p->std_y && cf.y_channels == 0), "pc_sample: std_y given for an unconditional network");
  if (scratch_bytes < csd_pc_scratch_bytes(net, B)) {
    set_error("pc_sample: scratch too small");
    return CSD_ERR_WORKSPACE;
  }
  CSD_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 255) == 0, "pc_sample: scratch must be 256-byte aligned");
  const size_t hw = (size_t)cf.image_size * cf.image_size;
  c->n = &net->net; c->pl = pl; c->pk = static_cast<const float*>(packed); c->ws = static_cast<float*>(workspace);
  c->p = p; c->x = x; c->y = y; c->B = B; c->s = (hipStream_t)stream;
  c->nx = (size_t)B * cf.x_channels * hw; c->ny = (size_t)B * cf.y_channels * hw;
  const size_t no = (size_t)B * cf.out_channels * hw;
  float* f = static_cast<float*>(scratch);
  c->net_out = f; f += align_up(no, 64);
  c->x_mean = f; f += align_up(c->nx, 64);
  c->z = f; f += align_up(c->nx, 64);
  c->zy = f; f += align_up(std::max(c->ny, (size_t)B * hw), 64);
  c->labels = f; f += align_up((size_t)B, 64);
  c->ystate = f; f += align_up(std::max(c->ny, (size_t)B * hw), 64);
  c->partial = reinterpret_cast<double*>(f);
  c->nonfinite = reinterpret_cast<int*>(c->partial + (size_t)B * 64 * 2);      // (csd_pc_scratch_bytes keeps 256 bytes behind the partials)
  c->path = p->path_coef != nullptr;
  CSD_REQUIRE(!c->path || (cf.y_channels > 0 && !p->std_y), "pc_sample: use_path needs a conditioning image and no marginal std_y");
  c->per = (int64_t)cf.x_channels * hw;
  c->nchunk = sumsq_nchunk(c->per);
  c->perturb_y = p->std_y != nullptr;
  // paired networks emit [score_x | score_y] per sample: the x block of sample b starts at b*out_channels*hw
  c->net_stride = (int64_t)cf.out_channels * hw;
  return CSD_OK;
}

// phase 0: corrector (sampling/conditional.py:208-209), phase 1: predictor (:211).  part bit 0: network + noise + (corrector:
// norm partials); bit 1: the update.  norm_sums != null: the corrector's step size comes from those two (all-reduced) sums.
static int pc_phase(const PCCtx& c, int i, int phase, int part, float* sums_out, const float* sums_in, int Bg) {
  int rc;
  const csd_pc_params* p = c.p;
  if (!c.has_phase(phase)) {                       // 'none': x stays, x_mean = x (sampling/predictors.py:182-200, correctors.py:145-163)
    if ((part & 2) && phase == (c.path ? 0 : 1) && i == p->n_steps - 1 && p->denoise)      // (the step's LAST phase)
      CSD_CHECK_HIP(hipMemcpyAsync(c.x_mean, c.x, c.nx * sizeof(float), hipMemcpyDeviceToDevice, c.s));
    return CSD_OK;
  }
  const int k0 = (phase == 1 && c.has_phase(0) ? 1 : 0) * c.draws_per_phase();
  const int kp = 1 + (phase == 0 && c.has_phase(1) ? 1 : 0);       // use_path: the predictor draws first
  const float* zp = c.p->noise_tape ? (c.path ? c.noise_path(i, kp, nullptr, c.nx) : c.noise(i, k0 + (c.perturb_y ? 1 : 0), nullptr, c.nx))
                                    : c.z;
  if (part & 1) {
    if (c.path ? (phase == 1 || !c.has_phase(1)) : (phase == 0 || !c.has_phase(0))) {      // (once per step: by the first phase that evaluates the network)
      hipLaunchKernelGGL(fill_labels_kernel, dim3(cdiv(c.B, 256)), dim3(256), 0, c.s, c.labels, p->labels[i], c.B);
      CSD_LAUNCH_CHECK();
    }
    const float* zyp = nullptr;
    if (c.perturb_y) { zyp = c.noise(i, k0, c.zy, c.ny); if (!zyp) return CSD_ERR_HIP; }
    rc = run_plan(*c.n, *c.pl, c.pk, c.ws, c.x, c.path ? c.ystate : c.y, c.labels, c.net_out, zyp, c.perturb_y ? p->std_y[i] : 0.f, c.s);
    if (rc) return rc;
    zp = c.path ? c.noise_path(i, kp, c.z, c.nx) : c.noise(i, k0 + (c.perturb_y ? 1 : 0), c.z, c.nx);
    if (!zp) return CSD_ERR_HIP;
    if (phase == 0 && p->corrector == 0) {
      ProfScope prof(CSD_PROF_SAMPLER, 0, 2.0 * c.nx * 4, c.s);
      if ((rc = sumsq_rows_launch(c.net_out, c.net_stride, zp, c.partial, c.B, c.per, c.nchunk, c.s))) return rc;
      if (sums_out && (rc = norm_sums_launch(c.partial, c.nchunk, p->std_x[i], c.B, sums_out, c.s))) return rc;
    }
  }
  if (part & 2) {
    ProfScope prof(CSD_PROF_SAMPLER, 0, 6.0 * c.nx * 4, c.s);
    const int rule = phase == 0 ? p->corrector : p->predictor;
    if (rule == 1) {                               // affine table: x_mean = p x + a score, x = x_mean + b z
      const float* co = (phase == 0 ? p->corr_coef : p->pred_coef) + (size_t)i * 3;
      rc = affine_net_update_launch(c.x, c.x_mean, c.net_out, c.net_stride, zp, p->std_x[i], co[0], co[1], co[2], c.B, c.per, c.s);
    } else if (phase == 0) {
      const float alpha = p->corr_alpha ? p->corr_alpha[i] : 1.0f;      // sde.alphas[timestep] (VP / subVP); 1 for the VE SDEs
      rc = sums_in ? langevin_update_global_launch(c.x, c.x_mean, c.net_out, c.net_stride, zp, sums_in, Bg, p->std_x[i], p->snr,
                                                   alpha, c.B, c.per, c.s, c.nonfinite)
                   : langevin_update_launch(c.x, c.x_mean, c.net_out, c.net_stride, zp, c.partial, c.nchunk, p->std_x[i],
                                            p->snr, alpha, c.B, c.per, c.s, c.nonfinite);
    } else {
      rc = reverse_diffusion_update_launch(c.x, c.x_mean, c.net_out, c.net_stride, zp, p->std_x[i], p->G[i], c.B, c.per, c.s);
    }
    if (rc) return rc;
  }
  return CSD_OK;
}

static int pc_step_tail(const PCCtx& c, int i) {
  const csd_pc_params* p = c.p;
  if (p->record)
    CSD_CHECK_HIP(hipMemcpyAsync(p->record + (size_t)i * c.nx, c.x, c.nx * sizeof(float), hipMemcpyDeviceToDevice, c.s));
  if (i == p->n_steps - 1 && p->denoise)
    CSD_CHECK_HIP(hipMemcpyAsync(c.x, c.x_mean, c.nx * sizeof(float), hipMemcpyDeviceToDevice, c.s));
  return CSD_OK;
}

// The finiteness contract of the fused loop (BASELINE.json north_star: outputs within 1e-3 of the reference - a NaN image is not):
// every Langevin step's norms and one pass over the returned state set a device flag; the loop's LAST call reads it back behind the
// stream (the only synchronisation of the sampler) and fails with CSD_ERR_NONFINITE instead of returning NaN images silently.
static int pc_finish(const PCCtx& c) {
  int rc = finite_check_launch(c.x, c.nx, c.nonfinite, c.s);
  if (rc) return rc;
  int flag = 0;
  CSD_CHECK_HIP(hipMemcpyAsync(&flag, c.nonfinite, sizeof(int), hipMemcpyDeviceToHost, c.s));
  CSD_CHECK_HIP(hipStreamSynchronize(c.s));
  if (flag) {
    set_error("pc_sample: the sampler's state or a corrector norm is not finite - an operand of an fp16-operand mode (fp16x3 / fp16f8 / fp16) "
              "left the fp16 range (65504); run this network with csd_precision = 'fp32'");
    return CSD_ERR_NONFINITE;
  }
  return CSD_OK;
}

extern "C" int csd_pc_sample(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes,
                             void* scratch, size_t scratch_bytes, float* x, const float* y, int B,
                             const csd_pc_params* p, void* stream) {
  PCCtx c;
  int rc = pc_setup(&c, net, packed, workspace, workspace_bytes, scratch, scratch_bytes, x, y, B, p, stream);
  if (rc) return rc;
  CSD_CHECK_HIP(hipMemsetAsync(c.nonfinite, 0, sizeof(int), c.s));
  if (c.path) {                                 // y_{T+tau} = y + sigma_y(T+tau) z (sampling/conditional.py:146-149)
    const float* z0 = c.noise_path(-1, 0, c.zy, c.ny);
    if (!z0) return CSD_ERR_HIP;
    if ((rc = bridge_update_launch(c.y, c.ystate, z0, 1.f, 0.f, p->path_std0, 0, c.ny, c.s))) return rc;
  }
  for (int i = 0; i < p->n_steps; ++i) {
    g_prof.step_on = (i % g_prof.step_stride) == 0;
    if (c.path) {                               // y_t from the bridge, predictor, then corrector on the same y_t (:151-170)
      const float* zy = c.noise_path(i, 0, c.zy, c.ny);
      if (!zy) return CSD_ERR_HIP;
      const float* co = p->path_coef + (size_t)i * 3;
      if ((rc = bridge_update_launch(c.y, c.ystate, zy, co[0], co[1], co[2], 1, c.ny, c.s))) return rc;
      for (int phase = 1; phase >= 0; --phase)
        if ((rc = pc_phase(c, i, phase, 3, nullptr, nullptr, 0))) return rc;
    } else {
      for (int phase = 0; phase < 2; ++phase)     // corrector, then predictor (sampling/conditional.py:208-211)
        if ((rc = pc_phase(c, i, phase, 3, nullptr, nullptr, 0))) return rc;
    }
    if ((rc = pc_step_tail(c, i))) return rc;
  }
  g_prof.step_on = true;
  return pc_finish(c);
}

extern "C" int csd_pc_step_begin(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes, void* scratch,
                                 size_t scratch_bytes, float* x, const float* y, int B, const csd_pc_params* p, int step,
                                 float* norm_sums, void* stream) {
  PCCtx c;
  int rc = pc_setup(&c, net, packed, workspace, workspace_bytes, scratch, scratch_bytes, x, y, B, p, stream);
  if (rc) return rc;
  CSD_REQUIRE(norm_sums && step >= 0 && step < p->n_steps, "pc_step_begin: bad step %d / null norm_sums", step);
  CSD_REQUIRE(!c.path, "pc_step_begin: use_path runs through csd_pc_sample only");
  if (step == 0) CSD_CHECK_HIP(hipMemsetAsync(c.nonfinite, 0, sizeof(int), c.s));
  return pc_phase(c, step, 0, 1, norm_sums, nullptr, 0);
}

extern "C" int csd_pc_step_end(csd_unet* net, const void* packed, void* workspace, size_t workspace_bytes, void* scratch,
                               size_t scratch_bytes, float* x, const float* y, int B, const csd_pc_params* p, int step,
                               const float* norm_sums, int global_batch, void* stream) {
  PCCtx c;
  int rc = pc_setup(&c, net, packed, workspace, workspace_bytes, scratch, scratch_bytes, x, y, B, p, stream);
  if (rc) return rc;
  CSD_REQUIRE(norm_sums && global_batch >= B && step >= 0 && step < p->n_steps, "pc_step_end: bad arguments");
  if ((rc = pc_phase(c, step, 0, 2, nullptr, norm_sums, global_batch))) return rc;
  if ((rc = pc_phase(c, step, 1, 3, nullptr, nullptr, 0))) return rc;
  if ((rc = pc_step_tail(c, step))) return rc;
  return step == p->n_steps - 1 ? pc_finish(c) : CSD_OK;
}
