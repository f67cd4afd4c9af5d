// attention.hip - single-head self-attention core of AttnBlock on the fp32 matrix cores.
//
// Replaces (reference, behaviour only) models/layers.py:584-588:
//     w = einsum('bchw,bcij->bhwij', q, k) * C**-0.5 ; softmax over (i,j) ; einsum(w, v)
// q/k/v arrive as one NHWC tensor [B, L, 3C] produced by a single fused 1x1 contraction
// (NIN_0|NIN_1|NIN_2 concatenated along Cout), L = H*W in {25,100,400} for SR3-160 and
// {16,...,256} for the other configs.  The L x L score matrix never leaves registers.
//
// One workgroup = 4 waves = 128 queries of one image; wave w owns 32 queries.  Keys/values are
// streamed in tiles of 32 through LDS (shared by the 4 waves).  Per tile and wave:
//   S^T[key][query] = K_tile . Q^T      v_mfma_f32_32x32x2_f32, A = K rows from LDS (b128),
//                                        B = Q rows straight from L1/L2 (b128)
//   online softmax in registers: with the swapped product every lane owns ONE query column and
//                                16 of the 32 keys; the other 16 sit in lane^32 (one shuffle).
//   O[query][c] += P . V_tile           the S^T accumulator registers ARE the A operand
//                                        (lane half <-> MFMA k index), B = V rows from LDS (b32).
// All arithmetic fp32; exp via expf; fp32-exact MFMA => matches the reference to round-off.
#include "common.h"

namespace csd {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define ATT_THREADS 256
#define ATT_KT 32            // keys per tile

template <int NCT>           // C / 32
__global__ __launch_bounds__(ATT_THREADS) void attention_kernel(const float* __restrict__ qkv, int ld,
                                                                float* __restrict__ out, int L,
                                                                float scale) {
  constexpr int C = NCT * 32;
  constexpr int KS = C + 4;  // LDS row stride (floats): (C+4) mod 64 in {4,36} -> conflict-free b128
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ks = smem;                 // [32][KS]
  float* const Vs = smem + ATT_KT * KS;   // [32][KS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const bool wave_active = q0 < L;                   // whole wave beyond L: only helps staging
  const int qrow = min(q0 + (lane & 31), L - 1);     // clamped (discarded at the store)
  const float* qptr = qkv + ((size_t)b * L + qrow) * ld + half * 4;
  const float* kvbase = qkv + (size_t)b * L * ld;

  floatx16 o[NCT];
#pragma unroll
  for (int n = 0; n < NCT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
  float m_run = -INFINITY;   // running max of this lane's query (same in both halves)
  float l_run = 0.f;         // running sum over THIS lane's keys only

  const int ntiles = (L + ATT_KT - 1) / ATT_KT;
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * ATT_KT;
    __syncthreads();         // previous tile fully consumed
    // ---- stage K and V tiles ----
    for (int e = tid; e < 2 * ATT_KT * (C / 4); e += ATT_THREADS) {
      const int which = e / (ATT_KT * (C / 4));
      const int r = e - which * ATT_KT * (C / 4);
      const int key = r / (C / 4);
      const int c4 = r - key * (C / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key0 + key < L)
        v = *reinterpret_cast<const float4*>(kvbase + (size_t)(key0 + key) * ld + (1 + which) * C + c4 * 4);
      *reinterpret_cast<float4*>((which ? Vs : Ks) + key * KS + c4 * 4) = v;
    }
    __syncthreads();
    if (!wave_active) continue;

    // ---- S^T = K . Q^T ----
    floatx16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* krow = Ks + (lane & 31) * KS + half * 4;
#pragma unroll 4
    for (int c0 = 0; c0 < C; c0 += 8) {
      const float4 a4 = *reinterpret_cast<const float4*>(krow + c0);
      const float4 b4 = *reinterpret_cast<const float4*>(qptr + c0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, s, 0, 0, 0);
    }
    // ---- online softmax (this lane: one query, keys key0 + (r&3)+8(r>>2)+4*half) ----
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      s[r] = (key < L) ? s[r] * scale : -INFINITY;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);            // finite: every tile holds >= 1 valid key
    const float alpha = expf(m_run - m_new);         // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = expf(s[r] - m_new);                     // masked keys: exp(-inf) = 0
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // rescale O: its rows are queries (r&3)+8(r>>2)+4*half, alpha lives in lane == query
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float ar = __shfl(alpha, row);
#pragma unroll
      for (int n = 0; n < NCT; ++n) o[n][r] *= ar;
    }
    // ---- O += P . V : A operand = s[r] as is ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* vrow = Vs + ((r & 3) + 8 * (r >> 2) + 4 * half) * KS + (lane & 31);
#pragma unroll
      for (int n = 0; n < NCT; ++n)
        o[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], vrow[n * 32], o[n], 0, 0, 0);
    }
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    const float lr = __shfl(l_tot, row);
    const int q = q0 + row;
    if (q < L) {
      float* dst = out + ((size_t)b * L + q) * C + (lane & 31);
#pragma unroll
      for (int n = 0; n < NCT; ++n) dst[n * 32] = o[n][r] / lr;
    }
  }
}

template <int NCT>
static int launch_att(const float* qkv, int ld, float* out, int B, int L, hipStream_t s) {
  constexpr int C = NCT * 32;
  const size_t lds = (size_t)2 * ATT_KT * (C + 4) * sizeof(float);
  auto kern = attention_kernel<NCT>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const float scale = 1.0f / sqrtf((float)C);   // replaced below by the reference's expression
  (void)scale;
  const float ref_scale = (float)pow((double)C, -0.5);   // int(C) ** (-0.5) in Python (double) -> fp32 mul
  hipLaunchKernelGGL(kern, dim3(cdiv(L, 128), B), dim3(ATT_THREADS), lds, s, qkv, ld, out, L, ref_scale);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

// ---------------------------------------------------------------------------------------------
// The same attention core on the fp16 matrix cores for the fp16 arithmetic modes (NS = 2: operands split hi + lo, 3 MFMAs per
// product - fp32-class results; NS = 1: plain fp16 operands).  v_mfma_f32_32x32x2_f32 needs 64 cycles per 32 x 32 x 2 block, so
// the fp32 kernel above spends 12.3 k matrix cycles per 32-key tile and wave at C = 192 - 194 us per AttnBlock at 20 x 20, B = 64;
// v_mfma_f32_32x32x16_f16 does K = 16 in 32 cycles: 2.3 k cycles with the three-product split.
//   S^T[key][query] = K . (Q * C^-1/2)^T : A = K rows from LDS ([key][channel] fp16 planes, 16-byte fragments), B = the wave's Q
//                                          fragments, converted once and kept in registers
//   softmax as above (fp32, one query per lane, the other half of its keys in lane ^ 32)
//   O[query][c] += P . V                 : A = P from the S^T accumulator registers (lane half <-> K block; the 8 values of a
//                                          fragment are keys (j & 3) + 8 (j >> 2) + 16 t + 4 kb), B = V^T from LDS ([channel][key]
//                                          fp16 planes, keys stored in that fragment order: one ds_read_b128 per fragment)
// K / V tiles: fp32 rows -> registers one tile ahead (in flight under the MFMAs of the current tile) -> split -> LDS.
// ---------------------------------------------------------------------------------------------
typedef _Float16 half8a __attribute__((ext_vector_type(8)));
typedef _Float16 half4a __attribute__((ext_vector_type(4)));

// CS: waves that share a block of 32 queries, each owning 1/CS of the output channels (C >= 256: the Q fragments + ALL the O accumulators
// of a wave would not fit the 512 registers; S^T is then computed by both waves of a pair)
template <int NCT, int NS, int CS>
__global__ __launch_bounds__(ATT_THREADS) void attention16_kernel(const float* __restrict__ qkv, int ld,
                                                                  float* __restrict__ out, int L, float scale) {
  constexpr int C = NCT * 32, KSTEPS = C / 16;
  constexpr int KS = 2 * C + 16;           // bytes per key row of a K plane ((C/2 + 4) mod 64 dwords: conflict-free b128 fragments)
  constexpr int VS = 80;                   // bytes per channel row of a V^T plane (32 keys + 16 pad)
  constexpr int KPL = ATT_KT * KS, VPL = C * VS;
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  char* const Kp = smem16;                 // NS planes
  char* const Vp = smem16 + NS * KPL;      // NS planes

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kb = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y;
  constexpr int NOT = (NCT + CS - 1) / CS;      // output channel tiles of this wave: n0 .. n0 + NOT - 1 (those below NCT)
  const int n0 = (wave % CS) * NOT;
  const int q0 = blockIdx.x * (128 / CS) + (wave / CS) * 32;
  const bool wave_active = q0 < L;
  const float* kvbase = qkv + (size_t)b * L * ld;

  // ---- this wave's Q fragments: lane = (query l32, K block kb): channels 16 ks + 8 kb .. + 7, pre-multiplied by C^-1/2 ----
  half8a qh[KSTEPS], ql[NS == 2 ? KSTEPS : 1];
  {
    const float* qptr = kvbase + (size_t)min(q0 + l32, L - 1) * ld + kb * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const float4 a = *reinterpret_cast<const float4*>(qptr + ks * 16);
      const float4 c = *reinterpret_cast<const float4*>(qptr + ks * 16 + 4);
      const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, c.x * scale, c.y * scale, c.z * scale, c.w * scale};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        qh[ks][j] = h;
        if (NS == 2) ql[ks][j] = (_Float16)(v[j] - (float)h);
      }
    }
  }

  floatx16 o[NOT];
#pragma unroll
  for (int n = 0; n < NOT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging.  K: item i of a thread = float4 c4 of key `key` (c4 fastest: 8-byte hi / lo stores to consecutive LDS addresses).
  // V: a thread takes a 4 key x 4 channel block (keys 4m .. 4m+3 are consecutive slots of the fragment order), transposes it in
  // registers and stores 4 keys of one channel with one ds_write_b64; lanes = (m = lane & 7, c4 = lane >> 3 ...): the 16 lanes of a
  // store group cover all banks (a c4-major lane order puts 16 lanes on 2 banks: measured 12 k cycles per tile), and each load
  // instruction still reads whole 128-byte lines (8 keys x 8 consecutive float4).
  constexpr int NK = NCT;                                       // 32 keys x C/4 float4 over 256 threads
  constexpr int NVG = (2 * C + ATT_THREADS - 1) / ATT_THREADS;  // 8 key groups x C/4 channel groups
  float4 prek[NK], prev[NVG][4];
  auto prefetch = [&](int key0) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int e = tid + i * ATT_THREADS;
      const int key = e / (C / 4), c4 = e - key * (C / 4);
      prek[i] = key0 + key < L ? *reinterpret_cast<const float4*>(kvbase + (size_t)(key0 + key) * ld + C + c4 * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
      const int g = tid + i * ATT_THREADS;
      const int m = g & 7, c4 = min(g >> 3, C / 4 - 1);          // (threads past the last group reload it; their stores are skipped)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int key = key0 + 4 * m + u;
        prev[i][u] = key < L ? *reinterpret_cast<const float4*>(kvbase + (size_t)key * ld + 2 * C + c4 * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int e = tid + i * ATT_THREADS;
      const int key = e / (C / 4), c4 = e - key * (C / 4);
      const float v[4] = {prek[i].x, prek[i].y, prek[i].z, prek[i].w};
      half4a hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi[j] = (_Float16)v[j];
        lo[j] = (_Float16)(v[j] - (float)hi[j]);
      }
      *reinterpret_cast<half4a*>(Kp + key * KS + c4 * 8) = hi;
      if (NS == 2) *reinterpret_cast<half4a*>(Kp + KPL + key * KS + c4 * 8) = lo;
    }
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
      const int g = tid + i * ATT_THREADS;
      if (g >= 2 * C) continue;
      const int m = g & 7, c4 = g >> 3;
      // fragment order of the keys: key = (j & 3) + 8 (j >> 2) + 16 t + 4 kb  ->  slot 16 t + 8 kb + j; keys 4m .. 4m+3 -> 4 consecutive slots
      const int w = (4 * m) & 15;
      const int pos = ((4 * m) & 16) + 8 * ((w >> 2) & 1) + 4 * (w >> 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v[4] = {j == 0 ? prev[i][0].x : j == 1 ? prev[i][0].y : j == 2 ? prev[i][0].z : prev[i][0].w,
                            j == 0 ? prev[i][1].x : j == 1 ? prev[i][1].y : j == 2 ? prev[i][1].z : prev[i][1].w,
                            j == 0 ? prev[i][2].x : j == 1 ? prev[i][2].y : j == 2 ? prev[i][2].z : prev[i][2].w,
                            j == 0 ? prev[i][3].x : j == 1 ? prev[i][3].y : j == 2 ? prev[i][3].z : prev[i][3].w};
        half4a hi, lo;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          hi[u] = (_Float16)v[u];
          lo[u] = (_Float16)(v[u] - (float)hi[u]);
        }
        *reinterpret_cast<half4a*>(Vp + (c4 * 4 + j) * VS + pos * 2) = hi;
        if (NS == 2) *reinterpret_cast<half4a*>(Vp + VPL + (c4 * 4 + j) * VS + pos * 2) = lo;
      }
    }
  };

  // (C >= 256: Q fragments + O accumulators + a tile in flight exceed the 512 registers - those layers sit at the 10^2 / 5^2 levels,
  // <= 4 key tiles per image, and load each tile right before it is split)
  constexpr bool AHEAD = NCT <= 6;
  const int ntiles = (L + ATT_KT - 1) / ATT_KT;
  if (AHEAD) prefetch(0);
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * ATT_KT;
    if (!AHEAD) prefetch(key0);
    __syncthreads();         // previous tile fully consumed
    commit();
    __syncthreads();
    if (AHEAD && t + 1 < ntiles) prefetch(key0 + ATT_KT);
    if (!wave_active) continue;

    // ---- S^T = K . Q^T ----
    floatx16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const char* krow = Kp + l32 * KS + kb * 16;
    // fragments one step ahead, and no further (the scheduler would otherwise hoist all 2 * KSTEPS reads: 144 registers at C = 288)
    half8a ah_n = *reinterpret_cast<const half8a*>(krow), al_n;
    if (NS == 2) al_n = *reinterpret_cast<const half8a*>(krow + KPL);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const half8a ah = ah_n, al = al_n;
      if (ks + 1 < KSTEPS) {
        ah_n = *reinterpret_cast<const half8a*>(krow + (ks + 1) * 32);
        if (NS == 2) al_n = *reinterpret_cast<const half8a*>(krow + KPL + (ks + 1) * 32);
      }
      if (NS == 2) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], s, 0, 0, 0);
      }
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], s, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- online softmax (this lane: one query, keys key0 + (r&3)+8(r>>2)+4*kb) ----
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      s[r] = (key < L) ? s[r] : -INFINITY;
      mt = fmaxf(mt, s[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);            // finite: every tile holds >= 1 valid key
    const float alpha = __expf(m_run - m_new);       // first tile: exp(-inf) = 0 (v_exp_f32: ~1 ulp, as the SiLU of the conv loaders)
    float psum = 0.f;
    half8a ph[2], pl[NS == 2 ? 2 : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __expf(s[r] - m_new);         // masked keys: exp(-inf) = 0
      psum += pv;
      const _Float16 h = (_Float16)pv;
      ph[r >> 3][r & 7] = h;
      if (NS == 2) pl[r >> 3][r & 7] = (_Float16)(pv - (float)h);
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // rescale O: its rows are queries (r&3)+8(r>>2)+4*kb, alpha lives in lane == query (skipped when no query's maximum moved)
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ar = __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * kb);
#pragma unroll
        for (int n = 0; n < NOT; ++n) o[n][r] *= ar;
      }
    }
    // ---- O += P . V ----
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const char* vrow = Vp + (n0 * 32 + l32) * VS + (16 * tt + 8 * kb) * 2;
      half8a vh_n = *reinterpret_cast<const half8a*>(vrow), vl_n;
      if (NS == 2) vl_n = *reinterpret_cast<const half8a*>(vrow + VPL);
#pragma unroll
      for (int n = 0; n < NOT; ++n) {
        if (CS > 1 && n0 + n >= NCT) break;          // (uniform: the last wave of a pair may own one tile less)
        const half8a vh = vh_n, vl = vl_n;
        if (n + 1 < NOT && (CS == 1 || n0 + n + 1 < NCT)) {
          vh_n = *reinterpret_cast<const half8a*>(vrow + (n + 1) * 32 * VS);
          if (NS == 2) vl_n = *reinterpret_cast<const half8a*>(vrow + VPL + (n + 1) * 32 * VS);
        }
        if (NS == 2) {
          o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[tt], vh, o[n], 0, 0, 0);
          o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[tt], vl, o[n], 0, 0, 0);
        }
        o[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[tt], vh, o[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
    const float lr = __shfl(l_tot, row);
    const int q = q0 + row;
    if (q < L) {
      float* dst = out + ((size_t)b * L + q) * C + n0 * 32 + l32;
#pragma unroll
      for (int n = 0; n < NOT; ++n)
        if (CS == 1 || n0 + n < NCT) dst[n * 32] = o[n][r] / lr;
    }
  }
}

template <int NCT, int NS>
static int launch_att16(const float* qkv, int ld, float* out, int B, int L, hipStream_t s) {
  constexpr int C = NCT * 32;
  const size_t lds = (size_t)NS * (ATT_KT * (2 * C + 16) + C * 80);
  constexpr int CS = NCT >= 8 ? 2 : 1;
  auto kern = attention16_kernel<NCT, NS, CS>;
  static bool attr_set = false;
  if (!attr_set) {
    CSD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const float ref_scale = (float)pow((double)C, -0.5);   // int(C) ** (-0.5) in Python (double) -> fp32
  hipLaunchKernelGGL(kern, dim3(cdiv(L, 128 / CS), B), dim3(ATT_THREADS), lds, s, qkv, ld, out, L, ref_scale);
  CSD_LAUNCH_CHECK();
  return CSD_OK;
}

int attention16_launch(const float* qkv, int ld, float* out, int B, int L, int C, int ns, hipStream_t s) {
  CSD_REQUIRE(C % 32 == 0, "attention: C=%d must be a multiple of 32", C);
  CSD_REQUIRE(L >= 1, "attention: empty sequence");
  CSD_REQUIRE(ns == 1 || ns == 2, "attention16: %d operand planes", ns);
#define CSD_ATT16_CASE(N) case N: return ns == 2 ? launch_att16<N, 2>(qkv, ld, out, B, L, s) : launch_att16<N, 1>(qkv, ld, out, B, L, s);
  switch (C / 32) {
    CSD_ATT16_CASE(1) CSD_ATT16_CASE(2) CSD_ATT16_CASE(3) CSD_ATT16_CASE(4) CSD_ATT16_CASE(6) CSD_ATT16_CASE(8) CSD_ATT16_CASE(9)
    default:
      set_error("attention: C=%d not instantiated (supported: 32,64,96,128,192,256,288)", C);
      return CSD_ERR_INVALID;
  }
#undef CSD_ATT16_CASE
}

int attention_launch(const float* qkv, int ld, float* out, int B, int L, int C, hipStream_t s) {
  CSD_REQUIRE(C % 32 == 0, "attention: C=%d must be a multiple of 32", C);
  CSD_REQUIRE(L >= 1, "attention: empty sequence");
  switch (C / 32) {
    case 1: return launch_att<1>(qkv, ld, out, B, L, s);
    case 2: return launch_att<2>(qkv, ld, out, B, L, s);
    case 3: return launch_att<3>(qkv, ld, out, B, L, s);
    case 4: return launch_att<4>(qkv, ld, out, B, L, s);
    case 6: return launch_att<6>(qkv, ld, out, B, L, s);
    case 8: return launch_att<8>(qkv, ld, out, B, L, s);
    case 9: return launch_att<9>(qkv, ld, out, B, L, s);
    default:
      set_error("attention: C=%d not instantiated (supported: 32,64,96,128,192,256,288)", C);
      return CSD_ERR_INVALID;
  }
}

}  // namespace csd
