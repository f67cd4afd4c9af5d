// conv_ff.h - definitions of the fused-prologue block convolution (conv_ff.hip: 4-wave workgroups, two per CU): tile geometry,
// LDS patch layout, ring configuration, kernel arguments, the LDS hand-over barrier.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "conv_f16_kernel.h"

namespace csd {

typedef unsigned int uint4f __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));
typedef short short2v __attribute__((ext_vector_type(2)));

#define FF_THREADS 256
#define FF_TILE 16
#define FF_PW 18                                   // patch width / height in pixels
#define FF_NPATCH (FF_PW * FF_PW)
#define FF_PSB 64                                  // bytes per staged pixel: NS = 1: 32 channels fp16; NS = 2: 16 ch hi | 16 ch lo
#define FF_RS (FF_PW * FF_PSB + 16)                // LDS row pitch (conflict-free ds_read_b128 for 4 x 8 M tiles)
#define FF_PATCH_BYTES (FF_PW * FF_RS)

struct ConvFFArgs {
  ConvArgs a;
  int B, H, W, C0, C1, Cout;
  int tiles_x, tpi, n_groups, nblocks;
  int nstage;
  int abl;                   // tuning aid (CSD_FF_ABL): 1 no weight refills, 2 no patch conversion, 4 no patch prefetch, 8 no epilogue IO
};

// LDS hand-over: this wave's ds_writes / ds_reads have completed (lgkmcnt), then the workgroup barrier.  NOT a fence: with an
// LDS-DMA in flight a workgroup release fence makes hipcc drain vmcnt(0) in every wave at every barrier - which would expose
// the patch prefetch and the weight stream once per ring group.  LDS-DMA data is ordered by wave 0's counted vmcnt before
// its barrier arrival instead.
__device__ __forceinline__ void ff_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0) only (a builtin, so hipcc's own wait bookkeeping sees it)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void ff_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NS, int NT>
struct FFCfg {
  static constexpr int KC = NS == 1 ? 32 : 16;                // channels per stage
  static constexpr int KSUB = NS == 1 ? 2 : 1;                // MFMA K steps per tap per stage
  static constexpr int STEPS = KSUB * 9;                      // K steps per stage
  static constexpr int TG = NS == 1 ? 2 : 1;                  // steps per ring group (6 KiB at NT = 3)
  static constexpr int GPS = STEPS / TG;                      // ring groups per stage
  static constexpr int SB = NT * NS * 1024;                   // weight bytes per step
  static constexpr int GB = TG * SB;                          // bytes per ring group
  static constexpr int GL = GB / 1024;                        // LDS-DMA instructions per group
  static constexpr int R = 5;                                 // ring depth (groups): current, next (landed), two in flight, one being refilled
  static constexpr int G4 = KC / 4;                           // 4-channel groups per pixel per stage
  // waves 0..NDMA-1 stream the weights (an LDS-DMA costs its wave 60-180 cycles of issue: six per ring group on ONE wave made
  // that wave the pace of the workgroup), the others prefetch + convert the patch
  static constexpr int NDMA = 1;      // (2 measured slower in the split mode: 11 conversion slots on two waves outweigh the DMA relief)
  static constexpr int GLW = GL / NDMA;                       // LDS-DMA instructions per group per DMA wave
  static constexpr int LOADERS = (4 - NDMA) * 64;             // threads of the patch waves
  static constexpr int NSLOT = (FF_NPATCH * G4 + LOADERS - 1) / LOADERS;
  static constexpr int PPJ = LOADERS / G4;                    // patch pixels between a thread's consecutive slots
  static constexpr size_t LDS = 2 * (size_t)FF_PATCH_BYTES + (size_t)R * GB + 2 * FF_NPATCH * sizeof(int);
};

// conv_fx.hip: the fp16f8 form with one workgroup per CU (whole-stage weight buffers, one barrier per stage)
int convfx_launch(const ConvFFArgs& k, int nt, hipStream_t s);
// conv_xk.hip: the fp16x3 form - 1-D Winograd F(2,3), one software-pipelined stream per SIMD, one transform component per wave
bool convxk_supported(const ConvFFArgs& k, int nt);
int convxk_launch(const ConvFFArgs& k, int nt, hipStream_t s);

}  // namespace csd
