// fp32 3x3 convolution on the bf16 matrix pipes of gfx950: every fp32 operand is written EXACTLY as the sum of three bf16
// terms, x = xh + xm + xl (8 + 8 + 8 significand bits; bf16 has the fp32 exponent range), and the product of two operands is
// accumulated in fp32 from the six partial products of order <= 2^-16:
//     x * w  ~=  xh*wh + xh*wm + xm*wh + xm*wm + xh*wl + xl*wh          (dropped: xm*wl + xl*wm + xl*wl  <=  2^-23 |x w|)
// Each partial product of two bf16 values is exact in fp32, so the only rounding is the fp32 accumulation inside
// v_mfma_f32_32x32x16_bf16.  That accumulation is coarser than an fp32 fma chain (measured: with all six products in one
// accumulator the error against an fp64 convolution is 2.4x the fp32 MFMA kernel's, archive/profiles/r3_bf16x3_first_run.txt), so the
// five small products (<= 2^-8 of the leading one) go to a SECOND accumulator set whose rounding is 2^-8 smaller, and the
// leading product's accumulator sees one MFMA per 16 channels and tap -- 8x fewer accumulation steps than the fp32 kernel's
// K = 2 instructions; the two sets are added once in the epilogue.  The dropped terms are 30x below the accumulation error
// (tools/split_accuracy.py, CPU); tests/test_conv_bf16x3_gpu.py holds the kernel to the fp32 MFMA kernel's error against
// fp64 (measured 1.7e-7 against 2.3e-7).  The bf16 pipes run at 16x the fp32 MFMA rate: six products cost 6/16 of the fp32
// kernel's matrix time.  NOT a reduced-precision mode: tensors, norms, epilogue and accumulation are fp32 and the operands
// keep all 24 significand bits.
//
// Structure (one block = 64 output channels x 256 positions = a 4 x 64 pixel tile, 4 waves of 64 x 64, ONE block per CU:
// 150 KB of LDS, 512 registers per lane):
//   * operand planes in LDS, bf16, 16 bytes = 8 channels per slot (both MFMA operands are one ds_read_b128 per lane):
//       weights  W[kernel row][plane][tap column s][half][64][8]   16 input channels; each row buffer is refilled for the next
//                                                           channel group by LDS-DMA (host-packed, host-split tensor) as soon
//                                                           as its last fragments have been read: two rows of latency budget
//       patch    P[2][plane][8-channel group][slot][8]      16 input channels of the (4+2) x (64+2) source patch, split on
//                                                           the way in (16-byte quads; the halo pixels ride on the lanes
//                                                           that own no quad; slots as in conv_igemm_f16.h)
//   * K loop over channel groups cg (16 channels [x depth tap]) x kernel rows r x tap columns s; one step (cg, r, s) =
//     12 ds_read_b128 (3 planes x (2 weight + 2 patch fragments)) feeding 24 MFMAs (4 tiles x 6 products) = 768 matrix
//     cycles per SIMD; fragments are read one step ahead into three rotating register sets.
//   * one barrier per kernel row, placed between its steps 1 and 2 ("MIDBAR"): at that point every wave has read the last
//     weight fragments of row r (they are fetched one step ahead), so W[r] is free for the DMA of (cg + 1, r), and the DMA of
//     the next row (issued two rows ago) is waited for right there -- the fragments of its step 0 are prefetched behind it.
//   * the patch of group cg + 1 is converted during (cg, row 0, step 2) .. (cg, row 1, step 1) from registers loaded during
//     cg - 1; the loads of cg + 2 are issued behind MIDBAR(cg, 1).  vmcnt bookkeeping (all VMEM of the loop is inline asm;
//     every wave issues exactly 5 DMA instructions per row and 8 quad loads per stage):
//         behind MIDBAR(cg,0): DMA(cg+1,0)    behind MIDBAR(cg,1): DMA(cg+1,1), Q(cg+2)    behind MIDBAR(cg,2): DMA(cg+1,2)
//         MIDBAR(cg,0) needs DMA(cg,1) and Q(cg+1), leaves DMA(cg,2) in flight: vmcnt(5)
//         MIDBAR(cg,1) needs DMA(cg,2), leaves DMA(cg+1,0): vmcnt(5)
//         MIDBAR(cg,2) needs DMA(cg+1,0), leaves DMA(cg+1,1) and Q(cg+2): vmcnt(13)
// SPLIT = 2 (opt-in, emo_conv_igemm_f16x2): the same kernel on v_mfma_f32_32x32x16_f16 with TWO fp16 terms of the scaled
// operand -- x * in_scale = x1 + x2 (+ <= 2^-24 relative; in_scale a power of two folded into the producer's affine, weights
// scaled per layer on the host) -- and the THREE products x1 w1 + x1 w2 + x2 w1 (dropped: x2 w2 <= 2^-24), result multiplied
// by 1 / (in_scale * w_scale) when the accumulator sets are combined.  Half the matrix work and two thirds of the LDS planes;
// measured 279-359 TF fp32-equivalent against 189-226 (archive/profiles/r3_conv_microbench.jsonl), error against an fp64 convolution
// 1.09x the fp32 MFMA kernel's, end-to-end parity figures those of SPLIT = 3.  Its contract is narrower -- |x * in_scale|
// saturates at 65504 (inputs beyond +-2047 after norm + ReLU), terms below 2^-14 / scale lose relative (not absolute)
// precision -- which is why the exact SPLIT = 3 is the default.
// Covers 3x3 (and 3x3x3 with the depth taps as K stages) layers on maps whose width is a multiple of 64 and height a
// multiple of 4 (4 x 64 pixel tiles; 8 x 32 on 32-wide maps), optional fused nearest x2 upsample, Cin % 8 == 0; epilogue, K split and GroupNorm tile statistics are the
// shared conv_epilogue.  Anything else runs conv_igemm.h.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "conv_igemm_f16.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));

// The fp16 split of two values at once: (h, m) planes with h = fp16(v), m = fp16(v - h), round to nearest even at both levels --
// the bits of `(_Float16)v` and `(_Float16)(v - (float)h)`.  Written on PAIRS so that each plane costs one v_cvt_pk_f16_f32 per
// two values: from the scalar form the compiler derives the residual from a v_cvt_f16_f32 of its own and converts the same
// value a second time to build the packed operand (ISA of round 5's first builds: 6.5 vector instructions per element of the
// patch, 5.5 here; the subtractions stay scalar -- a packed one costs more beside an MFMA stream than the two it replaces).
#ifndef EMO_S_PAIR_CVT
#define EMO_S_PAIR_CVT 1      // 0: the scalar form (A/B builds only; the same bits)
#endif
template <typename V8>
__device__ __forceinline__ void emo_split_f16x2_pair(float v0, float v1, V8& h, V8& m, int u) {
#if EMO_S_PAIR_CVT
  const halfx2 h2 = __builtin_convertvector(floatx2{v0, v1}, halfx2);
  const float r0 = v0 - (float)h2[0], r1 = v1 - (float)h2[1];
  const halfx2 m2 = __builtin_convertvector(floatx2{r0, r1}, halfx2);
  h[u] = h2[0]; h[u + 1] = h2[1];
  m[u] = m2[0]; m[u + 1] = m2[1];
#else
  h[u] = (_Float16)v0; m[u] = (_Float16)(v0 - (float)h[u]);
  h[u + 1] = (_Float16)v1; m[u + 1] = (_Float16)(v1 - (float)h[u + 1]);
#endif
}

#ifndef EMO_S_PIN
#define EMO_S_PIN 1   /* 0: A/B switch -- the compiler schedules the inside of a step on its own */
#endif
#ifndef EMO_S_ABLATE
#define EMO_S_ABLATE 0   /* timing experiments only (results are WRONG for any value != 0), bits: 1 = no weight DMA in the K loop,
                            2 = no s_barrier in the K loop (the waitcnt stays), 4 = no quad loads in the K loop */
#endif
#ifndef EMO_S_TIMING
#define EMO_S_TIMING 0   /* measurement builds only (tools/conv_phase_timing.py): wave 0 of every work item logs s_memtime at the start
                            of its prologue, K loop, epilogue, after the epilogue's last instruction and after its stores have
                            drained, with HW_ID / XCC_ID, into a per-translation-unit device array read by emo_debug_conv_timing_* */
#endif
#if EMO_S_TIMING
#define EMO_S_TLOG_N 65536
#define EMO_S_TLOG_W 16
static __device__ unsigned long long emo_s_tlog[EMO_S_TLOG_N * EMO_S_TLOG_W];
#define EMO_S_STAMP(k_) if (EMO_S_TIMING) { tstamp[k_] = __builtin_amdgcn_s_memtime(); }
#define EMO_S_TSTAMP_ARG , tstamp
#else
#define EMO_S_STAMP(k_)
#define EMO_S_TSTAMP_ARG
#endif
#ifndef EMO_S_CHAIN
#define EMO_S_CHAIN 1   /* fp16 split: consecutive items of a persistent block run through one pipeline (0: A/B builds) */
#endif
#ifndef EMO_S_STAGGER
#define EMO_S_STAGGER 0
#endif
#ifndef EMO_S_PRODUCTS
#define EMO_S_PRODUCTS 6   /* measurement builds: 3 = (h,h) (h,m) (m,h) only (error 2^-16: NOT fp32-equivalent), 1 = plain bf16 */
#endif

// SPLIT = 3: bf16 x 3 terms, 6 products (exact operands).  SPLIT = 2: fp16 x 2 terms of the SCALED operand, 3 products (header
// comment at the end of this file's introduction): half the matrix work, operands to 2^-24 relative inside +-65504 / in_scale.
// BMT: output channels per tile, 64 -- or 32 (fp16 split only, round 5): a layer with 32 output channels ran the 64-row tile half
// empty (the two 32-channel 3-D layers of the WarpGenerator: 160-186 TF); with a 32-row tile a wave's tile is 32 x 64 -- half the
// MFMAs, half the weight stage (18 DMA pieces: five per wave, the last two re-copied by waves 2, 3)
template <int TR, int TW, bool UPS, int SPLIT = 3, int BMT = 64>
struct ConvCfgS {
  static constexpr int BM = BMT, BP = 256, TM = BMT / 32, TP = 2, WGP = 4, KC = 16;
  static_assert(BMT == 64 || (BMT == 32 && SPLIT == 2), "32-row channel tiles: the one-barrier (fp16-split) schedule only");
  static constexpr int NPL = SPLIT;                      // operand planes
  static constexpr int NPROD = SPLIT == 3 ? 6 : 3;       // partial products per fp32 product
  static constexpr int TRS = UPS ? TR / 2 : TR;          // tile extent in SOURCE pixels
  static constexpr int TWS = UPS ? TW / 2 : TW;
  static constexpr int PR = TRS + 2;                     // source rows of the patch
  static constexpr int NQ = TWS / 4;                     // interior quads per row
  static constexpr int NQ1 = NQ + 1;                     // + the pseudo-quad that holds the row's two halo pixels
  static constexpr int SUB = ((PR * NQ1 + 4 + 11) / 16) * 16 + 4;   // slots per sub-row: = 4 (mod 16), >= PR * NQ1 used + 4 dump
                                                         // slots at its tail (pixels a lane loads but does not own)
  static constexpr int CHS = 4 * SUB;                    // slots per 8-channel group
  static constexpr int NG = 2;                           // 8-channel groups per stage
  static constexpr int QPG = 128;                        // threads per group
  static constexpr int NHQ = 2 * PR;                     // halo pixels of a group: they ride on the group's lanes that own no quad
  // everything below in 16-byte slots
  static constexpr int WPLANE = 3 * 2 * BM;              // one plane of a kernel row: [s][half][BM]
  static constexpr int WROW = NPL * WPLANE;              // one kernel row (all planes)
  static constexpr int WROW_BYTES = WROW * 16;
  static constexpr int PPL = NG * CHS;                   // one plane of the patch
  static constexpr int PBUF = NPL * PPL;
  static constexpr int OFF_P = 0;                        // the two patch buffers first: with the buffer a compile-time constant
                                                         // (stage loop unrolled by two) every fragment read but one plane's is
                                                         // lane base + immediate offset (ds_read offsets reach 64 KiB)
  static constexpr int OFF_W = 2 * PBUF;                 // the kernel-row weight buffers: three rows, recycled row by row behind
                                                         // three barriers per stage (NWB = 1) -- or, where LDS has the room (the
                                                         // two-plane fp16 split), two whole stages: ONE barrier per stage
  static constexpr int NWB = SPLIT == 2 ? 2 : 1;
  static constexpr int WSTAGE = 3 * WROW;                // one stage of weights (three kernel rows)
  static constexpr int OFF_SCT = OFF_W + NWB * WSTAGE;   // scale / shift tables (fp32)
  static constexpr int SCT = 1024;
  // epilogue (conv_epilogue_rows): per-block bias table, the GroupNorm (mean, M2) exchange, and per wave a [32 channels][64
  // positions] fp32 transposition scratch.  The scratch lives in the patch buffer the last stage has just finished with when
  // that is large enough, in a region of its own otherwise
  static constexpr int EPI_ROWF = 68;                    // floats per channel row: 64 positions + 4 (bank spread of the b128 stores)
  static constexpr int EPI_WAVE = 32 * EPI_ROWF;         // floats per wave
  static constexpr int OFF_BIAS_F = OFF_SCT * 4 + 2 * SCT;            // (float index) [BM], permuted for the row layout
  static constexpr int OFF_STAT_F = OFF_BIAS_F + BM;                   // [WGP][BM][2]
  static constexpr int OFF_EPI_F = OFF_STAT_F + 2 * WGP * BM;
  static constexpr bool EPI_IN_PATCH = PBUF * 4 >= WGP * EPI_WAVE;
  // (both patch buffers together: they are adjacent and both idle behind the K loop's closing barrier)
  // CHAIN (kernel comment "work items"): the one-barrier schedule only.  A chained item transposes through the weight stage
  // buffer its last stage read (W[1]: the other one already holds the next item's first stage); an unchained one through the
  // idle stage buffer, which holds the dead re-staged rows
  static constexpr bool CHAIN = EMO_S_CHAIN && NWB == 2;
  static constexpr bool EPI_IN_W = !EPI_IN_PATCH && NWB == 2 && WSTAGE * 4 >= WGP * EPI_WAVE;
  // (a chained epilogue needs scratch that the next item's first stage does not occupy: the free weight stage buffer, or -- the
  // 32-row tile, whose weight stage is too small for it -- a region of its own)
  static constexpr int LDS_BYTES = (OFF_EPI_F + ((EPI_IN_PATCH || EPI_IN_W) ? 0 : WGP * EPI_WAVE)) * 4;
  // LDS-DMA instructions EVERY wave issues per kernel row (1 KiB each).  3 planes: 18 pieces, waves 2, 3 re-copy pieces 16,
  // 17 (uniform vmcnt counts); 2 planes: 12 pieces, 3 per wave
  static constexpr int NDMA = SPLIT == 3 ? 5 : 3;
  // one-barrier schedule: 1 KiB pieces of a whole stage (three kernel rows, contiguous) per wave -- piece k of wave w is piece
  // w + 4 k of the stage; 36 pieces: nine per wave; 18 pieces (32-row tile): five, the fifth = piece 16 + (w & 1)
  static constexpr int NSTP = SPLIT == 3 ? 0 : (BMT == 64 ? 9 : 5);
  static_assert(WROW_BYTES == (SPLIT == 3 ? 18 : 12) * 1024 * BMT / 64, "DMA pieces per kernel row");
  static_assert(TR * TW == BP, "planar position tile of BP pixels");
  static_assert(TWS % 4 == 0 && (!UPS || (TR % 2 == 0 && TW % 2 == 0)), "whole quads");
  static_assert(PR * NQ + NHQ <= QPG, "one interior quad or one halo pixel per thread and stage");
  static_assert(SUB >= PR * NQ1 + 4, "dump slots");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// sum over the 16 lanes of a DPP row, every lane ends with the total (butterfly: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror -- after the quad steps the lanes of a quad agree, so the mirrors swap equal-valued quads / halves).  All 64 lanes
// must be active.
__device__ __forceinline__ float emo_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

// the same for N independent values, stage by stage (N butterflies in step: the DPP latency of one hides behind the others)
template <int N>
__device__ __forceinline__ void emo_row16_sum_n(float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0xB1, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x4E, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x141, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x140, 0xf, 0xf, true));
}

// One accumulator element, read where it is used.  The MFMA results live in accumulation registers; left to itself the compiler
// copies all 128 of them to ordinary registers at the K loop's exit, and the epilogue then runs out of registers: residual
// values and addresses are spilled to scratch, and every scratch reload (a vector-memory load as well) drags a vmcnt(0) in
// front of the next store.  The caller is many instructions (a waitcnt, a barrier) behind the last MFMA.
__device__ __forceinline__ float emo_acc_read(float acc_element) {
  float v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc_element));
  return v;
}

// Tile statistics, second half (both epilogues below): the WGP waves' (mean, M2) of a channel combined with the equal-count update
template <int TP, int WGP, int BM>
__device__ __forceinline__ void conv_epilogue_rows_stats(const ConvArgs& a, const float* st_lds, int n, int cotile, int ptile, int tid) {
  __syncthreads();
  if (tid < BM) {
    const int co = cotile * BM + tid;
    if (co < a.Cout) {
      float mean = 0.0f, m2 = 0.0f;
#pragma unroll
      for (int w = 0; w < WGP; ++w) mean += st_lds[(w * BM + tid) * 2 + 0];
      mean *= 1.0f / (float)WGP;
#pragma unroll
      for (int w = 0; w < WGP; ++w) {
        const float d = st_lds[(w * BM + tid) * 2 + 0] - mean;
        m2 += st_lds[(w * BM + tid) * 2 + 1] + (float)(TP * 32) * d * d;
      }
      const long nptiles = (long)a.tiles_x * a.tiles_y * a.tiles_z;
      float2* dst = reinterpret_cast<float2*>(a.gn_stats) + ((long)n * nptiles + ptile) * a.Cout + co;
      *dst = make_float2(mean, m2);
    }
  }
}

// Epilogue of the split kernel.  conv_epilogue (conv_igemm.h) stores straight from the accumulator layout -- a lane owns one
// channel, so one store instruction touches 32 cache lines with 32 bytes each -- and measured 16.2-16.9 k cycles per block
// whatever the layer (tools/conv_phase_timing.py, profiles/r4_conv_phase_timing.jsonl: 15 % of a 128 -> 128 block at 512^2 in the
// bf16 split, 21 % in the fp16 split).  Here every wave transposes its 64 channels x 64 positions through LDS, 32 channels at a
// time, and works in ROW layout: lane (g = lane >> 4, t = lane & 15) holds positions 4t .. 4t + 3 of channel 4 * it + g, so a
// store instruction writes 4 channels x 256 contiguous bytes (the wave's 64 positions are one 64-pixel tile row, or two
// 32-pixel rows), the residual is read the same way, and the GroupNorm statistics of a channel are one 16-lane DPP reduction.
//   scratch   this wave's [32][EPI_ROWF] floats      sbias  [BM] bias, entry i * 32 + g * 8 + it = channel i * 32 + 4 * it + g
//   st_lds    [WGP][BM][2] (mean, M2) exchange
// Same arithmetic per element as conv_epilogue: (leading + small accumulator) [* out_scale], + bias, + residual, activation;
// tile statistics: mean over the wave's 64 values, M2 centred at that mean, waves combined with the equal-count update.
template <int TR, int TW, int TM, int TP, int WGP, int BM, int SPLIT, int ROWF>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& a, floatx16 (&acc_lo)[TM][TP], floatx16 (&acc_hi)[TM][TP],
                                                   float* scratch, const float* sbias, float* st_lds, int n, int cotile, int ptile,
                                                   int ks, int x0, int y0, int z0, int wp, int half, int l32, int lane, int tid,
                                                   unsigned long long* tstamp = nullptr) {
  static_assert(TP == 2 && TM * 32 * 1 <= BM, "wave tile of 64 positions");
  constexpr int NIT = 8;                                  // 4 channels per iteration, 32 per half
  const int g = lane >> 4, t = lane & 15;
  const long plane = (long)a.Hl * a.Wl;
  const long ovol = (long)a.Dl * plane;
  const bool to_partial = a.partial != nullptr;
  const bool has_res = a.res != nullptr && !to_partial;
  const bool want_stats = a.gn_stats != nullptr && !to_partial;
  float* const obase = to_partial ? a.partial + ((long)ks * a.N + n) * a.Cout * ovol : a.out + (long)n * a.Cout * ovol;
  const bool out_al = (reinterpret_cast<unsigned long long>(obase) & 15ull) == 0;
  const int p = wp * (TP * 32) + 4 * t;                  // first of the lane's 4 positions inside the block's tile
  const int y = y0 + p / TW, x = x0 + p % TW;
  const long sp = (long)z0 * plane + (long)y * a.Wl + x;
  const int co0 = cotile * BM + g;                        // + i * 32 + 4 * it

  // residual: every load of the tile is issued before the first use.  The alignment cases are separate straight-line loops --
  // with the test inside the loop the compiler waits vmcnt(0) behind every single load (seen in the ISA)
  floatx4 rv[TM][NIT];
  if (has_res) {
    const long rvol = a.res_ups ? (long)a.Dl * (a.Hl >> 1) * (a.Wl >> 1) : ovol;
    const long rsp = a.res_ups ? ((long)z0 * (a.Hl >> 1) + (y >> 1)) * (a.Wl >> 1) + (x >> 1) : sp;
    const float* rbase = a.res + (long)n * a.Cout * rvol + rsp;
    const unsigned long long ra = reinterpret_cast<unsigned long long>(a.res);
    long roff[TM][NIT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int co = co0 + i * 32 + 4 * it;
        roff[i][it] = (long)(co < a.Cout ? co : a.Cout - 1) * rvol;
      }
    if (a.res_ups) {
      if ((ra & 7ull) == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const float2 r2 = *reinterpret_cast<const float2*>(rbase + roff[i][it]);
            rv[i][it] = floatx4{r2.x, r2.x, r2.y, r2.y};
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const float* rp = rbase + roff[i][it];
            const float r0 = rp[0], r1 = rp[1];
            rv[i][it] = floatx4{r0, r0, r1, r1};
          }
      }
    } else if ((ra & 15ull) == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < NIT; ++it) rv[i][it] = *reinterpret_cast<const floatx4*>(rbase + roff[i][it]);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const float* rp = rbase + roff[i][it];
          rv[i][it] = floatx4{rp[0], rp[1], rp[2], rp[3]};
        }
    }
  }

  EMO_S_STAMP(7)       // (measurement builds: residual loads issued)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i == 1) { EMO_S_STAMP(8) }   // (first 32 channels stored)
    // accumulator layout -> LDS: tile (i, j), register quad q of lane (half, l32) = channel i * 32 + l32, positions
    // j * 32 + 8 * q + 4 * half .. + 3
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        floatx4 c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sum = emo_acc_read(acc_lo[i][j][4 * q + e]) + emo_acc_read(acc_hi[i][j][4 * q + e]);
          c[e] = SPLIT == 3 ? sum : sum * a.out_scale;
        }
        *reinterpret_cast<floatx4*>(scratch + l32 * ROWF + j * 32 + 8 * q + 4 * half) = c;
      }
    // (LDS operations of one wave execute in order: the row reads below see the stores above)
    floatx4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const floatx4*>(scratch + (4 * it + g) * ROWF + 4 * t);
    float bs[NIT];
    if (!to_partial) {
      const floatx4 b0 = *reinterpret_cast<const floatx4*>(sbias + i * 32 + g * 8);
      const floatx4 b1 = *reinterpret_cast<const floatx4*>(sbias + i * 32 + g * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bs[e] = b0[e]; bs[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int co = co0 + i * 32 + 4 * it;
      if (!to_partial) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = v[it][e] + bs[it];
          if (has_res) u += rv[i][it][e];
          v[it][e] = u;
        }
        if (a.act != EMO_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[it][e] = emo_act(v[it][e], a.act);
        }
      }
      if (co < a.Cout) emo_store4(obase + (long)co * ovol + sp, v[it], out_al);
    }
    if (want_stats) {                                     // wave-uniform; all lanes take part in the row reductions
      constexpr float inv_cnt = 1.0f / (float)(TP * 32);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const float s4 = (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]);
        const float mean = emo_row16_sum(s4) * inv_cnt;
        float m2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[it][e] - mean; m2 = __fmaf_rn(d, d, m2); }
        m2 = emo_row16_sum(m2);
        if (t == 0) *reinterpret_cast<float2*>(st_lds + (wp * BM + i * 32 + 4 * it + g) * 2) = make_float2(mean, m2);
      }
    }
  }
  EMO_S_STAMP(9)
  if (want_stats) conv_epilogue_rows_stats<TP, WGP, BM>(a, st_lds, n, cotile, ptile, tid);
}

// The same epilogue for the launch form the decoders use almost everywhere -- final output (no K split), no activation, a full
// 64-channel tile, 16-byte aligned tensors, offsets inside 2^31 bytes per sample -- with every one of those a compile-time fact.
// conv_epilogue_rows tests them per element: its per-value activation switch alone is four taken branches per stored value, and
// the residual registers of its three alignment cases meet in phi nodes, so the compiler waits for every load right behind its
// issue (ISA: `s_waitcnt vmcnt(15) ... vmcnt(2)` + moves): 20-21 k cycles per tile whatever the layer, 30 % of a 128 -> 128 tile
// of the fp16 split (profiles/r4_conv_phase_timing_final.jsonl).  Here the path is straight-line code:
//   conv_epilogue_fast_issue   the residual loads of 32 channels (8 per lane): the first 32 at the top, the second 32 behind the
//                              first half's accumulator -> LDS stores;
//   conv_epilogue_fast_finish  LDS transposition, + bias (+ residual), stores (one scalar base + a 32-bit lane offset), tile
//                              statistics.
// RES: 0 no residual, 1 residual of the output's size (16-byte aligned), 2 half-size residual, nearest x2 (8-byte aligned).
// Arithmetic per element identical to conv_epilogue_rows (same operations in the same order): the two are bit-identical.
// LAUNDER (conv_igemm_f16x2_ct2.h): the volume sizes pass through an empty asm statement, so that the 32 per-lane offsets
// (channel * volume + position) they enter are computed where they are used.  Left visible, the compiler hoists them out of the
// item loop as loop invariants; a kernel whose accumulators fill all 256 accumulation registers has nowhere to keep them but
// scratch, and reloaded each one in the epilogue behind a vmcnt(0) of its own (seen in the ISA: 44 serialised scratch loads per item).
template <int TW, int TP, int BM, int RES, int I, bool LAUNDER = false>
__device__ __forceinline__ void conv_epilogue_fast_issue(const ConvArgs& a, floatx4 (&rv)[8], int n, int cotile, int x0, int y0,
                                                         int z0, int wp, int lane) {
  if constexpr (RES != 0) {
    const int g = lane >> 4, t = lane & 15;
    const int p = wp * (TP * 32) + 4 * t;
    const int y = y0 + p / TW, x = x0 + p % TW;
    const unsigned Hr = RES == 2 ? a.Hl >> 1 : a.Hl, Wr = RES == 2 ? a.Wl >> 1 : a.Wl;
    unsigned rvol = (unsigned)a.Dl * Hr * Wr;
    if constexpr (LAUNDER) asm volatile("" : "+s"(rvol));
    const unsigned rsp = RES == 2 ? ((unsigned)z0 * Hr + (y >> 1)) * Wr + (x >> 1) : ((unsigned)z0 * Hr + y) * Wr + x;
    const float* rbase = a.res + ((long)n * a.Cout + (long)cotile * BM) * rvol;      // wave-uniform
    const unsigned roff = (unsigned)g * rvol + rsp;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const float* rp = rbase + (roff + (unsigned)(I * 32 + 4 * it) * rvol);
      if constexpr (RES == 1) rv[it] = *reinterpret_cast<const floatx4*>(rp);
      else { const float2 r2 = *reinterpret_cast<const float2*>(rp); rv[it] = floatx4{r2.x, r2.y, 0.0f, 0.0f}; }
    }
  }
}

// DEFER_STATS (conv_igemm_f16x2_ct2.h): the waves' (mean, M2) entries are left in st_lds; the caller combines them behind a
// barrier of its own (the one that ends its item anyway) instead of a second one per channel tile.
template <int TW, int TM, int TP, int WGP, int BM, int SPLIT, int ROWF, int RES, bool LAUNDER = false, bool DEFER_STATS = false>
__device__ __forceinline__ void conv_epilogue_fast_finish(const ConvArgs& a, floatx16 (&acc_lo)[TM][TP], floatx16 (&acc_hi)[TM][TP],
                                                          floatx4 (&rv0)[8], float* scratch, const float* sbias,
                                                          float* st_lds, int n, int cotile, int ptile, int x0, int y0, int z0, int wp,
                                                          int half, int l32, int lane, int tid, unsigned long long* tstamp = nullptr) {
  static_assert(TP == 2 && (TM == 1 || TM == 2) && BM == 32 * TM, "wave tile of 64 (32) channels x 64 positions");
  constexpr int NIT = 8;
  const int g = lane >> 4, t = lane & 15;
  const bool want_stats = a.gn_stats != nullptr;
  const unsigned plane = (unsigned)a.Hl * a.Wl;
  unsigned ovol = (unsigned)a.Dl * plane;
  if constexpr (LAUNDER) asm volatile("" : "+s"(ovol));
  const int p = wp * (TP * 32) + 4 * t;
  const int y = y0 + p / TW, x = x0 + p % TW;
  float* const obase = a.out + ((long)n * a.Cout + (long)cotile * BM) * ovol;          // wave-uniform
  const unsigned off = (unsigned)g * ovol + (unsigned)z0 * plane + (unsigned)y * a.Wl + x;
  floatx4 rv1[8];
  EMO_S_STAMP(7)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i == 1) { EMO_S_STAMP(8) }
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (two values per instruction -- v_pk_add_f32 / v_pk_mul_f32: behind the K loop there is no MFMA stream to disturb, and
        // the epilogue is bound by its instruction count; per element the operations and their order are those of
        // conv_epilogue_rows)
        floatx2 c01 = floatx2{emo_acc_read(acc_lo[i][j][4 * q + 0]), emo_acc_read(acc_lo[i][j][4 * q + 1])} +
                      floatx2{emo_acc_read(acc_hi[i][j][4 * q + 0]), emo_acc_read(acc_hi[i][j][4 * q + 1])};
        floatx2 c23 = floatx2{emo_acc_read(acc_lo[i][j][4 * q + 2]), emo_acc_read(acc_lo[i][j][4 * q + 3])} +
                      floatx2{emo_acc_read(acc_hi[i][j][4 * q + 2]), emo_acc_read(acc_hi[i][j][4 * q + 3])};
        if constexpr (SPLIT != 3) {
          const floatx2 sc2 = floatx2{a.out_scale, a.out_scale};
          c01 = c01 * sc2;
          c23 = c23 * sc2;
        }
        *reinterpret_cast<floatx4*>(scratch + l32 * ROWF + j * 32 + 8 * q + 4 * half) = floatx4{c01[0], c01[1], c23[0], c23[1]};
      }
    // the second half's residual: issued now, into the registers the first half's accumulators have just left (with all 16
    // loads in flight from the start the compiler spills the landed values to scratch and waits vmcnt(0) before every store)
    if (i == 0 && TM == 2) conv_epilogue_fast_issue<TW, TP, BM, RES, 1, LAUNDER>(a, rv1, n, cotile, x0, y0, z0, wp, lane);
    floatx4 (&rv)[8] = i == 0 ? rv0 : rv1;
    floatx4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) v[it] = *reinterpret_cast<const floatx4*>(scratch + (4 * it + g) * ROWF + 4 * t);
    const floatx4 b0 = *reinterpret_cast<const floatx4*>(sbias + i * 32 + g * 8);
    const floatx4 b1 = *reinterpret_cast<const floatx4*>(sbias + i * 32 + g * 8 + 4);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const float bs = it < 4 ? b0[it & 3] : b1[it & 3];
      const floatx2 bs2 = floatx2{bs, bs};
      floatx2 u01 = floatx2{v[it][0], v[it][1]} + bs2, u23 = floatx2{v[it][2], v[it][3]} + bs2;
      if constexpr (RES == 1) { u01 = u01 + floatx2{rv[it][0], rv[it][1]}; u23 = u23 + floatx2{rv[it][2], rv[it][3]}; }
      if constexpr (RES == 2) { u01 = u01 + floatx2{rv[it][0], rv[it][0]}; u23 = u23 + floatx2{rv[it][1], rv[it][1]}; }
      v[it] = floatx4{u01[0], u01[1], u23[0], u23[1]};
      float* const op = obase + (off + (unsigned)(i * 32 + 4 * it) * ovol);
      if (EMO_CONV_NT_STORE) __builtin_nontemporal_store(v[it], reinterpret_cast<floatx4*>(op));
      else *reinterpret_cast<floatx4*>(op) = v[it];
    }
    if (want_stats) {
      // The eight channels of the lane are reduced IN STEP -- stage by stage of the butterfly, all eight before the next stage --
      // and stored behind one test.  Written channel by channel (reduce, store under `if (t == 0)`, next channel) the exec-mask
      // region of every store fenced the scheduler: sixteen fully serialised chains of 8 dependent DPP adds per tile, each add
      // behind its two wait states -- 2.8 k of the epilogue's 9.2 k cycles (profiles/r5_conv_phase_epilogue_parts.jsonl).  Same
      // operations in the same order per channel: the statistics are bit-identical to conv_epilogue_rows'
      constexpr float inv_cnt = 1.0f / (float)(TP * 32);
      float mean[NIT], m2[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const floatx2 p2 = floatx2{v[it][0], v[it][2]} + floatx2{v[it][1], v[it][3]};      // (v0 + v1, v2 + v3)
        mean[it] = p2[0] + p2[1];
      }
      emo_row16_sum_n<NIT>(mean);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        mean[it] *= inv_cnt;
        const floatx2 mean2 = floatx2{mean[it], mean[it]};
        const floatx2 d01 = floatx2{v[it][0], v[it][1]} - mean2, d23 = floatx2{v[it][2], v[it][3]} - mean2;
        float q = 0.0f;
        q = __fmaf_rn(d01[0], d01[0], q);
        q = __fmaf_rn(d01[1], d01[1], q);
        q = __fmaf_rn(d23[0], d23[0], q);
        q = __fmaf_rn(d23[1], d23[1], q);
        m2[it] = q;
      }
      emo_row16_sum_n<NIT>(m2);
      if (t == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          *reinterpret_cast<float2*>(st_lds + (wp * BM + i * 32 + 4 * it + g) * 2) = make_float2(mean[it], m2[it]);
      }
    }
  }
  EMO_S_STAMP(9)
  if constexpr (!DEFER_STATS) {
    if (want_stats) conv_epilogue_rows_stats<TP, WGP, BM>(a, st_lds, n, cotile, ptile, tid);
  }
}

template <int TR, int TW, bool UPS, int SPLIT, int BMT = 64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_igemm_bf16x3_kernel(const ConvArgs a) {
  using Cfg = ConvCfgS<TR, TW, UPS, SPLIT, BMT>;
  using opx8 = typename std::conditional<SPLIT == 3, bf16x8, halfx8>::type;      // one LDS slot: 8 channels of one plane
  constexpr int NPL = Cfg::NPL;
  constexpr int BM = Cfg::BM, TM = Cfg::TM, TP = Cfg::TP, WGP = Cfg::WGP, KC = Cfg::KC;
  constexpr int PR = Cfg::PR, NQ = Cfg::NQ, NQ1 = Cfg::NQ1, SUB = Cfg::SUB, CHS = Cfg::CHS, QPG = Cfg::QPG;
  constexpr int NHQ = Cfg::NHQ, TWS = Cfg::TWS, WPLANE = Cfg::WPLANE, WROW = Cfg::WROW, PPL = Cfg::PPL, PBUF = Cfg::PBUF;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  opx8* const lds8 = reinterpret_cast<opx8*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int wp = wave;
  const int m0 = 0, p0 = wp * TP * 32;

  if (a.run_if != nullptr && *a.run_if == 0) return;   // guarded fallback launch: the fp16-split launch of the layer stayed in range
#if EMO_S_STAGGER
  // measurement builds: the persistent blocks of an XCD start in four groups, EMO_S_STAGGER cycles apart -- identical items on
  // every CU otherwise keep the whole chip in lock-step (every CU in its memory-heavy prologue / epilogue at the same time)
  for (int k = 0; k < (int)((blockIdx.x >> 3) & 3) * (EMO_S_STAGGER / 4096); ++k) __builtin_amdgcn_s_sleep(64);
#endif
  float sat_m = 0.0f;                                    // SPLIT == 2: largest |scaled staged value| this thread has seen


  // ---- constants of the launch and of the thread ----
  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const bool has_affine = a.scale != nullptr;
  // epilogue form (wave-uniform, see conv_epilogue_fast_finish): 0 / 1 / 2 = the straight-line form without / with a same-size /
  // with a half-size residual, -1 = the general one
  const int epi_mode = __builtin_amdgcn_readfirstlane(
      (a.partial != nullptr || a.act != EMO_ACT_NONE || a.Cout % BM != 0 || (a.Wl & 3) != 0 ||
       (long)a.Dl * a.Hl * a.Wl > (1l << 23) || (reinterpret_cast<unsigned long long>(a.out) & 15ull) != 0) ? -1
      : a.res == nullptr ? 0
      : !a.res_ups ? ((reinterpret_cast<unsigned long long>(a.res) & 15ull) == 0 ? 1 : -1)
      : ((reinterpret_cast<unsigned long long>(a.res) & 7ull) == 0 ? 2 : -1));
  const float in_scale = SPLIT == 3 ? 1.0f : a.in_scale;
  const int padD = a.KD >> 1;
  // bounds of the staged value: ReLU or none; the fp16 split saturates at the fp16 range (of the scaled value)
  // (bf16 split: the largest finite bf16, 0x7f7f0000 -- the first term of a larger value would round to infinity and the residual
  // inf - inf to NaN; +-inf inputs therefore saturate at +-3.39e38, include/emo_hip.h)
  constexpr float CLAMP_HI = SPLIT == 3 ? 3.3895313892515355e38f : 65504.0f;
  const float clamp_lo = a.relu_in ? 0.0f : -CLAMP_HI;
  const int nstages_all = a.n_cchunks * a.KD;
  const int nptiles = a.tiles_x * a.tiles_y * a.tiles_z;

  // ---- staging map: thread t belongs to channel group t / QPG; within the group, lane u < PR * NQ owns interior quad u (4
  //      consecutive pixels of one patch row), lane PR * NQ + 2 * row + side owns one halo pixel.  A halo lane issues the SAME
  //      16-byte loads as everybody else -- the aligned quad that contains its pixel (left: x0 - 4 .. x0 - 1, element 3;
  //      right: x0 + TWS .. + 3, element 0) -- and runs the same conversion; the three pixels it does not own go to dump slots
  //      at the tail of their sub-row.  No wave does extra work: with a dedicated halo wave (8 dword loads + an 8-value
  //      conversion per stage on wave 3 only) the other three waves sat 700-850 cycles per stage in the K loop's barriers
  //      (tools/conv_phase_timing.py on an EMO_S_TIMING=2 build, profiles/r4_conv_phase_waves.jsonl) ----
  const int q_u = tid % QPG;
  const int q_g = __builtin_amdgcn_readfirstlane(tid / QPG);
  const bool is_quad = q_u < PR * NQ;
  const int hq = q_u - PR * NQ;
  const bool is_halo = !is_quad && hq < NHQ;
  const int h_side = hq & 1;
  const int q_r = is_quad ? q_u / NQ : (is_halo ? hq >> 1 : 0);
  const int q_c = is_quad ? q_u - q_r * NQ : 0;
  int q_slb[4];                                           // byte offsets of the lane's four staging slots inside a patch buffer
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int dump = q_g * CHS + i * SUB + PR * NQ1 + (q_u & 3);
    const int own = is_quad ? q_g * CHS + i * SUB + q_r * NQ1 + q_c : q_g * CHS + h_side * SUB + q_r * NQ1 + NQ;
    q_slb[i] = ((is_quad || (is_halo && i == (h_side ? 0 : 3))) ? own : dump) * 16;
  }

  constexpr int TPH = TP;
  floatx16 acc_lo[TM][TPH], acc_hi[TM][TPH];

  // ---- work items.  Item -> (sample, position tile, channel tile, K split): XCD-contiguous order, channel tile fastest
  //      (conv_igemm.h).  A block walks the items of its XCD's contiguous range with a stride: min(n_work, CUs) persistent
  //      blocks by default (no workgroup launch between two items: 3.5-4 k cycles of a 50-110 k cycle item), one block per item
  //      with EMO_CONV_BF16X3_PERSISTENT=0 (conv_igemm_bf16x3_launch).
  //      Consecutive items of a block are chained through one stage pipeline in the fp16 split ("the item loop" below); a
  //      first build of that, earlier this round, lost in the K loop what it saved in the prologue (260 accumulation-register
  //      spill moves per stage pair against 47: tools/session/r4_call13.sh) and computed wrong tiles in the bf16 split -- the
  //      former went away with the accumulators read where they are used and a branch-free loop, the latter was, in all
  //      likelihood, the scalar-register hazard described at EMO_SGPR_HAZARD_NOP (conv_igemm.h) ----
  const int q8 = a.n_work >> 3, r8 = a.n_work & 7;
  const int xcd = blockIdx.x & 7;
  const int n_mine = q8 + (xcd < r8 ? 1 : 0);
  const int l_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int l_stride = (gridDim.x + 7) >> 3;
  // the item being computed (it_*) and the tile of its patch loads (lq_*)
  int it_L = 0, it_cotile = 0, it_ks = 0, it_n = 0, it_ptile = 0, it_x0 = 0, it_y0 = 0, it_z0 = 0, it_st_begin = 0, it_st_end = 0;
  unsigned lq_off = 0;
  bool lq_ok = false;
  int lq_z0 = 0;
// (every result through readfirstlane: the uniform divisions run on the vector ALU, and once these variables are carried from
// one iteration of the item loop to the next the compiler would otherwise keep the whole web in vector registers -- which the
// "s" operands of the pinned loads cannot take)
#define EMO_S_DECODE(P_, L_)                                                                          \
  {                                                                                                   \
    P_##L = (L_);                                                                                     \
    int cot_ = P_##L % a.n_cotiles;                                                                   \
    int rest_ = P_##L / a.n_cotiles;                                                                  \
    int ks_ = 0;                                                                                      \
    if (a.ksplit > 1) {                                                                               \
      ks_ = rest_ % a.ksplit;                                                                         \
      rest_ /= a.ksplit;                                                                              \
    }                                                                                                 \
    const int n_ = rest_ / nptiles;                                                                   \
    int bx_ = rest_ - n_ * nptiles;                                                                   \
    P_##ptile = __builtin_amdgcn_readfirstlane(bx_);                                                  \
    const int tx_ = bx_ % a.tiles_x; bx_ /= a.tiles_x;                                                \
    const int ty_ = bx_ % a.tiles_y; bx_ /= a.tiles_y;                                                \
    P_##cotile = __builtin_amdgcn_readfirstlane(cot_ + a.cot0);                                       \
    P_##ks = __builtin_amdgcn_readfirstlane(ks_);                                                     \
    P_##n = __builtin_amdgcn_readfirstlane(n_);                                                       \
    P_##x0 = __builtin_amdgcn_readfirstlane(tx_ * TW);                                                \
    P_##y0 = __builtin_amdgcn_readfirstlane(ty_ * TR);                                                \
    P_##z0 = __builtin_amdgcn_readfirstlane(bx_);                                                     \
    P_##st_begin = __builtin_amdgcn_readfirstlane(P_##ks * a.stages_per_split);                       \
    P_##st_end = __builtin_amdgcn_readfirstlane(min(nstages_all, P_##st_begin + a.stages_per_split)); \
  }
// the packed kernel rows of an item's channel tile
#define EMO_S_WSRC(P_) (reinterpret_cast<const char*>(a.wpk) + ((long)P_##cotile * nstages_all) * (3 * Cfg::WROW_BYTES))
#define it_wsrc EMO_S_WSRC(it_)
// points the patch-load cursor at the tile of item P_: the lane's 16-byte load (its quad, or the aligned quad that contains its
// halo pixel) and whether it lies inside the image.  Per-lane values are derived where the cursor moves, not carried per item
#define EMO_S_CURSOR_OF(P_, ok_, off_)                                                                \
  {                                                                                                   \
    const int x0s_ = UPS ? P_##x0 >> 1 : P_##x0, y0s_ = UPS ? P_##y0 >> 1 : P_##y0;   /* tile origin in source pixels */ \
    const int q_y_ = y0s_ - 1 + q_r;                                                                  \
    const int q_x_ = is_quad ? x0s_ + 4 * q_c : (h_side ? x0s_ + TWS : x0s_ - 4);   /* first pixel of the lane's 16-byte load */ \
    ok_ = (is_quad || is_halo) && (unsigned)q_y_ < (unsigned)a.H && q_x_ >= 0 && q_x_ < a.W;          \
    off_ = ok_ ? (unsigned)(q_y_ * a.W + q_x_) * 4u : 0u;                                             \
  }
#define EMO_S_CURSOR_TO(P_) { EMO_S_CURSOR_OF(P_, lq_ok, lq_off) lq_z0 = P_##z0; }

  // LDS byte offsets of the lane's operands: weight fragment (+ row buffer / plane / tap column immediates) and, for every tap,
  // the patch slot of the lane's output pixel (+ half * CHS: the lane's 8-channel group; + buffer / plane immediates)
  // Kernel row r adds r patch rows = r * NQ1 slots (an immediate); with the fused upsample the source row of kernel row r is
  // (row + r + 1) >> 1: kernel row 2 is kernel row 0 plus one patch row, kernel row 1 depends on the parity of the pixel row
  // and keeps its own registers.
  const int a_off = (half * BM + m0 + l32) * 16;
  constexpr int NBR = UPS ? 2 : 1;
  int b_off[TP][NBR][3];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW, row = p / TW;
#pragma unroll
    for (int r = 0; r < NBR; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int pr = UPS ? ((row + r - 1 + 2) >> 1) - 1 + 1 : row + r;
        const int pc = UPS ? ((col + s - 1 + 2) >> 1) - 1 : col + s - 1;
        const int slot = pc < 0 ? pr * NQ1 + NQ : (pc >= TWS ? SUB + pr * NQ1 + NQ : (pc & 3) * SUB + pr * NQ1 + (pc >> 2));
        b_off[j][r][s] = (half * CHS + slot) * 16;
      }
  }
#define EMO_S_B_OFF(j_, r_, s_) (UPS ? ((r_) == 2 ? b_off[j_][0][s_] + NQ1 * 16 : b_off[j_][(r_) < NBR ? (r_) : 0][s_]) \
                                     : b_off[j_][0][s_] + (r_) * NQ1 * 16)

  const char* const lds_c = reinterpret_cast<const char*>(smem);
  char* const lds_w = reinterpret_cast<char*>(smem);
  opx8 fa_[2][NPL][TM], fb_[2][NPL][TP];     // [register set: this step / the next][plane][tile]
#define EMO_S_LOAD_FRAGS_PLANE(set_, pl_, wbase_, pbase_, r_, s_)                                      \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
      fa_[set_][pl_][i] = *reinterpret_cast<const opx8*>(lds_c + a_off + ((wbase_) + (pl_) * WPLANE + (s_) * 2 * BM + i * 32) * 16); \
    _Pragma("unroll") for (int j = 0; j < TP; ++j)                                                    \
      fb_[set_][pl_][j] = *reinterpret_cast<const opx8*>(lds_c + EMO_S_B_OFF(j, r_, s_) + ((pbase_) + (pl_) * PPL) * 16); \
  }
#define EMO_S_LOAD_FRAGS(set_, wbase_, pbase_, r_, s_)                                                \
  { _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) EMO_S_LOAD_FRAGS_PLANE(set_, pl, wbase_, pbase_, r_, s_) }

  float* const sct = smem + Cfg::OFF_SCT * 4;
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);
  const unsigned lane16 = (unsigned)lane * 16u;

  // raw patch registers, double-buffered by stage parity: the loads of stage k + 2 are issued (into buffer k & 1) while the
  // conversion of stage k + 1 still reads buffer (k + 1) & 1 -- that lets the conversion spread over eight steps of a stage,
  // half a pixel (4 channels) per step, instead of 2 + 1 + 1 pixels in three steps (where 120 VALU beside 24 resp. 12 MFMAs
  // made those steps issue-bound: 1690 / 880 / 820 cycles against 768, profiles/r4_conv_phase_steps.jsonl)
  floatx4 qv[2][8];
  float q_lo[2], q_hi[2];
  int q_tix[2];
  floatx4 q_sc, q_sh;                       // scale / shift of the four channels being converted (read at the top of the step)
  opx8 cv_h, cv_m, cv_l;                    // the pixel under conversion: 8 channels per plane, filled in two halves
  emo_intx4 xrs = emo_raw_buffer(a.x);      // (re-based on the item's sample by the prologue)
  unsigned usoff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) usoff[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)u * (unsigned)DHW * 4u));

  int n_ci0, n_zu;
  bool n_zv;
  int ld_stage, ld_cc, ld_kd;   // the stage whose patch is being loaded: channel chunk and depth tap, stepped (no division in the loop)
#define EMO_S_SET_STAGE_VARS()                                                                        \
  {                                                                                                   \
    n_ci0 = ld_cc * KC;                                                                               \
    n_zu = lq_z0 + ld_kd - padD;                                                                      \
    n_zv = (unsigned)n_zu < (unsigned)a.D;                                                            \
  }
#define EMO_S_SET_STAGE_INIT(stage_)                                                                  \
  {                                                                                                   \
    ld_stage = (stage_);                                                                              \
    ld_cc = ld_stage / a.KD;                                                                          \
    ld_kd = ld_stage - ld_cc * a.KD;                                                                  \
    EMO_S_SET_STAGE_VARS()                                                                            \
  }
// target is ld_stage or ld_stage + 1 (the clamped stage sequence st + 1, st + 2, ...)
#define EMO_S_SET_STAGE_STEP(target_)                                                                 \
  {                                                                                                   \
    if ((target_) != ld_stage) {                                                                      \
      ++ld_stage;                                                                                     \
      if (++ld_kd == a.KD) { ld_kd = 0; ++ld_cc; }                                                    \
    }                                                                                                 \
    EMO_S_SET_STAGE_VARS()                                                                            \
  }
  unsigned q_vo;
#define EMO_S_ISSUE_BEGIN(b_)                                                                         \
  {                                                                                                   \
    const int c0_ = n_ci0 + q_g * 8;                                                                  \
    const bool cv_ = c0_ < a.Cin;                                                                     \
    const int cs_ = cv_ ? c0_ : 0;                                                                    \
    const bool keep_ = lq_ok && cv_ && n_zv;                                                          \
    q_lo[b_] = keep_ ? clamp_lo : 0.0f;                                                               \
    q_hi[b_] = keep_ ? CLAMP_HI : 0.0f;                                                               \
    q_vo = lq_off + ((unsigned)cs_ * (unsigned)DHW + (unsigned)((n_zv ? n_zu : 0) * HW)) * 4u;        \
    q_tix[b_] = (has_affine ? cs_ : (cs_ & (Cfg::SCT - 1))) >> 2;                                     \
  }
#define EMO_S_ISSUE_LOADS(b_, u0_, u1_)                                                               \
  { _Pragma("unroll") for (int u = (u0_); u < (u1_); u += 2) emo_bload4x2_pinned(xrs, q_vo, usoff[u], usoff[u + 1], qv[b_][u], qv[b_][u + 1]); }
#define EMO_S_HALF_TABLE(b_, hf_)                                                                     \
  {                                                                                                   \
    const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + q_tix[b_] + (hf_);                   \
    q_sc = t4_[0]; q_sh = t4_[Cfg::SCT / 4];                                                          \
  }
#define EMO_S_TOUCH_QUAD(b_) { _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch4(qv[b_][u]); }
// Conversion of channels 4 * hf_ .. + 3 of pixel i_ of buffer b_: fp32 transform (GroupNorm affine of the producer; ReLU,
// saturation and zero padding in one v_med3: bounds [0, 0] where the pixel is padding), then the exact split into the operand
// planes -- v = h + m + l, round-to-nearest-even at every level, the residuals are exact (SPLIT = 2: h + m of the scaled value).
// The second half stores the pixel's slot of every plane (every lane stores all four pixels of its loads: its own, or into
// dump slots).
#define EMO_S_CONV_HALF(b_, pbase_, i_, hf_)                                                          \
  {                                                                                                   \
    float t_[4];                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                     \
      t_[k] = __fmaf_rn(qv[b_][4 * (hf_) + k][i_], q_sc[k], q_sh[k]);                                 \
    if constexpr (SPLIT == 2) {   /* range check of the fp16 split: max |pre-clamp value| (v_max3 with |.| modifiers) */ \
      sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[0])), __builtin_fabsf(t_[1])); \
      sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[2])), __builtin_fabsf(t_[3])); \
    }                                                                                                 \
    if constexpr (SPLIT == 3) {                                                                       \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                 \
        const int u = 4 * (hf_) + k;                                                                  \
        const float v = __builtin_amdgcn_fmed3f(t_[k], q_lo[b_], q_hi[b_]);                           \
        cv_h[u] = (__bf16)v;                                                                          \
        const float r1 = v - (float)cv_h[u];                                                          \
        cv_m[u] = (__bf16)r1;                                                                         \
        const float r2 = r1 - (float)cv_m[u];                                                         \
        cv_l[u] = (__bf16)r2;                                                                         \
      }                                                                                               \
    } else {                                                                                          \
      _Pragma("unroll") for (int k = 0; k < 4; k += 2)                                                \
        emo_split_f16x2_pair(__builtin_amdgcn_fmed3f(t_[k], q_lo[b_], q_hi[b_]),                      \
                             __builtin_amdgcn_fmed3f(t_[k + 1], q_lo[b_], q_hi[b_]), cv_h, cv_m, 4 * (hf_) + k); \
    }                                                                                                 \
    if ((hf_) == 1) {                                                                                 \
      char* d_ = lds_w + q_slb[i_] + (pbase_) * 16;                                                   \
      *reinterpret_cast<opx8*>(d_) = cv_h;                                                            \
      *reinterpret_cast<opx8*>(d_ + PPL * 16) = cv_m;                                                 \
      if constexpr (SPLIT == 3) *reinterpret_cast<opx8*>(d_ + 2 * PPL * 16) = cv_l;                   \
    }                                                                                                 \
  }
// one kernel row of one stage by LDS-DMA (1 KiB per wave-instruction), lane-linear = the packed order
#define EMO_S_DMA_PIECE_TO(ptr_, wb_, row_, i_)                                                       \
  {                                                                                                   \
    const int j = (SPLIT == 2 || (i_) < 4) ? wave + 4 * (i_) : 16 + (wave & 1);                       \
    emo_dma16_pinned_s((ptr_) + ((row_) * Cfg::WROW_BYTES + j * 1024), lane16,                        \
                       smem_lds + (unsigned)((Cfg::OFF_W + (wb_) * Cfg::WSTAGE) * 16 + (row_) * Cfg::WROW_BYTES + j * 1024)); \
  }
#define EMO_S_DMA_PIECE(ptr_, row_, i_) EMO_S_DMA_PIECE_TO(ptr_, 0, row_, i_)
#define EMO_S_DMA_ROW(ptr_, row_)                                                                     \
  { _Pragma("unroll") for (int i = 0; i < Cfg::NDMA; ++i) EMO_S_DMA_PIECE(ptr_, row_, i) }
// piece k = 0 .. NSTP - 1 of a whole stage (one-barrier schedule; the three kernel rows of a stage are contiguous on both sides):
// piece wave + 4 k of the stage -- for the 64-row tile that is row k / 3, piece wave + 4 (k % 3) of the row; the 18-piece stage of
// the 32-row tile ends with pieces 16, 17, copied twice (waves 0 / 2 and 1 / 3: the same bytes to the same place)
#define EMO_S_DMA_STAGE_PIECE(ptr_, wb_, k_)                                                          \
  {                                                                                                   \
    const int j = (BMT == 64 || (k_) < 4) ? wave + 4 * (k_) : 16 + (wave & 1);                        \
    emo_dma16_pinned_s((ptr_) + j * 1024, lane16, smem_lds + (unsigned)((Cfg::OFF_W + (wb_) * Cfg::WSTAGE) * 16 + j * 1024)); \
  }
#define EMO_S_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#if EMO_S_TIMING == 2
// measurement build: how long every wave sits in the waitcnt of a K-loop barrier (memory / LDS latency it did not hide) and in
// the s_barrier itself (skew between the block's waves), summed per work item and wave
#define EMO_S_BARRIER(n_)                                                                             \
  {                                                                                                   \
    const unsigned long long b0_ = __builtin_amdgcn_s_memtime();                                      \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n_) : "memory");                              \
    const unsigned long long b1_ = __builtin_amdgcn_s_memtime();                                      \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                  \
    const unsigned long long b2_ = __builtin_amdgcn_s_memtime();                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    tw_wait += b1_ - b0_; tw_bar += b2_ - b1_; ++tw_n;                                                \
  }
#else
#define EMO_S_BARRIER(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")
#endif
#define EMO_S_LOOP_BARRIER(n_) { if (EMO_S_ABLATE & 2) { EMO_S_WAIT(n_); } else { EMO_S_BARRIER(n_); } }


  // the partial products, smallest first: (weight plane, patch plane); the last one is the leading product
  constexpr bool ONEBAR = Cfg::NWB == 2;      // two whole weight stages in LDS: one barrier per stage (ConvCfgS)
  constexpr int NPROD = SPLIT == 3 ? EMO_S_PRODUCTS : 3;
  constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};
  constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};

  // ---- prologue, first part: everything that is only ISSUED -- the scale / shift / bias table entries (through registers: their
  //      global loads go out BEFORE the pinned loads and are stored to LDS behind them, one memory round trip for everything;
  //      while pinned loads are in flight the compiler must never copy or spill a register whose load has not landed,
  //      tools/kernel_resources.py --audit), the kernel rows of the first stage by DMA, the patch loads of the first two stages.
  //      A persistent block runs this part for its NEXT item behind the closing barrier of the current item's K loop, in front
  //      of the epilogue: the raw patch registers and the weight buffers are idle there, and the round trip to memory (most of the
  //      7-10 k cycles a prologue took, profiles/r4_conv_phase_timing_final.jsonl) hides behind the epilogue's own traffic ----
  constexpr int NTE = Cfg::SCT / 256;
  float te_sc[NTE], te_sh[NTE], te_b = 0.0f;
#define EMO_S_PROLOGUE_ISSUE(P_, LOADS_)                                                              \
  {                                                                                                   \
    xrs = emo_raw_buffer(a.x + (long)P_##n * a.Cin * DHW);                                            \
    EMO_S_CURSOR_TO(P_)                                                                               \
    if (LOADS_) {                                                                                     \
      _Pragma("unroll") for (int k = 0; k < NTE; ++k) {                                               \
        const int c = tid + 256 * k;                                                                  \
        const bool real = has_affine && c < a.Cin;                                                    \
        te_sc[k] = real ? a.scale[(long)P_##n * a.Cin + c] : 1.0f;                                    \
        te_sh[k] = real ? a.shift[(long)P_##n * a.Cin + c] : 0.0f;                                    \
      }                                                                                               \
      if (tid < BM && a.bias != nullptr && a.partial == nullptr) {                                    \
        const int co_ = P_##cotile * BM + tid;                                                        \
        te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];                                               \
      }                                                                                               \
      if constexpr (ONEBAR) {                                                                         \
        _Pragma("unroll") for (int k = 0; k < Cfg::NSTP; ++k)                                         \
          EMO_S_DMA_STAGE_PIECE(EMO_S_WSRC(P_) + (long)P_##st_begin * (3 * Cfg::WROW_BYTES), 0, k)    \
      } else {                                                                                        \
        EMO_S_DMA_ROW(EMO_S_WSRC(P_) + (long)P_##st_begin * (3 * Cfg::WROW_BYTES), 0);                \
        EMO_S_DMA_ROW(EMO_S_WSRC(P_) + (long)P_##st_begin * (3 * Cfg::WROW_BYTES), 1);                \
      }                                                                                               \
    }                                                                                                 \
    EMO_S_SET_STAGE_INIT(P_##st_begin);                                                               \
    EMO_S_ISSUE_BEGIN(0)                                                                              \
    if (LOADS_) EMO_S_ISSUE_LOADS(0, 0, 8)                                                            \
    {                                                                                                 \
      const int st1 = (P_##st_begin + 1) < P_##st_end ? (P_##st_begin + 1) : P_##st_begin;            \
      EMO_S_SET_STAGE_STEP(st1);                                                                      \
    }                                                                                                 \
    EMO_S_ISSUE_BEGIN(1)                                                                              \
    if (LOADS_) EMO_S_ISSUE_LOADS(1, 0, 8)                                                            \
  }

  // ---- the item loop.  CHAINED items (fp16 split): when the block's next item belongs to the same sample (same scale / shift
  //      tables) and both have an even number >= 2 of stages (buffer parities line up), the stage pipeline simply runs on into
  //      it -- the loads, kernel rows and conversion that the last two stages of an item issue for "stage + 1 / + 2" were dead
  //      re-stages of the last stage; they now fetch the next item's first two stages, at no extra instruction in the K loop.
  //      Behind the epilogue the next item then starts with a four-line prologue (accumulators, bias, two weight pieces, one
  //      barrier) instead of the full one (7-10 k cycles of a 57 k cycle item, bound by what one CU can pull from L2 in a burst:
  //      issuing the same loads in front of the epilogue or de-phasing the CUs moved that time, it did not remove it --
  //      tools/session/r4_call24.sh, r4_call25.sh).
  //      No uniform state is carried around the loop except the flag: the item and its successor are decoded afresh in every
  //      iteration (a web of loop-carried uniform values ends up in vector registers, which the "s" operands of the pinned
  //      loads cannot take) ----
  constexpr bool CHAIN = Cfg::CHAIN;
  bool chained_in = false;                 // this item's first two stages were staged by the previous item's last two
  for (int idx8 = blockIdx.x >> 3; idx8 < n_mine; idx8 += l_stride) {
#if EMO_S_TIMING
  unsigned long long tstamp[12];
  for (int k = 0; k < 12; ++k) tstamp[k] = 0;
  unsigned long long tw_wait = 0, tw_bar = 0, tw_n = 0;
  unsigned long long tstep[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // EMO_S_TIMING == 3: cycles per step index of the K loop, summed
#endif
  EMO_S_STAMP(0)
  EMO_S_DECODE(it_, l_base + idx8)
  int nx_L = 0, nx_cotile = 0, nx_ks = 0, nx_n = 0, nx_ptile = 0, nx_x0 = 0, nx_y0 = 0, nx_z0 = 0, nx_st_begin = 0, nx_st_end = 0;
  bool chain_out = false, nxq_ok = false;
  unsigned nxq_off = 0;
  int nx_cc0 = 0, nx_kd0 = 0;
  if (CHAIN && idx8 + l_stride < n_mine) {
    EMO_S_DECODE(nx_, l_base + idx8 + l_stride)
    const int nst_ = it_st_end - it_st_begin, nxst_ = nx_st_end - nx_st_begin;
    chain_out = nx_n == it_n && nst_ >= 2 && (nst_ & 1) == 0 && nxst_ >= 2 && (nxst_ & 1) == 0;
    // the next item's load cursor, ready for the stage that switches to it (the K loop stays free of branches: with one
    // around the switch the compiler moved 109 accumulators to ordinary registers and back in every stage pair)
    EMO_S_CURSOR_OF(nx_, nxq_ok, nxq_off)
    nx_cc0 = __builtin_amdgcn_readfirstlane(nx_st_begin / a.KD);
    nx_kd0 = nx_st_begin - nx_cc0 * a.KD;
  }
  // (what no path below reads before writing it -- the fragment sets, the first raw patch buffer, the pixel under conversion --
  // is declared dead here: carried around the item loop as live values, the allocator kept second copies of ~100 registers in
  // accumulation registers, refreshed in every stage pair of the K loop)
#pragma unroll
  for (int st_ = 0; st_ < 2; ++st_)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(fa_[st_][pl][i]));
#pragma unroll
      for (int j = 0; j < TP; ++j) asm volatile("" : "=v"(fb_[st_][pl][j]));
    }
#pragma unroll
  for (int u = 0; u < 8; ++u) asm volatile("" : "=v"(qv[0][u]));
  asm volatile("" : "=v"(cv_h));
  asm volatile("" : "=v"(cv_m));
  if constexpr (SPLIT == 3) asm volatile("" : "=v"(cv_l));
  if (CHAIN && chained_in) {
    // P[0] holds the converted patch of the first stage, W[0] its kernel rows, qv[1] the landed loads of the second stage, q_sc /
    // q_sh its first table entries (the state every stage leaves to the next); the tables are the sample's.  What is left:
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    if (tid < BM) smem[Cfg::OFF_BIAS_F + (tid >> 5) * 32 + (tid & 3) * 8 + ((tid & 31) >> 2)] = te_b;
    const char* const w1_ = it_wsrc + (long)(it_st_begin + 1) * (3 * Cfg::WROW_BYTES);
    EMO_S_DMA_STAGE_PIECE(w1_, 1, 0)
    EMO_S_DMA_STAGE_PIECE(w1_, 1, 1)
    EMO_S_BARRIER(2);
  } else {
  EMO_S_PROLOGUE_ISSUE(it_, true)
  // ---- prologue, second part: the tables into LDS, the first patch converted into P[0]; it leaves the state every stage leaves
  //      to the next: buffer 1 holds the landed loads of the second stage, the tables are that stage's, DMA(first stage, row 2)
  //      is in flight, fragment set 0 holds step 0 ----
#pragma unroll
  for (int k = 0; k < NTE; ++k) {       // (without an affine the index wraps at SCT: identity entries)
    const int c = tid + 256 * k;
    if (c < min(a.Cin, Cfg::SCT)) {
      sct[c] = te_sc[k] * in_scale;      // (in_scale: 1, or the fp16 split's power of two)
      sct[Cfg::SCT + c] = te_sh[k] * in_scale;
    }
  }
  // bias of the block's channels in the order the epilogue's row layout reads it (conv_epilogue_rows)
  if (tid < BM) smem[Cfg::OFF_BIAS_F + (tid >> 5) * 32 + (tid & 3) * 8 + ((tid & 31) >> 2)] = te_b;
  EMO_S_WAIT(0);
  EMO_S_TOUCH_QUAD(0)
  EMO_S_TOUCH_QUAD(1)
  __syncthreads();   // scale / shift tables visible
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    EMO_S_HALF_TABLE(0, 0)
    EMO_S_CONV_HALF(0, Cfg::OFF_P, i, 0)
    EMO_S_HALF_TABLE(0, 1)
    EMO_S_CONV_HALF(0, Cfg::OFF_P, i, 1)
  }
  EMO_S_HALF_TABLE(1, 0)                 // (what step 0 of the first stage converts with)
  if constexpr (ONEBAR) {
    // the first two pieces of the second stage's weights (its other seven follow in steps 0 .. 4 of the first stage)
    const char* const w1_ = it_wsrc + (long)((it_st_begin + 1) < it_st_end ? it_st_begin + 1 : it_st_begin) * (3 * Cfg::WROW_BYTES);
    EMO_S_DMA_STAGE_PIECE(w1_, 1, 0)
    EMO_S_DMA_STAGE_PIECE(w1_, 1, 1)
    EMO_S_BARRIER(2);                    // (LDS stores of P[0] visible; the two pieces stay in flight)
  } else {
    // the first two pieces of the first stage's kernel row 2 (its other three follow in steps 0 and 1)
    EMO_S_DMA_PIECE(it_wsrc + (long)it_st_begin * (3 * Cfg::WROW_BYTES), 2, 0)
    EMO_S_DMA_PIECE(it_wsrc + (long)it_st_begin * (3 * Cfg::WROW_BYTES), 2, 1)
    EMO_S_BARRIER(2);                    // (LDS stores of P[0] visible; the two pieces stay in flight)
  }
  }

  // ---- K loop, two stages per iteration (buffer parities and fragment sets are compile-time constants).  Stage cg, parity par:
  //        steps 0 .. 7   half a pixel each of the patch of stage cg + 1 is converted from qv[par ^ 1] into P[par ^ 1]
  //        MIDBAR(cg, 0) (top of step 2): DMA(cg, 1) has landed [vmcnt NDMA: DMA(cg, 2) may fly]; then DMA(cg + 1, 0) and the
  //                       quad loads of stage cg + 2 into qv[par] (free since the previous stage's step 7)
  //        MIDBAR(cg, 1) (top of step 5): DMA(cg, 2) has landed [vmcnt NDMA + 8]; then DMA(cg + 1, 1)
  //        MIDBAR(cg, 2) (top of step 8): DMA(cg + 1, 0) and the quad loads have landed [vmcnt NDMA: DMA(cg + 1, 1) may fly];
  //                       every wave has stored its share of P[par ^ 1] and is past its last read of weight row 2; then
  //                       DMA(cg + 1, 2) and the first fragments of stage cg + 1 from P[par ^ 1]
  //      every wave issues exactly NDMA pieces per row and 8 quad loads per stage, so the counts are uniform ----
  EMO_S_STAMP(1)
  // (the accumulators are cleared HERE, in the block in front of the K loop: cleared in front of the two prologue forms, the
  // compiler carried them through both as ordinary registers and moved 109 of them to their accumulation registers and back
  // in every stage pair)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TPH; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_lo[i][j][r] = 0.0f; acc_hi[i][j][r] = 0.0f; }
  EMO_S_LOAD_FRAGS(0, Cfg::OFF_W, Cfg::OFF_P, 0, 0)       // (first stage: W[0], P[0])
  for (int cg0 = it_st_begin; cg0 < it_st_end; cg0 += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int cg = cg0 + par;
      if (cg >= it_st_end) break;
      const int pcur = Cfg::OFF_P + par * PBUF, pnxt = Cfg::OFF_P + (par ^ 1) * PBUF;
      // stage cg + 1 (its weight rows are fetched during this stage) and stage cg + 2 (its patch is loaded during this stage),
      // clamped to the last stage: a harmless re-stage
      // (chained: past the item's end the sequence continues with the next item's stages)
      const bool nx1_ = CHAIN && par == 1 && chain_out && cg + 1 >= it_st_end, nx2_ = CHAIN && chain_out && cg + 2 >= it_st_end;
      // (pointers from a SELECTED stage index -- 32-bit selects, no branch in the loop; the packed rows of (channel tile c, stage
      // k) sit at index c * nstages_all + k)
      const int g1_ = nx1_ ? nx_cotile * nstages_all + nx_st_begin + (cg + 1 - it_st_end)
                           : it_cotile * nstages_all + ((cg + 1) < it_st_end ? cg + 1 : it_st_end - 1);
      const int g2_ = nx2_ ? nx_cotile * nstages_all + nx_st_begin + (cg + 2 - it_st_end)
                           : it_cotile * nstages_all + ((cg + 2) < it_st_end ? cg + 2 : it_st_end - 1);
      const char* const dma_ptr = reinterpret_cast<const char*>(a.wpk) + (long)g1_ * (3 * Cfg::WROW_BYTES);
      const char* const dma_ptr2 = reinterpret_cast<const char*>(a.wpk) + (long)g2_ * (3 * Cfg::WROW_BYTES);
      // (the last stage of a chained item leaves W[1] alone: it is the epilogue's scratch; the chained prologue fetches the pieces)
      const bool skip_w2_ = CHAIN && par == 1 && chain_out && cg + 1 >= it_st_end;
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int gs = 0; gs < 9; ++gs) {
        const int r = gs / 3, s = gs % 3;
        const int fcur = (par * 9 + gs) & 1, fnxt = fcur ^ 1;     // fragment register sets of this step and the next
#if EMO_S_TIMING == 3
        const unsigned long long ts0_ = __builtin_amdgcn_s_memtime();
#endif
        // ---- barriers.  Every vector-memory instruction of a stage has a fixed step (table below), at most four per step and
        //      wave: issued in bursts behind the barriers (5 weight pieces + 8 loads at once, by all four waves) they queued at
        //      the CU's address unit (one wave-instruction per ~17 cycles) and the waves -- in-order -- stood with them:
        //      260 / 820 / 1050 cycles on top of the 768 of a barrier step (profiles/r4_conv_phase_steps*.jsonl) ----
        if (ONEBAR) {
          // one-barrier schedule (two whole weight stages in LDS): the only barrier of the stage, at the top of step 8.  Every
          // wave has read its last fragments of this stage (buffers W[par], P[par] are free behind it), stored its share of
          // the next patch, and drained ALL its loads -- the next stage's weights and the patch registers of the one behind
          // it were issued by step 4 at the latest, three steps ago (vmcnt(0): nothing to count)
          if (gs == 8) { EMO_S_LOOP_BARRIER(0) }
        } else if (s == 2) {
          // MIDBAR(cg, r), three-row schedule.  Issue order of a stage and wave (R = weight pieces of a kernel row, Q = quad loads):
          //   step 8 (stage cg - 1)  R2(cg) x2      step 0  R2(cg) x2, Q x2     step 1  R2(cg) x1, Q x2
          //   step 2  R0(cg + 1) x2, Q x2           step 3  R0(cg + 1) x2, Q x2  step 4  R0(cg + 1) x1
          //   step 5  R1(cg + 1) x2                 step 6  R1(cg + 1) x2        step 7  R1(cg + 1) x1
          //   MIDBAR(cg, 0), top of step 2: needs R1(cg) (steps 5 - 7 of the stage before); behind it 2 + 4 + 3 = 9 may fly
          //   MIDBAR(cg, 1), top of step 5: needs R2(cg) (last piece: first instruction of step 1); behind it 2 + 4 + 4 + 1 = 11
          //   MIDBAR(cg, 2), top of step 8: needs R0(cg + 1) and every Q (last: step 4); behind it 2 + 2 + 1 = 5
          if (EMO_S_ABLATE & 5) { EMO_S_LOOP_BARRIER(0) }       // (ablation builds issue fewer loads: drain instead of counting)
          else if (r == 0) { EMO_S_LOOP_BARRIER(9) } else if (r == 1) { EMO_S_LOOP_BARRIER(11) } else { EMO_S_LOOP_BARRIER(5) }
        }
        if (gs == 8) EMO_S_TOUCH_QUAD(par)       // (the loads of stage cg + 2 have landed behind the barrier above)
        if (gs == 0) {
          // the patch loads of stage cg + 2 (clamped to the last stage: a harmless re-stage).  Chained item: past its end they
          // are the next item's first two stages -- stage counts of chained items are even, so the cursor moves on to the next
          // item's tile in a stage of parity 0 (selects, no branch) and takes a plain step in the stage behind it
          const bool sw_ = CHAIN && par == 0 && chain_out && cg + 2 == it_st_end;
          const int tgt_ = (CHAIN && par == 1 && chain_out && cg + 2 > it_st_end) ? nx_st_begin + 1
                           : (cg + 2) < it_st_end ? cg + 2 : it_st_end - 1;
          if (CHAIN && par == 0) {
            lq_ok = sw_ ? nxq_ok : lq_ok;
            lq_off = sw_ ? nxq_off : lq_off;
            lq_z0 = sw_ ? nx_z0 : lq_z0;
            const int adv_ = (!sw_ && tgt_ != ld_stage) ? 1 : 0;
            int kd_ = ld_kd + adv_, cc_ = ld_cc;
            if (kd_ == a.KD) { kd_ = 0; ++cc_; }
            ld_stage = sw_ ? nx_st_begin : ld_stage + adv_;
            ld_cc = sw_ ? nx_cc0 : cc_;
            ld_kd = sw_ ? nx_kd0 : kd_;
            EMO_S_SET_STAGE_VARS()
          } else {
            EMO_S_SET_STAGE_STEP(tgt_);
          }
          EMO_S_ISSUE_BEGIN(par)
        }
        // ---- one step: the fragments of the NEXT step, half a pixel of conversion, 4 * NPROD MFMAs.  The order inside the step
        //      is pinned with sched_group_barrier: the compiler left to itself sinks the fragment reads to their first use
        //      (LDS latency in front of every MFMA) and emits the conversion as one VALU burst.  The step's vector-memory
        //      instructions sit BETWEEN the planes of the fragment reads: their asm statements order against the LDS reads on
        //      both sides, which the pinning spreads one per MFMA, so they issue among the MFMAs ----
        __builtin_amdgcn_sched_barrier(0);
        {
          const int rn = gs < 8 ? (gs + 1) / 3 : 0, sn = gs < 8 ? (gs + 1) % 3 : 0;
          const int wbn = Cfg::OFF_W + (ONEBAR ? (gs < 8 ? par : par ^ 1) * Cfg::WSTAGE : 0) + rn * WROW, pbn = gs < 8 ? pcur : pnxt;
          const char* const dma_ptr0 = it_wsrc + (long)cg * (3 * Cfg::WROW_BYTES);       // this stage (three-row schedule: its row 2)
#pragma unroll
          for (int pl = 0; pl < NPL + 1; ++pl) {
            if (pl < NPL) EMO_S_LOAD_FRAGS_PLANE(fnxt, pl, wbn, pbn, rn, sn)
            if (ONEBAR) {
              // weights of stage cg + 1 into W[par ^ 1]: pieces 2 .. 8 in steps 0 .. 4; the first two pieces of stage cg + 2 into
              // W[par] behind the barrier of step 8; the quad loads two per step in steps 0 .. 3
              static_assert(!ONEBAR || (NPL == 2 && (Cfg::NSTP == 9 || Cfg::NSTP == 5)), "piece schedule of the one-barrier stage");
              if (gs == 8 && pl < 2) {
                // (chained item, last stage: the two pieces go to the patch buffer this stage has finished with, a dump)
                const int j_ = wave + 4 * pl;
                const unsigned dst_ = smem_lds + (skip_w2_ ? (unsigned)((Cfg::OFF_P + par * PBUF) * 16 + j_ * 1024)
                                                           : (unsigned)((Cfg::OFF_W + par * Cfg::WSTAGE) * 16 + j_ * 1024));
                emo_dma16_pinned_s(dma_ptr2 + j_ * 1024, lane16, dst_);
              }
              if (gs < 4 && pl == 0 && 2 + gs < Cfg::NSTP) EMO_S_DMA_STAGE_PIECE(dma_ptr, par ^ 1, 2 + gs)
              if (gs < 4 && pl == 1) EMO_S_ISSUE_LOADS(par, 2 * gs, 2 * gs + 2)
              if (gs == 4 && 6 + pl < Cfg::NSTP) EMO_S_DMA_STAGE_PIECE(dma_ptr, par ^ 1, 6 + pl)
            } else if (!(EMO_S_ABLATE & 1)) {
              static_assert(ONEBAR || (NPL == 3 && Cfg::NDMA == 5), "piece schedule of the three-row stage");
              if (gs == 8 && pl < 2) EMO_S_DMA_PIECE(dma_ptr, 2, pl)                      // row 2 of stage cg + 1
              if (gs == 0 && pl < 2) EMO_S_DMA_PIECE(dma_ptr0, 2, 2 + pl)                 // row 2 of THIS stage, continued
              if (gs == 1 && pl == 0) EMO_S_DMA_PIECE(dma_ptr0, 2, 4)
              if ((gs == 2 || gs == 3) && pl < 2) EMO_S_DMA_PIECE(dma_ptr, 0, 2 * (gs - 2) + pl)
              if (gs == 4 && pl == 0) EMO_S_DMA_PIECE(dma_ptr, 0, 4)
              if ((gs == 5 || gs == 6) && pl < 2) EMO_S_DMA_PIECE(dma_ptr, 1, 2 * (gs - 5) + pl)
              if (gs == 7 && pl == 0) EMO_S_DMA_PIECE(dma_ptr, 1, 4)
            }
            if (!ONEBAR && !(EMO_S_ABLATE & 4)) {
              if ((gs == 0 || gs == 2 || gs == 3) && pl == 2) EMO_S_ISSUE_LOADS(par, gs == 0 ? 0 : 2 * gs, gs == 0 ? 2 : 2 * gs + 2)
              if (gs == 1 && pl == 1) EMO_S_ISSUE_LOADS(par, 2, 4)
            }
          }
        }
        if (gs < 8) {
          EMO_S_CONV_HALF(par ^ 1, pnxt, gs >> 1, gs & 1)
          // scale / shift of the NEXT step's four channels, read behind this step's last use of the registers: a read at the top
          // of the step that uses it stalls the step's first VALU -- and, in order, every MFMA behind it -- for the LDS latency
          // (+130 cycles per step, profiles/r4_conv_phase_steps_spread.jsonl)
          if (gs < 7) { EMO_S_HALF_TABLE(par ^ 1, (gs + 1) & 1) } else { EMO_S_HALF_TABLE(par, 0) }
        }
#pragma unroll
        for (int p = 0; p < NPROD; ++p) {
          const int pa = SPLIT == 2 ? PA3[p] : PA6[p + 6 - NPROD], pb = SPLIT == 2 ? PB3[p] : PB6[p + 6 - NPROD];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              // operands swapped: the result tile is [position][channel] (conv_epilogue).  The leading product accumulates in
              // acc_lo, the small ones in acc_hi (header comment)
              floatx16& acc_ = (pa == 0 && pb == 0) ? acc_lo[i][j] : acc_hi[i][j];
              if constexpr (SPLIT == 3) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb_[fcur][pb][j], fa_[fcur][pa][i], acc_, 0, 0, 0);
              else acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[fcur][pb][j], fa_[fcur][pa][i], acc_, 0, 0, 0);
            }
        }
        if (EMO_S_PIN && TM == 1) {
          // 32-row tile: 6 MFMAs and 6 fragment reads per step -- { MFMA, two reads, VALU } x 3, then { MFMA, VALU, LDS store } x 3
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
          }
#pragma unroll
          for (int k = 3; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        } else if (EMO_S_PIN) {
          // { MFMA, fragment read, <= 5 VALU } for the 4 * NPL reads of the step, then { MFMA, <= 6 VALU, LDS store }
#pragma unroll
          for (int k = 0; k < 4 * NPL; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          }
#pragma unroll
          for (int k = 4 * NPL; k < 4 * NPROD; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#if EMO_S_TIMING == 3
        tstep[gs] += __builtin_amdgcn_s_memtime() - ts0_;
#endif
      }
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
  }
  const int ep_L = it_L, ep_nst = it_st_end - it_st_begin;   // (measurement builds log under the item's own index)
  EMO_S_STAMP(2)
  {
    // transposition scratch: the patch buffer of the last stage; a weight stage buffer (one-barrier schedule: the one the last
    // stage did not read, which holds the dead re-staged rows -- chained item: the one it did read, the other holds the next
    // item's first stage); or a region of its own
    const int last_par = (it_st_end - it_st_begin - 1) & 1;
    float* const scratch = smem + (Cfg::EPI_IN_PATCH ? (Cfg::OFF_P + last_par * PBUF) * 4
                                   : Cfg::EPI_IN_W ? (Cfg::OFF_W + ((CHAIN && chain_out) ? last_par : (last_par ^ 1)) * Cfg::WSTAGE) * 4
                                   : Cfg::OFF_EPI_F) + wave * Cfg::EPI_WAVE;
    // the item the epilogue writes
    const int ep_n = it_n, ep_cotile = it_cotile, ep_ptile = it_ptile, ep_ks = it_ks, ep_x0 = it_x0, ep_y0 = it_y0, ep_z0 = it_z0;
    // Between the K loop and the epilogue: the re-issued loads / DMA of the clamped last stages are dead and are drained (a
    // chained item's are the next item's first stages: they have landed behind this), and every wave must be past its last
    // fragment read of the item's last stage before its buffers become the epilogue's scratch
    EMO_S_WAIT(0);
    EMO_S_STAMP(5)
    __syncthreads();
    EMO_S_STAMP(6)
    if (CHAIN && chain_out && tid < BM && a.bias != nullptr && a.partial == nullptr) {   // the next item's bias entries
      const int co_ = nx_cotile * BM + tid;
      te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];
    }
#define EMO_S_EPI_FAST(RES_)                                                                                                      \
    {                                                                                                                              \
      floatx4 rv_[8];                                                                                                              \
      conv_epilogue_fast_issue<TW, TP, BM, RES_, 0>(a, rv_, ep_n, ep_cotile, ep_x0, ep_y0, ep_z0, wp, lane);                       \
      conv_epilogue_fast_finish<TW, TM, TP, WGP, BM, SPLIT, Cfg::EPI_ROWF, RES_>(a, acc_lo, acc_hi, rv_, scratch,                  \
                                                                                 smem + Cfg::OFF_BIAS_F, smem + Cfg::OFF_STAT_F, ep_n, \
                                                                                 ep_cotile, ep_ptile, ep_x0, ep_y0, ep_z0, wp, half, \
                                                                                 l32, lane, tid EMO_S_TSTAMP_ARG);                  \
    }
    if (epi_mode == 1) EMO_S_EPI_FAST(1)
    else if (epi_mode == 2) EMO_S_EPI_FAST(2)
    else if (epi_mode == 0) EMO_S_EPI_FAST(0)
    else
      conv_epilogue_rows<TR, TW, TM, TP, WGP, BM, SPLIT, Cfg::EPI_ROWF>(a, acc_lo, acc_hi, scratch, smem + Cfg::OFF_BIAS_F,
                                                                        smem + Cfg::OFF_STAT_F, ep_n, ep_cotile, ep_ptile, ep_ks, ep_x0,
                                                                        ep_y0, ep_z0, wp, half, l32, lane, tid EMO_S_TSTAMP_ARG);
#undef EMO_S_EPI_FAST
  }
  if constexpr (SPLIT == 2) {
    if (a.sat_flag != nullptr && sat_m > 65504.0f) *a.sat_flag = 1;   // (every writer stores the same value)
  }
#if EMO_S_TIMING
  EMO_S_STAMP(3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  EMO_S_STAMP(4)
  if (tid == 0 && ep_L < EMO_S_TLOG_N) {
    unsigned long long* t_ = emo_s_tlog + (long)ep_L * EMO_S_TLOG_W;
#pragma unroll
    for (int k = 0; k < 12; ++k) t_[k] = tstamp[k];
    t_[12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
    t_[13] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    t_[14] = (unsigned long long)blockIdx.x;
  }
#if EMO_S_TIMING == 3
  if (lane == 0 && ep_L < EMO_S_TLOG_N / 8) {   // per-step cycles of every wave (rows N/4 .. 3N/4)
    unsigned long long* w_ = emo_s_tlog + ((long)EMO_S_TLOG_N / 4 + (long)ep_L * 4 + wave) * EMO_S_TLOG_W;
#pragma unroll
    for (int k = 0; k < 9; ++k) w_[k] = tstep[k];
    w_[9] = (unsigned long long)ep_nst;
  }
#endif
#if EMO_S_TIMING == 2
  if (lane == 0 && ep_L < EMO_S_TLOG_N / 8) {   // per-wave barrier accounting in rows N/4 .. 3N/4 of the log (first N/8 items)
    unsigned long long* w_ = emo_s_tlog + ((long)EMO_S_TLOG_N / 4 + (long)ep_L * 4 + wave) * EMO_S_TLOG_W;
    w_[0] = tw_wait; w_[1] = tw_bar; w_[2] = tw_n; w_[3] = tstamp[2] - tstamp[1];
  }
#endif
#endif
  // the next prologue overwrites the tables, the statistics exchange and (conversion) the patch buffer the epilogue transposed
  // through: every wave must be out of the epilogue first
  __syncthreads();
  chained_in = chain_out;
  }
#undef EMO_S_SET_STAGE_VARS
#undef EMO_S_SET_STAGE_INIT
#undef EMO_S_SET_STAGE_STEP
#undef EMO_S_ISSUE_BEGIN
#undef EMO_S_ISSUE_LOADS
#undef EMO_S_HALF_TABLE
#undef EMO_S_B_OFF
#undef EMO_S_CONV_HALF
#undef EMO_S_TOUCH_QUAD
#undef EMO_S_DMA_PIECE
#undef EMO_S_DMA_PIECE_TO
#undef EMO_S_DMA_STAGE_PIECE
#undef EMO_S_DMA_ROW
#undef EMO_S_WAIT
#undef EMO_S_BARRIER
#undef EMO_S_LOOP_BARRIER
#undef EMO_S_LOAD_FRAGS_PLANE
#undef EMO_S_LOAD_FRAGS

#undef EMO_S_DECODE
#undef it_wsrc
#undef EMO_S_WSRC
#undef EMO_S_PROLOGUE_ISSUE
#undef EMO_S_CURSOR_TO
#undef EMO_S_CURSOR_OF
}

template <int TR, int TW, bool UPS, int SPLIT = 3, int BMT = 64>
int conv_igemm_bf16x3_launch(ConvArgs a, hipStream_t s) {
  using Cfg = ConvCfgS<TR, TW, UPS, SPLIT, BMT>;
  if (a.Wl % TW || a.Hl % TR) return EMO_ERR_UNSUPPORTED;
  if (a.Cin % 8) return EMO_ERR_UNSUPPORTED;   // whole 8-channel groups
  if (a.scale && a.Cin > Cfg::SCT) return EMO_ERR_UNSUPPORTED;   // scale / shift tables in LDS
  if ((unsigned long long)a.Cin * a.D * a.H * a.W * 4ull >= (1ull << 32)) return EMO_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  if ((reinterpret_cast<unsigned long long>(a.x) & 15ull) || (a.W & 3)) return EMO_ERR_UNSUPPORTED;     // 16-byte quads
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl;
  a.n_cchunks = (a.Cin + Cfg::KC - 1) / Cfg::KC;
  const long nt = (long)a.tiles_x * a.tiles_y * a.tiles_z;
  if (nt > 0x7fffffffL || a.N > 65535) return EMO_ERR_UNSUPPORTED;
  const int cot = (a.Cout + Cfg::BM - 1) / Cfg::BM;
  auto kern = conv_igemm_bf16x3_kernel<TR, TW, UPS, SPLIT, BMT>;
  const int rc = emo_raise_dynamic_lds(kern);
  if (rc != EMO_OK) return rc;
  // persistent blocks (min(n_work, CUs)) by default: no workgroup launch between the items of a CU (kernel comment; measured
  // +2 % on the bench step, tools/session/r4_call12.sh); EMO_CONV_BF16X3_PERSISTENT=0 launches one block per item (A/B)
  static const int persistent = [] { const char* e = getenv("EMO_CONV_BF16X3_PERSISTENT"); return e ? atoi(e) : 1; }();
  const int ncu = emo_cu_count();
  if (a.cot0 < 0 || a.cot0 >= cot) return EMO_ERR_BAD_ARG;
  a.n_cotiles = cot - a.cot0;                  // (cot0 > 0: the odd last tile behind conv_f16x2_ct2_launch's pairs)
  if (a.ksplit < 1 || (a.ksplit > 1 && !a.partial)) return EMO_ERR_BAD_ARG;
  if (a.ksplit == 1) { a.stages_per_split = a.n_cchunks * a.KD; a.partial = nullptr; }
  if (a.ksplit > 1 && a.gn_stats) return EMO_ERR_BAD_ARG;
  if (nt * a.n_cotiles * a.N * a.ksplit > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  a.n_work = (int)(nt * a.n_cotiles * a.N * a.ksplit);
  // (a guarded fallback launch is normally skipped: min(n_work, CUs) blocks read the flag and leave instead of n_work)
  const int grid = (persistent || a.run_if != nullptr) && a.n_work > ncu ? ncu : a.n_work;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)Cfg::LDS_BYTES, s, a);
  return emo_launch_status();
}
