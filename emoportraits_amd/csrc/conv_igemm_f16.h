// Implicit-GEMM convolution with fp16 MFMA operands and fp32 accumulation for gfx950 -- the reduced-precision mode of
// BASELINE.json configs[4] ("stage_2 refinement ... fp16 MFMA convs").  Opt-in per layer (PackedConv(precision="f16"));
// the exact-fp32 kernel of conv_igemm.h stays the default everywhere and is what bench.py measures.
//
// Same GEMM view, block order, K split and epilogue (shared: conv_epilogue, incl. the GroupNorm tile statistics) as
// conv_igemm.h; what differs is the operand path:
//   * v_mfma_f32_32x32x16_f16: the 16 k-values of one MFMA are 16 input CHANNELS at one tap, lane (l&31, l>>5) holds the 8
//     consecutive channels 8*(l>>5) .. +7 of its row / column, so both operands are one 16-byte LDS read per lane:
//       weights Ah[q][tap][half][BM][8]     packed like that on the host, copied by LDS-DMA
//       patch   Ph[group = 2q + half][slot][8]      (slot layout below)
//   * activations stay fp32 NC(D)HW in HBM and are staged by QUADS: a lane owns 4 consecutive x positions of one patch row
//     for the 8 channels of a group: 8 buffer_load_dwordx4 (one per channel plane, 16-byte aligned), the producer's
//     GroupNorm affine + ReLU in fp32, zero padding, saturation to +-65504, round-to-nearest-even to fp16, 4 ds_write_b128.
//     Why quads: global loads on this chip are bound by wave-INSTRUCTIONS, not bytes -- one per ~17 cycles per CU whether a
//     lane asks for 4 or 16 bytes (tools/microbench/vmem_rate.hip: 15 / 27 / 53 B/clk/CU for dword / x2 / x4).  Staging
//     pixel by pixel (one dword load per element, the first kernel of round 2) needs 394 VMEM instructions per CU and stage
//     = ~6.9k cycles against 2.3k cycles of MFMA work; quads need 8 per wave and stage (+8 dword loads in one wave for the
//     two halo columns of a 3x3 patch): a third of that, worth +5-10 % (what bounds the kernel now: DESIGN.md section 3.1).
//   * with a fused nearest x2 upsample the LDS patch holds the SOURCE pixels (a quarter of the upsampled patch): staging
//     never expands, the B-fragment slots of a lane map its output pixel and tap to (y+r-1)>>1, (x+s-1)>>1.
//   * LDS patch slots: an interior pixel (patch row pr, quad qx, i = x & 3) lives at slot i * SUB + pr * NQ1 + qx, the left /
//     right halo pixel of a row at sub-row 0 / 1 of the pseudo-quad qx = NQ of that row.  The 4 stores of a lane then go to
//     4 sub-rows, consecutive lanes to consecutive 16-byte slots, and SUB = 4 (mod 16) staggers the sub-rows over the banks
//     (measured: a fifth of the LDS cycles are still conflict cycles; LDS is busy 37 % of the time, not the limiter).
//   * a stage is KC = 16 channels (3x3: 9 MFMA steps of K = 16) or 32 (1x1: 2 steps); tile 64 output channels x 256
//     positions, 2 blocks per CU (LDS: 18 KB weights + 17 KB patch per stage, double-buffered).
//   * software pipeline ("rolling"): EVERY VMEM instruction of the K loop is inline asm -- the patch loads and the weight
//     LDS-DMA -- so the compiler inserts no vmcnt waits of its own and the hand-counted ones are exact.  The quad of stage
//     s + 2 is loaded right after the quad of stage s + 1 has been converted (during stage s): a whole stage of latency
//     budget.  MFMA fragments are read from LDS two steps ahead into three rotating register sets, the per-channel scale /
//     shift come from an LDS table one stage ahead, and the conversion work is cut in pieces pinned between the MFMAs.
//     tools/kernel_resources.py --audit checks in the ISA that no in-flight load destination is touched or spilled.
// Needs Cin % 8 == 0 (whole channel groups), 16-byte aligned input planes (W % 4 == 0 follows from the tile shapes),
// Cin <= 1024 when a scale / shift is fused, and Cin * D * H * W * 4 < 2^32 (32-bit buffer offsets).  Rounding: operands
// carry 11 significand bits, products and sums are exact fp32 MFMA accumulation.
#pragma once
#include "conv_igemm.h"

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef int emo_intx4 __attribute__((ext_vector_type(4)));

#ifndef EMO_F16_EXPERIMENT
#define EMO_F16_EXPERIMENT 0   /* measurement builds only (tools/session/r2_exp_f16.sh): 1 = one block per CU (LDS request padded
                                  to 81 KB), 2 = no sched_barrier pinning of the staging pieces between the MFMAs */
#endif

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
struct ConvCfgH {
  static constexpr int BM = WGM * TM * 32;
  static constexpr int BP = WGP * TP * 32;
  static constexpr int TAPS = KH * KW;
  static constexpr int HALO = KW / 2;                    // 1 for 3x3, 0 for 1x1 (KH == KW)
  static constexpr int TRS = UPS ? TR / 2 : TR;          // tile extent in SOURCE pixels
  static constexpr int TWS = UPS ? TW / 2 : TW;
  static constexpr int PR = TRS + 2 * HALO;              // source rows of the patch
  static constexpr int NQ = TWS / 4;                     // interior quads per row
  static constexpr int NQ1 = NQ + HALO;                  // + the pseudo-quad that holds the row's two halo pixels
  static constexpr int SUB = ((PR * NQ1 + 11) / 16) * 16 + 4;   // slots per sub-row: >= PR * NQ1 and = 4 (mod 16)
  static constexpr int CHS = 4 * SUB;                    // slots per 8-channel group
  static constexpr int KQ = KC / 16;                     // MFMA k-steps (16 channels each) per tap and stage
  static constexpr int NG = KC / 8;                      // 8-channel groups per stage: group = 2 q + half
  static constexpr int QPG = 256 / NG;                   // threads per group: thread t stages quad t % QPG of group t / QPG
  static constexpr int NHALO = HALO ? NG * 2 * PR : 0;   // halo pixels of a stage (lanes of the halo wave)
  static constexpr int NSTEPS = KQ * TAPS;
  static constexpr int ASZ_H = KC * TAPS * BM;           // halfs of one stage's weight tile
  static constexpr int PSZ_H = NG * CHS * 8;             // halfs of one stage's patch
  static constexpr int ASZ = ASZ_H / 2;                  // in floats
  static constexpr int BUF = ASZ + ((PSZ_H / 2 + 3) & ~3);
  static constexpr int SCT = 1024;                       // entries of the per-sample scale / shift tables kept in LDS
  static constexpr int LDS_USED = (2 * BUF + 256 + 2 * SCT) * 4;    // two stage buffers + 64 dump slots + scale / shift tables
  static constexpr int LDS_BYTES = EMO_F16_EXPERIMENT == 1 ? 81 * 1024 : LDS_USED;
  static constexpr int NDMA_MIN = (ASZ_H * 2) / 4096;    // LDS-DMA instructions EVERY wave issues per stage (some issue one more)
  static constexpr int BY_LDS = (160 * 1024) / LDS_BYTES;
  // 2 blocks per CU at most: 64 accumulator + 48 fragment + 40 in-flight patch + 32 scale / shift registers per lane do
  // not fit the 168-VGPR budget of 3 waves per SIMD; with 8 accumulator tiles (CFG_G) one block per CU, 512 registers
  static constexpr int OCC = (BY_LDS < 1 || TM * TP > 4) ? 1 : (BY_LDS > 2 ? 2 : BY_LDS);
  static_assert(WGM * WGP == 4, "4 waves per block");
  static_assert(TZ == 1 && TR * TW == BP, "planar position tile of BP pixels");
  static_assert(KH == KW && (KH == 1 || KH == 3), "1x1 and 3x3 kernels");
  static_assert(KC % 16 == 0 && KC <= 32, "whole 16-channel MFMA steps");
  static_assert(TWS % 4 == 0 && (!UPS || (TR % 2 == 0 && TW % 2 == 0)), "whole quads");
  static_assert(PR * NQ <= QPG, "one interior quad per thread and stage");
  static_assert(NHALO <= 64, "the halo pixels of a stage are staged by one wave");
  static_assert((ASZ_H * 2) % 16 == 0, "weight tile must be 16-byte copyable");
  static_assert(TM * TP <= 8, "accumulator budget");
  static_assert(2 * BUF >= 2 * WGP * BM, "the GroupNorm tile statistics are exchanged through the stage buffers");
};

// LDS-DMA hidden from the compiler (asm): 16 bytes per lane from `gsrc` to the wave-uniform LDS byte address `lds_dst`
// + lane * 16.  M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void emo_dma16_pinned(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// the same with a wave-uniform source base in SGPRs and a 32-bit per-lane byte offset: no 64-bit vector address arithmetic per
// piece (the offset lane * 16 is one loop-invariant register).  (s_mov, s_mov, s_nop 2: the five wait states of
// EMO_SGPR_HAZARD_NOP in front of the load that reads the base)
__device__ __forceinline__ void emo_dma16_pinned_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// raw buffer loads hidden from the compiler: address = resource base + soff (SGPR) + voff (VGPR), both in bytes.  The
// eight channel planes of a staging item differ only in soff, which is loop-invariant: no address arithmetic per load.
__device__ __forceinline__ float emo_bload_pinned(emo_intx4 rsrc, unsigned voff, unsigned soff) {
  float v;
  asm volatile(EMO_SGPR_HAZARD_NOP "buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
__device__ __forceinline__ floatx4 emo_bload4_pinned(emo_intx4 rsrc, unsigned voff, unsigned soff) {
  floatx4 v;
  asm volatile(EMO_SGPR_HAZARD_NOP "buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  return v;
}
// two loads that differ in soff only, one statement (one set of wait states)
__device__ __forceinline__ void emo_bload4x2_pinned(emo_intx4 rsrc, unsigned voff, unsigned soff0, unsigned soff1, floatx4& v0,
                                                    floatx4& v1) {
  asm volatile(EMO_SGPR_HAZARD_NOP "buffer_load_dwordx4 %0, %2, %3, %4 offen\n\tbuffer_load_dwordx4 %1, %2, %3, %5 offen"
               : "=&v"(v0), "=&v"(v1) : "v"(voff), "s"(rsrc), "s"(soff0), "s"(soff1) : "memory");
}
__device__ __forceinline__ emo_intx4 emo_raw_buffer(const void* base) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  emo_intx4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));   // stride 0: raw buffer
  r[2] = -1;                                                                      // num_records: no range limit
  r[3] = 0x00020000;                                                              // gfx9 raw-buffer data format
  return r;
}

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>::OCC,
                                   ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>::OCC)))
void conv_igemm_f16_kernel(const ConvArgs a) {
  using Cfg = ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  constexpr int BM = Cfg::BM, TAPS = Cfg::TAPS, PR = Cfg::PR, NQ = Cfg::NQ, NQ1 = Cfg::NQ1, SUB = Cfg::SUB, CHS = Cfg::CHS;
  constexpr int ASZ = Cfg::ASZ, ASZ_H = Cfg::ASZ_H, BUF = Cfg::BUF, QPG = Cfg::QPG, HALO = Cfg::HALO;
  constexpr int NSTEPS = Cfg::NSTEPS, NHALO = Cfg::NHALO, TWS = Cfg::TWS;
  constexpr int HALO_WAVE = 3;          // the wave that also stages the halo columns
  constexpr int NSLOT = TM * TP;        // MFMAs per step

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int wm = wave / WGP, wp = wave % WGP;
  const int m0 = wm * TM * 32, p0 = wp * TP * 32;

  // block -> (sample, position tile, channel tile, K split): XCD-contiguous order, channel tile fastest (conv_igemm.h)
  int ks = 0;
  const int total = gridDim.x;
  const int q8 = total >> 3, r8 = total & 7;
  const int xcd = blockIdx.x & 7, idx8 = blockIdx.x >> 3;
  const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx8;
  const int cotile = L % a.n_cotiles + a.cot0;      // (cot0 > 0: the odd last tile behind the eight-wave kernel's pairs)
  int rest = L / a.n_cotiles;
  if (a.ksplit > 1) {
    ks = rest % a.ksplit;
    rest /= a.ksplit;
  }
  const int nptiles = a.tiles_x * a.tiles_y * a.tiles_z;
  const int n = rest / nptiles;
  int bx = rest - n * nptiles;
  const int ptile = bx;
  const int tx = bx % a.tiles_x; bx /= a.tiles_x;
  const int ty = bx % a.tiles_y; bx /= a.tiles_y;
  const int tz = bx;
  const int x0 = tx * TW, y0 = ty * TR, z0 = tz * TZ;
  const int x0s = UPS ? x0 >> 1 : x0, y0s = UPS ? y0 >> 1 : y0;   // tile origin in source pixels

  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const float* xn = a.x + (long)n * a.Cin * DHW;
  const bool has_affine = a.scale != nullptr;
  const int padD = a.KD >> 1;
  const float clamp_lo = a.relu_in ? 0.0f : -65504.0f;   // lower bound of the staged value: ReLU, or the fp16 range

  const int nstages_all = a.n_cchunks * a.KD;
  const int st_begin = ks * a.stages_per_split;
  const int st_end = min(nstages_all, st_begin + a.stages_per_split);
  const char* wsrc = reinterpret_cast<const char*>(a.wpk) + ((long)cotile * nstages_all) * (ASZ_H * 2);

  // ---- staging map.  Interior: thread t owns quad t % QPG (patch row q_r, quad column q_c) of channel group t / QPG (wave-
  //      uniform: QPG is a multiple of 64); halo (3x3): lane l of HALO_WAVE owns halo pixel l (group, patch row, side).
  //      The position part is the same in every stage: plane offset, validity and LDS slot once per thread. ----
  const int q_u = tid % QPG;
  const int q_g = __builtin_amdgcn_readfirstlane(tid / QPG);
  const int q_r = q_u / NQ, q_c = q_u - q_r * NQ;
  const int q_y = y0s - HALO + q_r;
  const bool q_live = q_u < PR * NQ;                                       // the thread has a quad at all
  const bool q_ok = q_live && (unsigned)q_y < (unsigned)a.H;              // ... inside the plane (x always is)
  const unsigned q_off = q_ok ? (unsigned)(q_y * a.W + x0s + 4 * q_c) * 4u : 0u;
  halfx8* const dump8 = reinterpret_cast<halfx8*>(smem + 2 * BUF) + lane;   // per-lane dump slot (written, never read)
  const int q_slot = q_g * CHS + q_r * NQ1 + q_c;                          // + i * SUB for pixel i of the quad

  const bool is_halo_wave = HALO && wave == HALO_WAVE;
  const int h_g = lane / (2 * PR), h_rem = lane - h_g * (2 * PR);
  const int h_r = h_rem >> 1, h_side = h_rem & 1;
  const int h_y = y0s - HALO + h_r, h_x = h_side ? x0s + TWS : x0s - 1;
  const bool h_live = lane < NHALO;
  const bool h_ok = h_live && (unsigned)h_y < (unsigned)a.H && (unsigned)h_x < (unsigned)a.W;
  const unsigned h_off = h_ok ? (unsigned)(h_y * a.W + h_x) * 4u : 0u;
  const int h_slot = h_g * CHS + h_side * SUB + h_r * NQ1 + NQ;

  constexpr int TPH = TP > 2 ? TP / 2 : TP;
  floatx16 acc_lo[TM][TPH], acc_hi[TM][TPH];
#define acc_at(i_, j_) ((j_) < TPH ? acc_lo[i_][(j_) < TPH ? (j_) : 0] : acc_hi[i_][(j_) >= TPH ? (j_) - TPH : 0])
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TPH; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_lo[i][j][r] = 0.0f; acc_hi[i][j][r] = 0.0f; }

  // lane bases into the LDS tiles, in units of 8 halfs (16 bytes).  Patch: slot of the lane's output pixel for every tap
  // (without an upsample the row part is a constant the compiler folds into the instruction offset)
  const int a_base = half * BM + m0 + l32;
  int b_slot[TP][KH][KW];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW, row = p / TW;
#pragma unroll
    for (int r = 0; r < KH; ++r)
#pragma unroll
      for (int s = 0; s < KW; ++s) {
        // source pixel of tap (r, s), relative to the patch origin (row 0 = y0s - HALO, column -1 = left halo)
        const int pr = UPS ? ((row + r - HALO + 2) >> 1) - 1 + HALO : row + r;
        const int pc = UPS ? ((col + s - HALO + 2) >> 1) - 1 : col + s - HALO;
        const int slot = pc < 0 ? pr * NQ1 + NQ : (pc >= TWS ? SUB + pr * NQ1 + NQ : (pc & 3) * SUB + pr * NQ1 + (pc >> 2));
        b_slot[j][r][s] = half * CHS + slot;
      }
  }

  // fragments of an MFMA step (TM weight and TP patch fragments, 16 bytes per lane each) in three rotating register sets:
  // the reads of step s + 2 are issued before the MFMAs of step s, so their LDS latency hides behind a whole step of MFMAs
  halfx8 fa_[3][TM], fb_[3][TP];
#define EMO_H_LOAD_FRAGS(set_, step_)                                                                 \
  {                                                                                                   \
    const int q = (step_) / TAPS, tap = (step_) % TAPS;                                               \
    const int r = tap / KW, s = tap % KW;                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa_[set_][i] = Ah[a_base + ((q * TAPS + tap) * 2) * BM + i * 32]; \
    _Pragma("unroll") for (int j = 0; j < TP; ++j) fb_[set_][j] = Ph[b_slot[j][r][s] + (q * 2) * CHS]; \
  }

  // ---- rolling prefetch: the quad / halo loads of stage s+2 are issued during stage s, right after the registers were
  //      converted for stage s+1 (schedule and vmcnt counts: at the K loop below).  The per-sample scale / shift vectors
  //      live in LDS (two SCT-entry tables, identity when the layer has no affine).
  float* const sct = smem + 2 * BUF + 256;
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);

  floatx4 qv[8];            // the quad: 4 pixels of channel plane u (pinned asm loads, in flight across the stage boundary)
  float hv[8];              // the halo pixel: channel plane u
  float q_lo, q_hi, h_lo, h_hi;   // per-lane clamp of the converted values: [ReLU or -65504, 65504], or [0, 0] where the
                                  // pixel is zero padding (outside the plane / volume / channel range): v_med3 does both
  floatx4 q_sc[2], q_sh[2], h_sc[2], h_sh[2];   // scale / shift of the 8 channels, read from the tables one stage ahead
  const emo_intx4 xrs = emo_raw_buffer(xn);
  unsigned usoff[8];        // byte offset of channel plane u inside an 8-channel group
#pragma unroll
  for (int u = 0; u < 8; ++u) usoff[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)u * (unsigned)DHW * 4u));

  int q_tix, h_tix;         // index of the 8 channels' scale / shift in the tables (in 16-byte units)
  int n_ci0, n_zu;          // stage being loaded: first input channel, depth slice
  bool n_zv;
#define EMO_H_SET_STAGE(stage_)                                                                       \
  {                                                                                                   \
    const int cc_ = (stage_) / a.KD;                                                                  \
    n_ci0 = cc_ * KC;                                                                                 \
    n_zu = z0 + ((stage_) - cc_ * a.KD) - padD;                                                       \
    n_zv = (unsigned)n_zu < (unsigned)a.D;                                                            \
  }
// loads of the quad / halo pixel of the stage set by EMO_H_SET_STAGE, clamp bounds and scale / shift of its 8 channels.
// Byte offsets inside the sample are 32-bit (the launcher checks Cin * D * H * W * 4 < 2^32).
#define EMO_H_ISSUE_QUAD()                                                                            \
  {                                                                                                   \
    const int c0_ = n_ci0 + q_g * 8;                                                                  \
    const bool cv_ = c0_ < a.Cin;                                                                     \
    const int cs_ = cv_ ? c0_ : 0;                                                                    \
    const bool keep_ = q_ok && cv_ && n_zv;                                                           \
    q_lo = keep_ ? clamp_lo : 0.0f;                                                                   \
    q_hi = keep_ ? 65504.0f : 0.0f;                                                                   \
    const unsigned vo_ = q_off + ((unsigned)cs_ * (unsigned)DHW + (unsigned)((n_zv ? n_zu : 0) * HW)) * 4u; \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) qv[u] = emo_bload4_pinned(xrs, vo_, usoff[u]);      \
    q_tix = (has_affine ? cs_ : (cs_ & (Cfg::SCT - 1))) >> 2;                                         \
  }
#define EMO_H_QUAD_TABLE()                                                                            \
  {                                                                                                   \
    const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + q_tix;                               \
    q_sc[0] = t4_[0]; q_sc[1] = t4_[1]; q_sh[0] = t4_[Cfg::SCT / 4]; q_sh[1] = t4_[Cfg::SCT / 4 + 1]; \
  }
#define EMO_H_ISSUE_HALO()                                                                            \
  {                                                                                                   \
    const int c0_ = n_ci0 + h_g * 8;                                                                  \
    const bool cv_ = c0_ < a.Cin;                                                                     \
    const int cs_ = cv_ ? c0_ : 0;                                                                    \
    const bool keep_ = h_ok && cv_ && n_zv;                                                           \
    h_lo = keep_ ? clamp_lo : 0.0f;                                                                   \
    h_hi = keep_ ? 65504.0f : 0.0f;                                                                   \
    const unsigned vo_ = h_off + ((unsigned)cs_ * (unsigned)DHW + (unsigned)((n_zv ? n_zu : 0) * HW)) * 4u; \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) hv[u] = emo_bload_pinned(xrs, vo_, usoff[u]);       \
    h_tix = (has_affine ? cs_ : (cs_ & (Cfg::SCT - 1))) >> 2;                                         \
  }
#define EMO_H_HALO_TABLE()                                                                            \
  {                                                                                                   \
    const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + h_tix;                               \
    h_sc[0] = t4_[0]; h_sc[1] = t4_[1]; h_sh[0] = t4_[Cfg::SCT / 4]; h_sh[1] = t4_[Cfg::SCT / 4 + 1]; \
  }
// transform in fp32 (affine of the producer's GroupNorm; ReLU, saturation and zero padding in one v_med3 -- the padding
// applies to the transformed tensor: bounds [0, 0]), round to fp16, one 16-byte ds_write per pixel.  Pixels i0_, i0_ + 1 of
// the quad per call (two calls per stage, pinned behind different MFMAs).
#define EMO_H_STORE_QUAD(buf_, i0_)                                                                   \
  {                                                                                                   \
    halfx8* Ph_ = reinterpret_cast<halfx8*>((buf_) + ASZ);                                            \
    halfx8* d_ = q_live ? Ph_ + q_slot : dump8;                                                       \
    _Pragma("unroll") for (int i = (i0_); i < (i0_) + 2; ++i) {                                       \
      halfx8 h_;                                                                                      \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                 \
        float v = __fmaf_rn(qv[u][i], q_sc[u / 4][u % 4], q_sh[u / 4][u % 4]);                        \
        v = __builtin_amdgcn_fmed3f(v, q_lo, q_hi);                                                   \
        h_[u] = (_Float16)v;                                                                          \
      }                                                                                               \
      d_[q_live ? i * SUB : 0] = h_;                                                                  \
    }                                                                                                 \
  }
#define EMO_H_STORE_HALO(buf_)                                                                        \
  {                                                                                                   \
    halfx8* Ph_ = reinterpret_cast<halfx8*>((buf_) + ASZ);                                            \
    halfx8 h_;                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                   \
      float v = __fmaf_rn(hv[u], h_sc[u / 4][u % 4], h_sh[u / 4][u % 4]);                             \
      v = __builtin_amdgcn_fmed3f(v, h_lo, h_hi);                                                     \
      h_[u] = (_Float16)v;                                                                            \
    }                                                                                                 \
    *(h_live ? Ph_ + h_slot : dump8) = h_;                                                            \
  }
#define EMO_H_TOUCH_QUAD() { _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch4(qv[u]); }
#define EMO_H_TOUCH_HALO() { _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch(hv[u]); }
// weight tile of one stage by LDS-DMA (1 KiB per wave-instruction), lane-linear = the packed order
#define EMO_H_DMA_WEIGHTS(stage_, dst_lds_)                                                           \
  {                                                                                                   \
    const char* ws_ = wsrc + (long)(stage_) * (ASZ_H * 2);                                            \
    constexpr int NGL = (ASZ_H * 2 + 4095) / 4096;                                                    \
    _Pragma("unroll") for (int i = 0; i < NGL; ++i) {                                                 \
      const int j = wave + 4 * i;                                                                     \
      const int boff = j * 1024 + lane * 16;                                                          \
      /* the first NDMA_MIN rounds are whole for every wave: unconditional, so that the vmcnt counts hold */ \
      if (i < Cfg::NDMA_MIN || boff < ASZ_H * 2) emo_dma16_pinned(ws_ + boff, (dst_lds_) + (unsigned)(j * 1024)); \
    }                                                                                                 \
  }
#define EMO_H_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define EMO_H_BARRIER(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")
#define EMO_H_STAGE_BARRIER()                                                                         \
  {                                                                                                   \
    if (is_halo_wave) { EMO_H_BARRIER(16); } else { EMO_H_BARRIER(8); }                               \
  }

  // ---- prologue: stage st_begin into buffer 0, issue the loads of stage st_begin + 1.  The first loads go out before the
  //      scale / shift tables are filled, so that the two latencies overlap ----
  EMO_H_DMA_WEIGHTS(st_begin, smem_lds);
  EMO_H_SET_STAGE(st_begin);
  EMO_H_ISSUE_QUAD()
  if (is_halo_wave) EMO_H_ISSUE_HALO()
  for (int c = tid; c < min(a.Cin, Cfg::SCT); c += 256) {   // (without an affine the index wraps at SCT: identity entries)
    const bool real = has_affine && c < a.Cin;
    sct[c] = real ? a.scale[(long)n * a.Cin + c] : 1.0f;
    sct[Cfg::SCT + c] = real ? a.shift[(long)n * a.Cin + c] : 0.0f;
  }
  EMO_H_WAIT(0);
  __syncthreads();   // scale / shift tables visible
  EMO_H_QUAD_TABLE()
  if (is_halo_wave) EMO_H_HALO_TABLE()
  EMO_H_TOUCH_QUAD()
  EMO_H_STORE_QUAD(smem, 0)
  EMO_H_STORE_QUAD(smem, 2)
  if (is_halo_wave) {
    EMO_H_TOUCH_HALO()
    EMO_H_STORE_HALO(smem)
  }
  {
    const int st1 = (st_begin + 1) < st_end ? (st_begin + 1) : st_begin;
    EMO_H_SET_STAGE(st1);
  }
  EMO_H_ISSUE_QUAD()
  EMO_H_QUAD_TABLE()
  if (is_halo_wave) {
    EMO_H_ISSUE_HALO()
    EMO_H_HALO_TABLE()
  }
  EMO_H_STAGE_BARRIER()

  // staging pieces of a stage, pinned (sched_barrier) at even distances behind its NSTEPS * NSLOT MFMAs: back-to-back
  // MFMAs stall an in-order wave for the length of the matrix pipe, with the pieces in between its own stream fills it.
  //   piece 0 (behind the first MFMA): vmcnt(0) -- only the quad and halo loads of stage s+1 are outstanding, issued most
  //            of a stage ago -- convert pixels 0, 1 of the quad; THEN start the weight DMA of stage s+1 (issued before the
  //            wait it would be waited for: the waves do not all issue the same number of DMA pieces)
  //   piece 1: convert pixels 2, 3, re-issue the quad loads for stage s+2
  //   piece 2 (halo wave): convert the halo pixel, re-issue its loads
  // In issue order a stage ends with [DMA] [quad] [halo]: vmcnt(8) / vmcnt(16) before the barrier = the DMA has landed.
  constexpr int NPIECE = HALO ? 3 : 2;
  constexpr int PIECE_SPAN = 4;   // pieces sit at slots 0, 1/4, 2/4 of the stage
  static_assert(PIECE_SPAN <= NSTEPS * NSLOT, "one MFMA slot per staging piece");

  for (int st = st_begin; st < st_end; ++st) {
    const int par = (st - st_begin) & 1;
    float* cur = smem + par * BUF;
    float* nxt = smem + (par ^ 1) * BUF;
    const int stn = (st + 1) < st_end ? (st + 1) : st;     // clamped on the last stages: harmless re-stage
    const int stn2 = (st + 2) < st_end ? (st + 2) : stn;
    const halfx8* Ah = reinterpret_cast<const halfx8*>(cur);
    const halfx8* Ph = reinterpret_cast<const halfx8*>(cur + ASZ);
    EMO_H_LOAD_FRAGS(0, 0)
    if (NSTEPS > 1) EMO_H_LOAD_FRAGS(1, 1)
    EMO_H_SET_STAGE(stn2);
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      if (step + 2 < NSTEPS) EMO_H_LOAD_FRAGS((step + 2) % 3, step + 2)
#pragma unroll
      for (int m = 0; m < NSLOT; ++m) {
        // operands swapped: the result tile is [position][channel] (conv_epilogue)
        acc_at(m / TP, m % TP) = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[step % 3][m % TP], fa_[step % 3][m / TP],
                                                                         acc_at(m / TP, m % TP), 0, 0, 0);
        if (EMO_F16_EXPERIMENT != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) {
          if ((pc * NSTEPS * NSLOT) / PIECE_SPAN != step * NSLOT + m) continue;
          if (pc == 0) {
            EMO_H_WAIT(0);
            EMO_H_TOUCH_QUAD()
            if (is_halo_wave) EMO_H_TOUCH_HALO()
            EMO_H_STORE_QUAD(nxt, 0)
            EMO_H_DMA_WEIGHTS(stn, smem_lds + (unsigned)((par ^ 1) * BUF * 4));
          } else if (pc == 1) {
            EMO_H_STORE_QUAD(nxt, 2)
            EMO_H_ISSUE_QUAD()
            EMO_H_QUAD_TABLE()
          } else if (is_halo_wave) {
            EMO_H_STORE_HALO(nxt)
            EMO_H_ISSUE_HALO()
            EMO_H_HALO_TABLE()
          }
        }
        if (EMO_F16_EXPERIMENT != 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    EMO_H_STAGE_BARRIER()
  }
  EMO_H_WAIT(0);   // the re-issued loads of the clamped last stage are dead: drain them before their registers are reused
#undef EMO_H_SET_STAGE
#undef EMO_H_ISSUE_QUAD
#undef EMO_H_ISSUE_HALO
#undef EMO_H_QUAD_TABLE
#undef EMO_H_HALO_TABLE
#undef EMO_H_STORE_QUAD
#undef EMO_H_STORE_HALO
#undef EMO_H_TOUCH_QUAD
#undef EMO_H_TOUCH_HALO
#undef EMO_H_DMA_WEIGHTS
#undef EMO_H_WAIT
#undef EMO_H_BARRIER
#undef EMO_H_STAGE_BARRIER
#undef EMO_H_LOAD_FRAGS
#undef acc_at

  conv_epilogue<TZ, TR, TW, TM, TP, WGP, BM>(a, acc_lo, acc_hi, smem, n, cotile, ptile, ks, x0, y0, z0, m0, p0, wp, half, l32, tid);
}

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
int conv_igemm_f16_launch(ConvArgs a, hipStream_t s) {
  using Cfg = ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (a.Wl % TW || a.Hl % TR || a.Dl % TZ) return EMO_ERR_UNSUPPORTED;
  if (a.Cin % 8) return EMO_ERR_UNSUPPORTED;   // whole 8-channel groups
  if (a.scale && a.Cin > Cfg::SCT) return EMO_ERR_UNSUPPORTED;   // scale / shift tables in LDS
  if ((unsigned long long)a.Cin * a.D * a.H * a.W * 4ull >= (1ull << 32)) return EMO_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  if ((reinterpret_cast<unsigned long long>(a.x) & 15ull) || (a.W & 3)) return EMO_ERR_UNSUPPORTED;     // 16-byte quads
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl / TZ;
  a.n_cchunks = (a.Cin + KC - 1) / KC;
  const long nt = (long)a.tiles_x * a.tiles_y * a.tiles_z;
  if (nt > 0x7fffffffL || a.N > 65535) return EMO_ERR_UNSUPPORTED;
  const int cot = (a.Cout + Cfg::BM - 1) / Cfg::BM;
  const size_t lds = (size_t)Cfg::LDS_BYTES;
  if (lds > 160 * 1024) return EMO_ERR_UNSUPPORTED;
  auto kern = conv_igemm_f16_kernel<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (lds > 64 * 1024) {
    const int rc = emo_raise_dynamic_lds(kern);
    if (rc != EMO_OK) return rc;
  }
  if (a.cot0 < 0 || a.cot0 >= cot) return EMO_ERR_BAD_ARG;
  const int ncot = cot - a.cot0;
  a.n_cotiles = ncot;
  if (a.ksplit < 1 || (a.ksplit > 1 && !a.partial)) return EMO_ERR_BAD_ARG;
  if (a.ksplit == 1) { a.stages_per_split = a.n_cchunks * a.KD; a.partial = nullptr; }
  if (a.ksplit > 1 && a.gn_stats) return EMO_ERR_BAD_ARG;
  if (nt * ncot * a.N * a.ksplit > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nt * ncot * a.N * a.ksplit)), dim3(256), lds, s, a);
  return emo_launch_status();
}
