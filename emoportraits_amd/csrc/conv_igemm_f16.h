// Implicit-GEMM convolution with fp16 MFMA operands and fp32 accumulation for gfx950 -- the reduced-precision mode of
// BASELINE.json configs[4] ("stage_2 refinement ... fp16 MFMA convs").  Opt-in per layer (PackedConv(precision="f16"));
// the exact-fp32 kernel of conv_igemm.h stays the default everywhere and is what bench.py measures.
//
// Round-2 kernel.  Same GEMM view, block order, K split and epilogue (shared: conv_epilogue, incl. the GroupNorm tile
// statistics) as conv_igemm.h; what differs is the operand path:
//   * v_mfma_f32_32x32x16_f16 (the gfx950 2xK form; the round-1 kernel used 32x32x8 = half the rate): the 16 k-values of one
//     MFMA are 16 input CHANNELS at one tap, lane (l&31, l>>5) holds the 8 consecutive channels 8*(l>>5) .. +7 of its row /
//     column, so both operands are one aligned 16-byte LDS read (ds_read_b128, consecutive lanes = consecutive 16 bytes):
//       weights Ah[q][tap][half][BM][8]     packed like that on the host, copied by LDS-DMA
//       patch   Ph[group = 2q + half][patch position][8]
//   * activations stay fp32 NC(D)HW in HBM.  Staging is by (position, 8-channel group) ITEMS, a lane owning the 8 channels
//     of one position: 8 coalesced global dword loads (one per channel plane), the producer's GroupNorm affine + ReLU in
//     fp32, zero padding, saturation to +-65504, round-to-nearest-even to fp16 and ONE 16-byte ds_write -- the round-1
//     kernel staged channel planes per wave and paid one ds_write_b16 per element, which (not the matrix pipe) bounded it at
//     0.16-0.22 of the fp16 peak.  The 8 + 8 scale / shift values of an item are wave-uniform: one vector load per stage,
//     broadcast with v_readlane.
//   * a stage is KC = 16 channels (3x3: 9 MFMA steps of K = 16) or 32 (1x1: 2 steps); tile 64 output channels x 256
//     positions (the fp32 kernel's config D), 2 blocks per CU (LDS: 18 KB weights + 16 KB patch per stage, double-buffered).
//   * same software pipeline as the fp32 kernel: LDS-DMA of the weights and pinned-asm loads of the patch of stage s+1 at
//     the top of stage s, transform + ds_write after 5/8 of the MFMAs, one barrier per stage.
// Needs Cin % 8 == 0 (whole channel groups).  Rounding: operands carry 11 significand bits, products and sums are exact
// fp32 MFMA accumulation.
#pragma once
#include "conv_igemm.h"

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

// identity affine for convolutions without a fused GroupNorm: the staging always applies x * scale + shift, with the
// 32 scale / shift values of a stage loaded through a wave-uniform address -- for scale == NULL that address points here
#define EMO_ONES_8 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f
static __device__ const float emo_identity_scale[32] = {EMO_ONES_8, EMO_ONES_8, EMO_ONES_8, EMO_ONES_8};
static __device__ const float emo_identity_shift[32] = {};
#undef EMO_ONES_8

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
struct ConvCfgH {
  static constexpr int BM = WGM * TM * 32;
  static constexpr int BP = WGP * TP * 32;
  static constexpr int TAPS = KH * KW;
  static constexpr int PR = TR + KH - 1;
  static constexpr int PW = TW + KW - 1;
  static constexpr int CHS = TZ * PR * PW;               // patch positions
  static constexpr int KQ = KC / 16;                     // MFMA k-steps (16 channels each) per tap and stage
  static constexpr int NG = KC / 8;                      // 8-channel groups per stage: group = 2 q + half
  static constexpr int CPG = (CHS + 63) / 64;            // 64-position chunks per group
  static constexpr int PPW = (CPG + 3) / 4;              // position chunks per wave: chunks w, w + 4, ...
  static constexpr int IPW = PPW * NG;                   // items (64 positions x 8 channels) per wave and stage
  static constexpr int NSTEPS = KQ * TAPS;
  static constexpr int ASZ_H = KC * TAPS * BM;           // halfs of one stage's weight tile
  static constexpr int PSZ_H = NG * CHS * 8;             // halfs of one stage's patch
  static constexpr int ASZ = ASZ_H / 2;                  // in floats
  static constexpr int BUF = ASZ + ((PSZ_H / 2 + 3) & ~3);
  static constexpr int SCT = 1024;                       // entries of the per-sample scale / shift tables kept in LDS
  static constexpr int LDS_BYTES = (2 * BUF + 256 + 2 * SCT) * 4;   // two stage buffers + 64 dump slots + scale / shift tables
  static constexpr int NDMA_MIN = (ASZ_H * 2) / 4096;    // LDS-DMA instructions EVERY wave issues per stage (some issue one more)
  static constexpr int BY_LDS = (160 * 1024) / LDS_BYTES;
  // 2 blocks per CU at most: 64 accumulator + 48-64 in-flight patch + 32-64 scale / shift registers per lane do not fit the
  // 168-VGPR budget of 3 waves per SIMD (at 3 the compiler spilled in-flight load destinations: tools/kernel_resources.py --audit)
  static constexpr int OCC = BY_LDS < 1 ? 1 : (BY_LDS > 2 ? 2 : BY_LDS);
  static_assert(WGM * WGP == 4, "4 waves per block");
  static_assert(TZ * TR * TW == BP, "position tile must equal BP");
  static_assert(KC % 16 == 0 && KC <= 32, "whole 16-channel MFMA steps; the identity-affine tables hold 32 entries");
  static_assert((ASZ_H * 2) % 16 == 0, "weight tile must be 16-byte copyable");
  static_assert(TM * TP <= 4, "accumulator budget");
  static_assert(2 * BUF >= 2 * WGP * BM, "the GroupNorm tile statistics are exchanged through the stage buffers");
};

#ifndef EMO_F16_ROLLING
#define EMO_F16_ROLLING 1   /* 1: rolling prefetch (below); 0: the patch of stage s+1 is loaded at the top of stage s */
#endif

// LDS-DMA hidden from the compiler (asm): 16 bytes per lane from `gsrc` to the wave-uniform LDS byte address `lds_dst`
// + lane * 16.  M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void emo_dma16_pinned(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>::OCC,
                                   ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>::OCC)))
void conv_igemm_f16_kernel(const ConvArgs a) {
  using Cfg = ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  constexpr int BM = Cfg::BM, TAPS = Cfg::TAPS, PR = Cfg::PR, PW = Cfg::PW, CHS = Cfg::CHS;
  constexpr int ASZ = Cfg::ASZ, ASZ_H = Cfg::ASZ_H, BUF = Cfg::BUF, CPG = Cfg::CPG, PPW = Cfg::PPW, NG = Cfg::NG;
  constexpr int NSTEPS = Cfg::NSTEPS;

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
  const int cotile = L % a.n_cotiles;
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

  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const float* xn = a.x + (long)n * a.Cin * DHW;
  const bool has_affine = a.scale != nullptr;
  const int padD = a.KD >> 1;
  const float* scale_n = has_affine ? a.scale + (long)n * a.Cin : emo_identity_scale;
  const float* shift_n = has_affine ? a.shift + (long)n * a.Cin : emo_identity_shift;
  const float clamp_lo = a.relu_in ? 0.0f : -65504.0f;   // lower bound of the staged value: ReLU, or the fp16 range

  // ---- staging map: wave w stages the 64-position chunks w, w + 4, ... of the patch, for every 8-channel group of the
  //      stage (item = (chunk, group); the group index is a compile-time constant of the unrolled loops, so the
  //      per-channel scale / shift registers are indexed statically).  The position part is the same in every stage:
  //      plane offset and validity once per thread. ----
  unsigned p_off[PPW];
  int p_pz[PPW];
  bool p_ok[PPW];
  int p_e[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int chunk = wave + 4 * k;
    const int e = chunk * 64 + lane;
    const int pz = e / (PR * PW);
    const int rem2 = e - pz * (PR * PW);
    const int pr = rem2 / PW;
    const int pc = rem2 - pr * PW;
    const int yl = y0 + pr - (KH >> 1);
    const int xl = x0 + pc - (KW >> 1);
    const bool ok = (e < CHS) && ((unsigned)yl < (unsigned)a.Hl) && ((unsigned)xl < (unsigned)a.Wl);
    const int ys = UPS ? (yl >> 1) : yl;
    const int xs = UPS ? (xl >> 1) : xl;
    p_ok[k] = ok;
    p_off[k] = ok ? (unsigned)(ys * a.W + xs) * 4u : 0u;
    p_pz[k] = pz;
    p_e[k] = e;
  }

  const int nstages_all = a.n_cchunks * a.KD;
  const int st_begin = ks * a.stages_per_split;
  const int st_end = min(nstages_all, st_begin + a.stages_per_split);
  const char* wsrc = reinterpret_cast<const char*>(a.wpk) + ((long)cotile * nstages_all) * (ASZ_H * 2);

  halfx8* const dump8 = reinterpret_cast<halfx8*>(smem + 2 * BUF) + lane;   // per-lane dump slot (written, never read)
  constexpr int IPW = PPW * NG;   // staging items (position chunk, 8-channel group) per wave and stage

  constexpr int TPH = TP > 2 ? TP / 2 : TP;
  floatx16 acc_lo[TM][TPH], acc_hi[TM][TPH];
#define acc_at(i_, j_) ((j_) < TPH ? acc_lo[i_][(j_) < TPH ? (j_) : 0] : acc_hi[i_][(j_) >= TPH ? (j_) - TPH : 0])
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TPH; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_lo[i][j][r] = 0.0f; acc_hi[i][j][r] = 0.0f; }

  // lane bases into the LDS tiles, in units of 8 halfs (16 bytes)
  const int a_base = half * BM + m0 + l32;
  int b_base[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW;
    const int row = (p / TW) % TR;
    const int pz = p / (TW * TR);
    b_base[j] = half * CHS + pz * (PR * PW) + row * PW + col;
  }

  // fragments of an MFMA step (TM weight and TP patch fragments, 16 bytes per lane each) in three rotating register sets:
  // the reads of step s + 2 are issued before the MFMAs of step s, so their LDS latency hides behind a whole step of MFMAs
  // even in the steps that carry no staging item
  halfx8 fa_[3][TM], fb_[3][TP];
#define EMO_H_LOAD_FRAGS(set_, step_)                                                                 \
  {                                                                                                   \
    const int q = (step_) / TAPS, tap = (step_) % TAPS;                                               \
    const int r = tap / KW, s = tap % KW;                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa_[set_][i] = Ah[a_base + ((q * TAPS + tap) * 2) * BM + i * 32]; \
    _Pragma("unroll") for (int j = 0; j < TP; ++j) fb_[set_][j] = Ph[b_base[j] + (q * 2) * CHS + r * PW + s]; \
  }
#define EMO_H_MFMAS(set_)                                                                             \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
      _Pragma("unroll") for (int j = 0; j < TP; ++j)                                                  \
        acc_at(i, j) = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_[set_][i], fb_[set_][j], acc_at(i, j), 0, 0, 0); \
  }
#define EMO_H_MFMA_STEP(step_)                                                                        \
  {                                                                                                   \
    EMO_H_LOAD_FRAGS(0, step_)                                                                        \
    EMO_H_MFMAS(0)                                                                                    \
  }

#if EMO_F16_ROLLING
  // ---- rolling prefetch.  Every VMEM instruction of the loop is inline asm (patch loads AND the weight LDS-DMA), so the
  //      compiler inserts no vmcnt waits of its own and the counts below are exact.  Per wave and stage s, in issue order:
  //        [items i+1 .. IPW-1 of stage s+1]  [DMA of stage s+1]  [items 0 .. i-1 of stage s+2]
  //      are newer than the loads of item i of stage s+1 when that item is converted during stage s: item i waits with
  //      vmcnt((IPW-1)*8 + NDMA_MIN) and is re-issued for stage s+2 right after its conversion, so every patch load has
  //      a whole stage of MFMA work to land (the non-rolling schedule gave it a third of a stage).  At the end of the stage
  //      vmcnt(IPW*8) leaves exactly the re-issued items outstanding: the DMA of stage s+1 has landed before the barrier.
  //      The per-sample scale / shift vectors live in LDS (two SCT-entry tables, identity when the layer has no affine).
  static_assert((IPW - 1) * 8 + Cfg::NDMA_MIN <= 63 && IPW * 8 <= 63, "vmcnt is a 6-bit counter");
  float* const sct = smem + 2 * BUF + 256;
  for (int c = tid; c < Cfg::SCT; c += 256) {
    const bool real = has_affine && c < a.Cin;
    sct[c] = real ? a.scale[(long)n * a.Cin + c] : 1.0f;
    sct[Cfg::SCT + c] = real ? a.shift[(long)n * a.Cin + c] : 0.0f;
  }
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);

  float pv[IPW][8];   // raw patch values of an item (pinned asm loads, in flight across the stage boundary)
  bool pzi[IPW];      // the item's depth slice lies inside the volume (per lane)
  bool gvi[IPW];      // wave-uniform: the item's channel group exists (and, for TZ == 1, the depth slice is inside)
  int tix[IPW];       // wave-uniform: first channel of the item in the scale / shift tables
  floatx4 sc_[2][2], sh_[2][2];   // scale / shift of the next two items (set = item & 1), read two items ahead of their use
  static_assert(IPW % 2 == 0, "two table register sets alternate over the items");

  int n_ci0, n_zu;    // stage being loaded: first input channel, first depth slice
  bool n_zv;
#define EMO_H_SET_STAGE(stage_)                                                                       \
  {                                                                                                   \
    const int cc_ = (stage_) / a.KD;                                                                  \
    n_ci0 = cc_ * KC;                                                                                 \
    n_zu = z0 + ((stage_) - cc_ * a.KD) - padD;                                                       \
    n_zv = (unsigned)n_zu < (unsigned)a.D;                                                            \
  }
#define EMO_H_ISSUE_ITEM(it_)                                                                         \
  {                                                                                                   \
    const int k_ = (it_) / NG, g_ = (it_) % NG;                                                       \
    unsigned off_ = p_off[k_];                                                                        \
    bool zok_ = true;                                                                                 \
    if (TZ > 1) {                                                                                     \
      const int zi = n_zu + p_pz[k_];                                                                 \
      zok_ = (unsigned)zi < (unsigned)a.D;                                                            \
      off_ += (unsigned)((zok_ ? zi : 0) * HW) * 4u;                                                  \
    }                                                                                                 \
    pzi[it_] = zok_;                                                                                  \
    const int c0_ = n_ci0 + g_ * 8;                                                                   \
    const bool cv_ = c0_ < a.Cin;                                                                     \
    const int cs_ = cv_ ? c0_ : 0;                                                                    \
    gvi[it_] = cv_ && (TZ > 1 || n_zv);                                                               \
    tix[it_] = has_affine ? cs_ : (cs_ & (Cfg::SCT - 1));                                             \
    const float* base_ = xn + (long)cs_ * DHW + (long)((TZ == 1 && n_zv) ? n_zu : 0) * HW;            \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) pv[it_][u] = emo_gload_pinned(base_ + (long)u * DHW, off_); \
  }
// transform in fp32 (affine of the producer's GroupNorm, ReLU + saturation in one v_med3, zero padding), round to fp16,
// ONE 16-byte ds_write per item: Ph[(g * CHS + e) * 8 .. + 7]
#define EMO_H_STORE_ITEM(buf_, it_)                                                                   \
  {                                                                                                   \
    const int k_ = (it_) / NG, g_ = (it_) % NG;                                                       \
    halfx8* Ph_ = reinterpret_cast<halfx8*>((buf_) + ASZ);                                            \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch(pv[it_][u]);                              \
    const bool keep_ = p_ok[k_] && gvi[it_] && pzi[it_];                                              \
    halfx8 h_;                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                   \
      float v = __fmaf_rn(pv[it_][u], sc_[(it_) & 1][u / 4][u % 4], sh_[(it_) & 1][u / 4][u % 4]);      \
      v = keep_ ? v : 0.0f;                           /* zero padding applies to the transformed tensor */ \
      v = __builtin_amdgcn_fmed3f(v, clamp_lo, 65504.0f);   /* ReLU (or -65504) and saturation instead of inf */ \
      h_[u] = (_Float16)v;                                                                            \
    }                                                                                                 \
    halfx8* d_ = (p_e[k_] < CHS) ? Ph_ + (g_ * CHS + p_e[k_]) : dump8;                                \
    *d_ = h_;                                                                                         \
  }
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
// scale / shift of the 8 channels of an item, read from the LDS tables one item ahead of their use
#define EMO_H_LOAD_TABLE(it_)                                                                         \
  {                                                                                                   \
    const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + (tix[it_] >> 2);                     \
    sc_[(it_) & 1][0] = t4_[0]; sc_[(it_) & 1][1] = t4_[1];                                           \
    sh_[(it_) & 1][0] = t4_[Cfg::SCT / 4]; sh_[(it_) & 1][1] = t4_[Cfg::SCT / 4 + 1];                 \
  }
#define EMO_H_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define EMO_H_BARRIER(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")

  __syncthreads();   // scale / shift tables visible (no asm VMEM issued yet: the compiler's own waits are complete here)

  // ---- prologue: stage st_begin into buffer 0, issue the items of stage st_begin + 1 ----
  EMO_H_DMA_WEIGHTS(st_begin, smem_lds);
  EMO_H_SET_STAGE(st_begin);
#pragma unroll
  for (int it = 0; it < IPW; ++it) EMO_H_ISSUE_ITEM(it)
  EMO_H_WAIT(0);
  {
    const int st1 = (st_begin + 1) < st_end ? (st_begin + 1) : st_begin;
    EMO_H_SET_STAGE(st1);
  }
#pragma unroll
  for (int it = 0; it < IPW; ++it) {
    EMO_H_LOAD_TABLE(it)
    EMO_H_STORE_ITEM(smem, it)
    EMO_H_ISSUE_ITEM(it)
  }
  EMO_H_LOAD_TABLE(0)
  EMO_H_LOAD_TABLE(1)
  EMO_H_BARRIER(IPW * 8);

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
    EMO_H_DMA_WEIGHTS(stn, smem_lds + (unsigned)((par ^ 1) * BUF * 4));
    EMO_H_SET_STAGE(stn2);
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
#pragma unroll
      for (int it = 0; it < IPW; ++it)
        if ((it * NSTEPS) / IPW == step) {
          EMO_H_WAIT((IPW - 1) * 8 + Cfg::NDMA_MIN);
          EMO_H_STORE_ITEM(nxt, it)
          EMO_H_ISSUE_ITEM(it)
          EMO_H_LOAD_TABLE((it + 2) % IPW)   // the last two items prefetch for items 0, 1 of the next stage (tix already set)
        }
      if (step + 2 < NSTEPS) EMO_H_LOAD_FRAGS((step + 2) % 3, step + 2)
      EMO_H_MFMAS(step % 3)
    }
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    EMO_H_BARRIER(IPW * 8);
  }
  EMO_H_WAIT(0);   // the re-issued items of the clamped last stage are dead: drain them before their registers are reused
#undef EMO_H_SET_STAGE
#undef EMO_H_ISSUE_ITEM
#undef EMO_H_STORE_ITEM
#undef EMO_H_DMA_WEIGHTS
#undef EMO_H_LOAD_TABLE
#undef EMO_H_WAIT
#undef EMO_H_BARRIER
#else

  float pv[PPW][NG][8];   // raw patch values of the next stage (pinned asm loads)
  bool pvz[PPW];          // depth slice of the chunk inside the volume
  bool gv[NG];            // wave-uniform: the channel group exists (and, for TZ == 1, the depth slice is inside the volume)
  floatx4 sc4[KC / 4], sh4[KC / 4];   // scale / shift of the KC channels of the next stage, the same value in every lane
                                      // (loaded with a wave-uniform address: no broadcast instructions)

#define EMO_H_ISSUE_PATCH(stage_)                                                                     \
  {                                                                                                   \
    const int cc_ = (stage_) / a.KD;                                                                  \
    const int t_ = (stage_) - cc_ * a.KD;                                                             \
    const int ci0_ = cc_ * KC;                                                                        \
    _Pragma("unroll") for (int j = 0; j < KC / 4; ++j) {                                              \
      /* Cin % 8 == 0: whole quads exist or not; without an affine the 32-entry identity tables are indexed by j alone */ \
      const int c4_ = !has_affine ? 4 * j : ((ci0_ + 4 * j) < a.Cin ? (ci0_ + 4 * j) : 0);            \
      sc4[j] = emo_gload4_pinned(scale_n + c4_, 0u);                                                  \
      sh4[j] = emo_gload4_pinned(shift_n + c4_, 0u);                                                  \
    }                                                                                                 \
    const int zu_ = z0 + t_ - padD;                                                                   \
    const bool zv_ = (unsigned)zu_ < (unsigned)a.D;                                                   \
    _Pragma("unroll") for (int k = 0; k < PPW; ++k) {                                                 \
      unsigned off_ = p_off[k];                                                                       \
      if (TZ == 1) {                                                                                  \
        pvz[k] = true;                                                                                \
      } else {                                                                                        \
        const int zi = zu_ + p_pz[k];                                                                 \
        const bool zok = (unsigned)zi < (unsigned)a.D;                                                \
        pvz[k] = zok;                                                                                 \
        off_ += (unsigned)((zok ? zi : 0) * HW) * 4u;                                                 \
      }                                                                                               \
      _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                \
        const int c0_ = ci0_ + g * 8;                                                                 \
        const bool cv_ = c0_ < a.Cin;                                                                 \
        const int cs_ = cv_ ? c0_ : 0;                                                                \
        gv[g] = cv_ && (TZ > 1 || zv_);                                                               \
        const float* base_ = xn + (long)cs_ * DHW + (long)((TZ == 1 && zv_) ? zu_ : 0) * HW;          \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) pv[k][g][u] = emo_gload_pinned(base_ + (long)u * DHW, off_); \
      }                                                                                               \
    }                                                                                                 \
  }

#define EMO_H_WAIT_PATCH()                                                                            \
  {                                                                                                   \
    emo_wait_vmem0();                                                                                 \
    _Pragma("unroll") for (int j = 0; j < KC / 4; ++j) {                                              \
      emo_touch4(sc4[j]); emo_touch4(sh4[j]);                                                         \
    }                                                                                                 \
    _Pragma("unroll") for (int k = 0; k < PPW; ++k)                                                   \
      _Pragma("unroll") for (int g = 0; g < NG; ++g)                                                  \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch(pv[k][g][u]);                         \
  }

// weight tile of one stage by LDS-DMA (1 KiB per wave-instruction), lane-linear = the packed order
#define EMO_H_ISSUE_WEIGHTS(stage_, dst_)                                                             \
  {                                                                                                   \
    const char* ws_ = wsrc + (long)(stage_) * (ASZ_H * 2);                                            \
    constexpr int NGL = (ASZ_H * 2 + 4095) / 4096;                                                    \
    _Pragma("unroll") for (int i = 0; i < NGL; ++i) {                                                 \
      const int j = wave + 4 * i;                                                                     \
      const int boff = j * 1024 + lane * 16;                                                          \
      if (boff < ASZ_H * 2)                                                                           \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws_ + boff), \
                                         (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(dst_) + j * 1024), 16, 0, 0); \
    }                                                                                                 \
  }

// transform in fp32 (affine of the producer's GroupNorm, ReLU + saturation in one v_med3, zero padding), round to fp16,
// ONE 16-byte ds_write per item: Ph[(g * CHS + e) * 8 .. + 7].  One item (position chunk k_, channel group g_) at a time:
// the items of a stage are spread over its MFMA steps, ~28 VALU instructions behind the 4 MFMAs of a step.
#define EMO_H_STORE_ITEM(buf_, k_, g_)                                                                \
  {                                                                                                   \
    halfx8* Ph_ = reinterpret_cast<halfx8*>((buf_) + ASZ);                                            \
    const bool keep_ = p_ok[k_] && gv[g_] && pvz[k_];                                                 \
    halfx8 h_;                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                   \
      constexpr_int_c_((g_) * 8 + u)                                                                  \
      float v = __fmaf_rn(pv[k_][g_][u], sc4[c_ / 4][c_ % 4], sh4[c_ / 4][c_ % 4]);                   \
      v = keep_ ? v : 0.0f;                           /* zero padding applies to the transformed tensor */ \
      v = __builtin_amdgcn_fmed3f(v, clamp_lo, 65504.0f);   /* ReLU (or -65504) and saturation instead of inf */ \
      h_[u] = (_Float16)v;                                                                            \
    }                                                                                                 \
    halfx8* d_ = (p_e[k_] < CHS) ? Ph_ + ((g_) * CHS + p_e[k_]) : dump8;                              \
    *d_ = h_;                                                                                         \
  }
#define EMO_H_STORE_PATCH(buf_)                                                                       \
  {                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < PPW; ++k)                                                   \
      _Pragma("unroll") for (int g = 0; g < NG; ++g) EMO_H_STORE_ITEM(buf_, k, g)                     \
  }
#define constexpr_int_c_(expr_) const int c_ = (expr_);

  // item i of the next stage is transformed and stored after MFMA step STORE_STEP0 + i * (NSTEPS - STORE_STEP0) / IPW
  constexpr int STORE_STEP0 = NSTEPS >= 3 ? NSTEPS / 3 : (NSTEPS > 1 ? 1 : 0);

  // ---- prologue: stage st_begin into buffer 0 ----
  EMO_H_ISSUE_WEIGHTS(st_begin, smem);
  EMO_H_ISSUE_PATCH(st_begin);
  EMO_H_WAIT_PATCH();
  EMO_H_STORE_PATCH(smem);
  __syncthreads();

  for (int st = st_begin; st < st_end; ++st) {
    float* cur = smem + ((st - st_begin) & 1) * BUF;
    float* nxt = smem + ((st - st_begin + 1) & 1) * BUF;
    const int stn = (st + 1) < st_end ? (st + 1) : st;   // clamped prefetch on the last stage: harmless re-stage
    const halfx8* Ah = reinterpret_cast<const halfx8*>(cur);
    const halfx8* Ph = reinterpret_cast<const halfx8*>(cur + ASZ);
    EMO_H_ISSUE_WEIGHTS(stn, nxt);
    EMO_H_ISSUE_PATCH(stn);
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      if (step == STORE_STEP0) { EMO_H_WAIT_PATCH(); }
#pragma unroll
      for (int it = 0; it < IPW; ++it)
        if (STORE_STEP0 + (it * (NSTEPS - STORE_STEP0)) / IPW == step) { EMO_H_STORE_ITEM(nxt, it / NG, it % NG) }
      EMO_H_MFMA_STEP(step)
    }
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
#undef EMO_H_ISSUE_PATCH
#undef EMO_H_WAIT_PATCH
#undef EMO_H_ISSUE_WEIGHTS
#undef EMO_H_STORE_PATCH
#undef EMO_H_STORE_ITEM
#undef constexpr_int_c_

#endif
#undef EMO_H_MFMA_STEP
#undef EMO_H_MFMAS
#undef EMO_H_LOAD_FRAGS
#undef acc_at

  conv_epilogue<TZ, TR, TW, TM, TP, WGP, BM>(a, acc_lo, acc_hi, smem, n, cotile, ptile, ks, x0, y0, z0, m0, p0, wp, half, l32, tid);
}

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
int conv_igemm_f16_launch(ConvArgs a, hipStream_t s) {
  using Cfg = ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (a.Wl % TW || a.Hl % TR || a.Dl % TZ) return EMO_ERR_UNSUPPORTED;
  if (a.Cin % 8) return EMO_ERR_UNSUPPORTED;   // whole 8-channel groups
  if (EMO_F16_ROLLING && a.scale && a.Cin > Cfg::SCT) return EMO_ERR_UNSUPPORTED;   // scale / shift tables in LDS
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
  a.n_cotiles = cot;
  if (a.ksplit < 1 || (a.ksplit > 1 && !a.partial)) return EMO_ERR_BAD_ARG;
  if (a.ksplit == 1) { a.stages_per_split = a.n_cchunks * a.KD; a.partial = nullptr; }
  if (a.ksplit > 1 && a.gn_stats) return EMO_ERR_BAD_ARG;
  if (nt * cot * a.N * a.ksplit > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nt * cot * a.N * a.ksplit)), dim3(256), lds, s, a);
  return emo_launch_status();
}
