// The fp16 split of conv_igemm_bf16x3.h (SPLIT = 2: fp32 3x3 convolution from two fp16 terms of the scaled operands, three
// products, fp32 accumulation, device-checked operand range) with TWO 64-channel output tiles per work item -- "CT2".
//
// Why.  The 64 x 256 item of conv_igemm_bf16x3.h loads, transforms (GroupNorm affine + ReLU + zero padding), range-checks and
// splits the fp32 patch of its position tile once per 64-channel tile of the output: Cout / 64 times per layer (measured: 1.5x
// the algorithmic HBM bytes, profiles/r4_pmc_conv_f16x2_traffic.json), and that staging -- 8 quad loads + 8 conversion units per
// stage and thread, beside 9 weight pieces -- is most of what keeps its K loop at 43-46 cycles per 32-cycle MFMA slot.  A block
// has one wave per SIMD and 512 registers per lane, of which the accumulators of one channel tile take 128: there is room for a
// second set.  Here one converted patch in LDS feeds both channel tiles: a "stage" (16 input channels [x depth tap]) is two
// HALF-STAGES h = 0 / 1 that run the nine tap steps of channel tile c0 + h on the SAME patch buffer with their own kernel rows
// and their own accumulators.  Per MFMA: half the patch loads, half the conversion and range-check VALU, half the patch LDS
// stores, 13 instead of 17 vector-memory instructions per 108 MFMAs and wave; the weight traffic, the fragment reads and the
// barrier per nine steps are unchanged.
//
// Schedule (one barrier per half-stage, everything in flight is drained there: vmcnt(0), nothing to count).  W[0] / W[1] hold the
// kernel rows of half-stage 0 / 1, P[pp] the patch of stage cg (pp toggles per stage), ONE set of raw patch registers qv:
//   (cg, h = 0)  steps 0 .. 7  the patch of stage cg + 1 is converted from qv into P[pp ^ 1], half a pixel per step
//                steps 0 .. 4  pieces 2 .. 8 of the kernel rows of (cg, 1) into W[1]
//                step 8        barrier; pieces 0, 1 of the rows of (cg + 1, 0) into W[0]; fragments of (cg, 1) step 0
//   (cg, h = 1)  steps 0 .. 3  the quad loads of stage cg + 2 into qv (free since (cg, 0) step 7), two per step
//                steps 0 .. 4  pieces 2 .. 8 of the rows of (cg + 1, 0) into W[0]
//                step 8        barrier (the quad loads have landed; every wave has stored its share of P[pp ^ 1]); pieces 0, 1
//                              of the rows of (cg + 1, 1) into W[1]; fragments of (cg + 1, 0) step 0 from P[pp ^ 1]
// Items of a persistent block are chained as in conv_igemm_bf16x3.h: past the last stage the sequence (cg + 1, cg + 2) continues
// with the next item's stages 0, 1 (same sample), so the next item starts behind the epilogue with bias entries, two weight
// pieces and one barrier.  The epilogue is conv_igemm_bf16x3.h's, run once per channel tile on its accumulator set; it
// transposes through W[1] (the last half-stage's rows; the dead pieces 0, 1 of that step go to the idle patch buffer).
// Launch: conv_f16x2_ct2_launch() takes the channel-tile PAIRS of a layer when there are enough of them to fill the chip; an
// odd last tile (Cout = 320: 2 pairs + 1) runs conv_igemm_bf16x3_kernel<SPLIT = 2> with ConvArgs::cot0 set.  No K split.
// Same arithmetic per output element as the single-tile kernel (same products in the same order into the same two accumulator
// sets): the two are BIT-IDENTICAL (tests/test_conv_bf16x3_gpu.py).
#pragma once
#include "conv_igemm_bf16x3.h"
#include "conv_split_pair_common.h"

#ifndef EMO_CT2_RES_EARLY
#define EMO_CT2_RES_EARLY 0   /* 1: the first tile's residual loads go out in front of the K loop instead of at the top of the epilogue.
                                 Measured (tools/session/r5_call9.sh): epilogue 13.1 k -> 12.6 k cycles per pair, but the K loop's first
                                 barrier (vmcnt(0): loads complete in order) then waits for them -- K loop + 3.8 k, prologue + 2.2 k: off */
#endif

template <int TR, int TW, bool UPS>
struct ConvCfgS2 : ConvCfgS<TR, TW, UPS, 2> {
  using Base = ConvCfgS<TR, TW, UPS, 2>;
  static_assert(Base::NWB == 2 && Base::EPI_IN_W, "two whole weight stages in LDS; the epilogue transposes through one of them");
  // second channel tile: its bias table and its (mean, M2) exchange, behind the first tile's
  static constexpr int OFF_BIAS2_F = Base::OFF_EPI_F;
  static constexpr int OFF_STAT2_F = OFF_BIAS2_F + Base::BM;
  static constexpr int LDS_BYTES = (OFF_STAT2_F + 2 * Base::WGP * Base::BM) * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(Base::PBUF * 16 >= 8 * 1024, "the dead weight pieces of an item's last step are dumped into a patch buffer");
};

template <int TR, int TW, bool UPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_igemm_bf16x3_ct2_kernel(const ConvArgs a) {
  using Cfg = ConvCfgS2<TR, TW, UPS>;
  using opx8 = halfx8;
  constexpr int SPLIT = 2, NPL = 2, NPROD = 3;
  constexpr int BM = Cfg::BM, TM = Cfg::TM, TP = Cfg::TP, WGP = Cfg::WGP, KC = Cfg::KC;
  constexpr int PR = Cfg::PR, NQ = Cfg::NQ, NQ1 = Cfg::NQ1, SUB = Cfg::SUB, CHS = Cfg::CHS, QPG = Cfg::QPG;
  constexpr int NHQ = Cfg::NHQ, TWS = Cfg::TWS, WPLANE = Cfg::WPLANE, WROW = Cfg::WROW, PPL = Cfg::PPL, PBUF = Cfg::PBUF;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int wp = wave;
  const int p0 = wp * TP * 32;
  float sat_m = 0.0f;                                    // largest |scaled staged value| this thread has seen

  // ---- constants of the launch and of the thread (conv_igemm_bf16x3.h) ----
  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const bool has_affine = a.scale != nullptr;
  // epilogue form: the straight-line one without / with a same-size / with a half-size residual (conv_ct2_epilogue_form() on the
  // host: a launch whose output needs the general epilogue -- activation, ragged channel tile, unaligned tensors -- never gets here;
  // with the general form inlined twice the register allocator, which has no free accumulation registers to park values in,
  // spilled 250 values to scratch around every item)
  const int epi_mode = __builtin_amdgcn_readfirstlane(a.res == nullptr ? 0 : (a.res_ups ? 2 : 1));
  const float in_scale = a.in_scale;
  const int padD = a.KD >> 1;
  constexpr float CLAMP_HI = 65504.0f;
  const float clamp_lo = a.relu_in ? 0.0f : -CLAMP_HI;
  const int nst = a.n_cchunks * a.KD;                    // stages of an item (no K split)
  const int nptiles = a.tiles_x * a.tiles_y * a.tiles_z;

  // ---- staging map (conv_igemm_bf16x3.h: interior quads and, on the lanes that own none, one halo pixel each) ----
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

  floatx16 acc_lo[2][TM][TP], acc_hi[2][TM][TP];         // [channel tile of the pair]: all 256 accumulation registers

  // ---- work items: (sample, position tile, channel-tile PAIR), XCD-contiguous, pair fastest; persistent blocks ----
  const int q8 = a.n_work >> 3, r8 = a.n_work & 7;
  const int xcd = blockIdx.x & 7;
  const int n_mine = q8 + (xcd < r8 ? 1 : 0);
  const int l_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int l_stride = (gridDim.x + 7) >> 3;
  int it_cotile = 0, it_n = 0, it_ptile = 0, it_x0 = 0, it_y0 = 0, it_z0 = 0;
  unsigned lq_off = 0;
  bool lq_ok = false;
  int lq_z0 = 0;
// byte address of the packed kernel rows of (channel tile c_, stage k_)
#define EMO_T_WPTR(c_, k_) (reinterpret_cast<const char*>(a.wpk) + (long)((c_) * nst + (k_)) * (3 * Cfg::WROW_BYTES))

  // LDS byte offsets of the lane's operands (conv_igemm_bf16x3.h); the patch buffer of a stage is a RUN-TIME parity here (two
  // half-stages per stage make the unrolled pair of loop bodies the two channel tiles, not two stages): + pcur_b / pnxt_b
  const int a_off = (half * BM + l32) * 16;
  EMO_P_DECLARE_B_OFF()

  const char* const lds_c = reinterpret_cast<const char*>(smem);
  char* const lds_w = reinterpret_cast<char*>(smem);
  opx8 fa_[2][NPL][TM], fb_[2][NPL][TP];     // [register set: this step / the next][plane][tile]
// (wbase_: slots, compile-time; pbyte_: byte offset of the patch buffer, run-time)
#define EMO_T_LOAD_FRAGS_PLANE(set_, pl_, wbase_, pbyte_, r_, s_)                                      \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
      fa_[set_][pl_][i] = *reinterpret_cast<const opx8*>(lds_c + a_off + ((wbase_) + (pl_) * WPLANE + (s_) * 2 * BM + i * 32) * 16); \
    _Pragma("unroll") for (int j = 0; j < TP; ++j)                                                    \
      fb_[set_][pl_][j] = *reinterpret_cast<const opx8*>(lds_c + (EMO_P_B_OFF(j, r_, s_) + (pbyte_)) + ((pl_) * PPL) * 16); \
  }

  float* const sct = smem + Cfg::OFF_SCT * 4;
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);
  const unsigned lane16 = (unsigned)lane * 16u;

  // raw patch registers: ONE buffer -- converted during half-stage 0, reloaded during half-stage 1
  floatx4 qv[8];
  float q_lo, q_hi;
  int q_tix;
  floatx4 q_sc, q_sh;
  opx8 cv_h, cv_m;
  emo_intx4 xrs = emo_raw_buffer(a.x);
  unsigned usoff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) usoff[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)u * (unsigned)DHW * 4u * (EMO_CT2_X_CONST ? 0u : 1u)));

  int n_ci0, n_zu;
  bool n_zv;
  int ld_stage, ld_cc, ld_kd;   // the stage whose patch is being loaded, stepped (no division in the loop)
#define EMO_T_SET_STAGE_VARS()                                                                        \
  {                                                                                                   \
    n_ci0 = ld_cc * KC;                                                                               \
    n_zu = lq_z0 + ld_kd - padD;                                                                      \
    n_zv = (unsigned)n_zu < (unsigned)a.D;                                                            \
  }
  unsigned q_vo;
#define EMO_T_ISSUE_BEGIN()                                                                           \
  {                                                                                                   \
    const int c0_ = n_ci0 + q_g * 8;                                                                  \
    const bool cv_ = c0_ < a.Cin;                                                                     \
    const int cs_ = cv_ ? c0_ : 0;                                                                    \
    const bool keep_ = lq_ok && cv_ && n_zv;                                                          \
    q_lo = keep_ ? clamp_lo : 0.0f;                                                                   \
    q_hi = keep_ ? CLAMP_HI : 0.0f;                                                                   \
    q_vo = lq_off + (EMO_CT2_X_CONST ? 0u : ((unsigned)cs_ * (unsigned)DHW + (unsigned)((n_zv ? n_zu : 0) * HW)) * 4u);   \
    q_tix = (has_affine ? cs_ : (cs_ & (Cfg::SCT - 1))) >> 2;                                         \
  }
#define EMO_T_ISSUE_LOADS(u0_, u1_)                                                                   \
  { _Pragma("unroll") for (int u = (u0_); u < (u1_); u += 2) emo_bload4x2_pinned(xrs, q_vo, usoff[u], usoff[u + 1], qv[u], qv[u + 1]); }
#define EMO_T_HALF_TABLE(hf_)                                                                         \
  {                                                                                                   \
    const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + q_tix + (hf_);                       \
    q_sc = t4_[0]; q_sh = t4_[Cfg::SCT / 4];                                                          \
  }
#define EMO_T_TOUCH_QUAD() { _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch4(qv[u]); }
// conversion of channels 4 * hf_ .. + 3 of pixel i_ (conv_igemm_bf16x3.h, SPLIT = 2); pbyte_: byte offset of the target patch buffer
#define EMO_T_CONV_HALF(pbyte_, i_, hf_)                                                              \
  {                                                                                                   \
    float t_[4];                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                     \
      t_[k] = __fmaf_rn(qv[4 * (hf_) + k][i_], q_sc[k], q_sh[k]);                                     \
    sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[0])), __builtin_fabsf(t_[1])); \
    sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[2])), __builtin_fabsf(t_[3])); \
    _Pragma("unroll") for (int k = 0; k < 4; k += 2)                                                  \
      emo_split_f16x2_pair(__builtin_amdgcn_fmed3f(t_[k], q_lo, q_hi), __builtin_amdgcn_fmed3f(t_[k + 1], q_lo, q_hi), \
                           cv_h, cv_m, 4 * (hf_) + k);                                                \
    if ((hf_) == 1) {                                                                                 \
      char* d_ = lds_w + (q_slb[i_] + (pbyte_));                                                      \
      *reinterpret_cast<opx8*>(d_) = cv_h;                                                            \
      *reinterpret_cast<opx8*>(d_ + PPL * 16) = cv_m;                                                 \
    }                                                                                                 \
  }
#define EMO_T_WSRC(p_) (EMO_CT2_W_CONST ? reinterpret_cast<const char*>(a.wpk) : (p_))
// piece k = 0 .. 8 of a half-stage's kernel rows (wave w copies pieces w, w + 4, w + 8 of each of the three rows) into W[wb_]
#define EMO_T_DMA_PIECE(ptr_, wb_, k_)                                                                \
  {                                                                                                   \
    const int row_ = (k_) / 3, j_ = wave + 4 * ((k_) % 3);                                            \
    emo_dma16_pinned_s(EMO_T_WSRC((ptr_) + (row_ * Cfg::WROW_BYTES + j_ * 1024)), lane16,             \
                       smem_lds + (unsigned)((Cfg::OFF_W + (wb_) * Cfg::WSTAGE) * 16 + row_ * Cfg::WROW_BYTES + j_ * 1024)); \
  }

  // the partial products, smallest first: (weight plane, patch plane); the last one is the leading product
  constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};
  constexpr int NTE = Cfg::SCT / 256;
  float te_sc[NTE], te_sh[NTE], te_b = 0.0f;

  bool chained_in = false;                 // this item's first stage (and its second patch) were staged by the previous item
  int pp = 0;                              // patch buffer of the item's current stage
  for (int idx8 = blockIdx.x >> 3; idx8 < n_mine; idx8 += l_stride) {
#if EMO_S_TIMING
  unsigned long long tstamp[12];     // measurement builds (tools/conv_phase_timing.py): s_memtime of wave 0 at the phase boundaries
  for (int k = 0; k < 12; ++k) tstamp[k] = 0;
#endif
  EMO_S_STAMP(0)
  EMO_P_DECODE(it_, l_base + idx8)
  int nx_cotile = 0, nx_n = 0, nx_ptile = 0, nx_x0 = 0, nx_y0 = 0, nx_z0 = 0;
  bool chain_out = false, nxq_ok = false;
  unsigned nxq_off = 0;
  if (EMO_S_CHAIN && idx8 + l_stride < n_mine) {
    EMO_P_DECODE(nx_, l_base + idx8 + l_stride)
    chain_out = nx_n == it_n && nst >= 2;
    EMO_P_CURSOR_OF(nx_, nxq_ok, nxq_off)
  }
  (void)nx_ptile;
  // (declared dead here: conv_igemm_bf16x3.h)
#pragma unroll
  for (int st_ = 0; st_ < 2; ++st_)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(fa_[st_][pl][i]));
#pragma unroll
      for (int j = 0; j < TP; ++j) asm volatile("" : "=v"(fb_[st_][pl][j]));
    }
  asm volatile("" : "=v"(cv_h));
  asm volatile("" : "=v"(cv_m));
  if (EMO_S_CHAIN && chained_in) {
    // P[pp] holds the converted patch of stage 0, W[0] the kernel rows of (c0, stage 0), qv the landed loads of stage 1, q_sc /
    // q_sh its first table entries; the tables are the sample's.  What is left: bias entries, the first two pieces of (c0 + 1, 0)
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    if (tid < 2 * BM) {
      const int t2_ = tid & (BM - 1);
      smem[(tid < BM ? Cfg::OFF_BIAS_F : Cfg::OFF_BIAS2_F) + (t2_ >> 5) * 32 + (t2_ & 3) * 8 + ((t2_ & 31) >> 2)] = te_b;
    }
    const char* const w1_ = EMO_T_WPTR(it_cotile + 1, 0);
    EMO_T_DMA_PIECE(w1_, 1, 0)
    EMO_T_DMA_PIECE(w1_, 1, 1)
    EMO_P_BARRIER(2);
  } else {
    // ---- full prologue: tables, the nine pieces of (c0, stage 0), the patch of stage 0 converted into P[0], the loads of stage 1 ----
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "=v"(qv[u]));
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    EMO_P_CURSOR_OF(it_, lq_ok, lq_off)
    lq_z0 = it_z0;
#pragma unroll
    for (int k = 0; k < NTE; ++k) {
      const int c = tid + 256 * k;
      const bool real = has_affine && c < a.Cin;
      te_sc[k] = real ? a.scale[(long)it_n * a.Cin + c] : 1.0f;
      te_sh[k] = real ? a.shift[(long)it_n * a.Cin + c] : 0.0f;
    }
    if (tid < 2 * BM && a.bias != nullptr) {
      const int co_ = it_cotile * BM + tid;
      te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];
    }
    {
      const char* const w0_ = EMO_T_WPTR(it_cotile, 0);
#pragma unroll
      for (int k = 0; k < 9; ++k) EMO_T_DMA_PIECE(w0_, 0, k)
    }
    ld_stage = 0; ld_cc = 0; ld_kd = 0;
    EMO_T_SET_STAGE_VARS()
    EMO_T_ISSUE_BEGIN()
    EMO_T_ISSUE_LOADS(0, 8)
#pragma unroll
    for (int k = 0; k < NTE; ++k) {       // (without an affine the index wraps at SCT: identity entries)
      const int c = tid + 256 * k;
      if (c < min(a.Cin, Cfg::SCT)) {
        sct[c] = te_sc[k] * in_scale;
        sct[Cfg::SCT + c] = te_sh[k] * in_scale;
      }
    }
    if (tid < 2 * BM) {
      const int t2_ = tid & (BM - 1);
      smem[(tid < BM ? Cfg::OFF_BIAS_F : Cfg::OFF_BIAS2_F) + (t2_ >> 5) * 32 + (t2_ & 3) * 8 + ((t2_ & 31) >> 2)] = te_b;
    }
    EMO_P_WAIT(0);
    EMO_T_TOUCH_QUAD()
    __syncthreads();   // scale / shift tables visible
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      EMO_T_HALF_TABLE(0)
      EMO_T_CONV_HALF(Cfg::OFF_P * 16, i, 0)
      EMO_T_HALF_TABLE(1)
      EMO_T_CONV_HALF(Cfg::OFF_P * 16, i, 1)
    }
    if (nst > 1) {                        // (one-stage item: the same patch again, a dead re-stage)
      ++ld_stage;
      if (++ld_kd == a.KD) { ld_kd = 0; ++ld_cc; }
    }
    EMO_T_SET_STAGE_VARS()
    EMO_T_ISSUE_BEGIN()
    EMO_T_ISSUE_LOADS(0, 8)
    EMO_T_HALF_TABLE(0)
    {
      const char* const w1_ = EMO_T_WPTR(it_cotile + 1, 0);
      EMO_T_DMA_PIECE(w1_, 1, 0)
      EMO_T_DMA_PIECE(w1_, 1, 1)
    }
    EMO_P_BARRIER(0);                    // (P[0] visible, W[0] and the loads of stage 1 landed)
    EMO_T_TOUCH_QUAD()
    pp = 0;
  }

  // (EMO_CT2_RES_EARLY: the first residual loads of the item's first channel tile issued here, a K loop ahead of their use --
  // measured slower, see the macro.  Ordinary loads: the compiler waits for them itself; its count of younger loads misses the
  // pinned ones, which only makes the wait stricter)
  floatx4 rv_first[8];
#if EMO_CT2_RES_EARLY
  if (epi_mode == 1) conv_epilogue_fast_issue<TW, TP, BM, 1, 0, true>(a, rv_first, it_n, it_cotile, it_x0, it_y0, it_z0, wp, lane);
  else if (epi_mode == 2) conv_epilogue_fast_issue<TW, TP, BM, 2, 0, true>(a, rv_first, it_n, it_cotile, it_x0, it_y0, it_z0, wp, lane);
#endif
  // ---- K loop: one stage = two half-stages (header comment) ----
  EMO_S_STAMP(1)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_lo[c][i][j][r] = 0.0f; acc_hi[c][i][j][r] = 0.0f; }
  {
    const int pb0_ = (Cfg::OFF_P + pp * PBUF) * 16;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) EMO_T_LOAD_FRAGS_PLANE(0, pl, Cfg::OFF_W, pb0_, 0, 0)      // (first half-stage: W[0], P[pp])
  }
  for (int cg = 0; cg < nst; ++cg) {
    const int pcur_b = (Cfg::OFF_P + pp * PBUF) * 16, pnxt_b = (Cfg::OFF_P + (pp ^ 1) * PBUF) * 16;
    const bool last_ = cg + 1 >= nst;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // half-stage t = (cg, h); t + 1 = (cg, 1) resp. (cg + 1, 0), t + 2 = (cg + 1, h); past the item's end: the next item's
      // stage 0 when chained, the last stage again (dead) otherwise.  Pointers from SELECTED indices: no branch in the loop
      const int c1_ = h == 0 ? it_cotile + 1 : (last_ && chain_out ? nx_cotile : it_cotile);
      const int k1_ = h == 0 ? cg : (last_ ? (chain_out ? 0 : nst - 1) : cg + 1);
      const int c2_ = (last_ && chain_out ? nx_cotile : it_cotile) + h;
      const int k2_ = last_ ? (chain_out ? 0 : nst - 1) : cg + 1;
      const char* const dma_ptr = EMO_T_WPTR(c1_, k1_);
      const char* const dma_ptr2 = EMO_T_WPTR(c2_, k2_);
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int gs = 0; gs < 9; ++gs) {
        const int r = gs / 3, s = gs % 3;
        const int fcur = (h * 9 + gs) & 1, fnxt = fcur ^ 1;
        (void)r; (void)s;
        if (gs == 8) { EMO_P_BARRIER(0); }
        if (gs == 8 && h == 1) EMO_T_TOUCH_QUAD()          // (the loads of stage cg + 2 have landed behind the barrier)
        if (gs == 0 && h == 1) {
          // the patch loads of stage cg + 2; past the item's end: the next item's stages 0 / 1 (chained), a dead re-stage otherwise
          const bool sw_ = chain_out && cg + 2 == nst;
          const int tgt_ = (chain_out && cg + 2 > nst) ? 1 : ((cg + 2) < nst ? cg + 2 : nst - 1);
          lq_ok = sw_ ? nxq_ok : lq_ok;
          lq_off = sw_ ? nxq_off : lq_off;
          lq_z0 = sw_ ? nx_z0 : lq_z0;
          const int adv_ = (!sw_ && tgt_ != ld_stage) ? 1 : 0;
          int kd_ = ld_kd + adv_, cc_ = ld_cc;
          if (kd_ == a.KD) { kd_ = 0; ++cc_; }
          ld_stage = sw_ ? 0 : ld_stage + adv_;
          ld_cc = sw_ ? 0 : cc_;
          ld_kd = sw_ ? 0 : kd_;
          EMO_T_SET_STAGE_VARS()
          EMO_T_ISSUE_BEGIN()
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const int rn = gs < 8 ? (gs + 1) / 3 : 0, sn = gs < 8 ? (gs + 1) % 3 : 0;
          const int wbn = Cfg::OFF_W + (gs < 8 ? h : h ^ 1) * Cfg::WSTAGE + rn * WROW;
          const int pbn = (gs == 8 && h == 1) ? pnxt_b : pcur_b;       // (half-stage 1 reads the same patch as half-stage 0)
#pragma unroll
          for (int pl = 0; pl < NPL + 1; ++pl) {
            if (pl < NPL) EMO_T_LOAD_FRAGS_PLANE(fnxt, pl, wbn, pbn, rn, sn)
            if (gs == 8 && pl < 2) {
              // pieces 0, 1 of half-stage t + 2 into W[h] (free behind the barrier).  Last stage, h = 1: W[1] is the epilogue's
              // scratch -- the pieces (dead, or the next item's, which its short prologue fetches) go to the idle patch buffer
              const int j_ = wave + 4 * pl;
              const unsigned dst_ = smem_lds + ((h == 1 && last_) ? (unsigned)(pcur_b + j_ * 1024)
                                                                   : (unsigned)((Cfg::OFF_W + h * Cfg::WSTAGE) * 16 + j_ * 1024));
              emo_dma16_pinned_s(EMO_T_WSRC(dma_ptr2 + j_ * 1024), lane16, dst_);
            }
            if (gs < 4 && pl == 0) EMO_T_DMA_PIECE(dma_ptr, h ^ 1, 2 + gs)
            if (h == 1 && gs < 4 && pl == 1) EMO_T_ISSUE_LOADS(2 * gs, 2 * gs + 2)
            if (gs == 4) EMO_T_DMA_PIECE(dma_ptr, h ^ 1, 6 + pl)
          }
        }
        if (h == 0 && gs < 8) {
          EMO_T_CONV_HALF(pnxt_b, gs >> 1, gs & 1)
          if (gs < 7) { EMO_T_HALF_TABLE((gs + 1) & 1) }     // (read behind this step's last use of the registers)
        }
        if (h == 1 && gs == 7) { EMO_T_HALF_TABLE(0) }       // (what the next stage's first unit converts with)
#pragma unroll
        for (int p = 0; p < NPROD; ++p) {
          const int pa = PA3[p], pb = PB3[p];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              floatx16& acc_ = (pa == 0 && pb == 0) ? acc_lo[h][i][j] : acc_hi[h][i][j];
              acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[fcur][pb][j], fa_[fcur][pa][i], acc_, 0, 0, 0);
            }
        }
        if (EMO_S_PIN) {
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
      }
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
    pp ^= 1;
  }
  EMO_S_STAMP(2)
  {
    // ---- epilogue, once per channel tile of the pair; the transposition scratch is W[1] ----
    float* const scratch = smem + (Cfg::OFF_W + Cfg::WSTAGE) * 4 + wave * Cfg::EPI_WAVE;
    const int ep_n = it_n, ep_cotile = it_cotile, ep_ptile = it_ptile, ep_x0 = it_x0, ep_y0 = it_y0, ep_z0 = it_z0;
    EMO_P_WAIT(0);
    EMO_S_STAMP(5)
    __syncthreads();
    EMO_S_STAMP(6)
    EMO_S_STAMP(7)
    if (EMO_S_CHAIN && chain_out && tid < 2 * BM && a.bias != nullptr) {   // the next item's bias entries
      const int co_ = nx_cotile * BM + tid;
      te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];
    }
// both tiles of the pair: the second tile's first residual loads are issued in front of the first tile's epilogue (they have landed
// when their turn comes: the residual is a tensor another kernel wrote long ago -- an HBM round trip that the single-tile kernel
// waits out in every item)
#define EMO_T_EPI_FAST(RES_)                                                                                                      \
    {                                                                                                                              \
      floatx4 (&rv_)[8] = rv_first;                                                                                                \
      floatx4 rvn_[8];                                                                                                             \
      if (!EMO_CT2_RES_EARLY) conv_epilogue_fast_issue<TW, TP, BM, RES_, 0, true>(a, rv_, ep_n, ep_cotile, ep_x0, ep_y0, ep_z0, wp, lane); \
      conv_epilogue_fast_issue<TW, TP, BM, RES_, 0, true>(a, rvn_, ep_n, ep_cotile + 1, ep_x0, ep_y0, ep_z0, wp, lane);            \
      conv_epilogue_fast_finish<TW, TM, TP, WGP, BM, SPLIT, Cfg::EPI_ROWF, RES_, true, true>(                                      \
          a, acc_lo[0], acc_hi[0], rv_, scratch, smem + Cfg::OFF_BIAS_F, smem + Cfg::OFF_STAT_F, ep_n, ep_cotile, ep_ptile, ep_x0,  \
          ep_y0, ep_z0, wp, half, l32, lane, tid EMO_S_TSTAMP_ARG);                                                                \
      EMO_S_STAMP(10)                                                                                                              \
      conv_epilogue_fast_finish<TW, TM, TP, WGP, BM, SPLIT, Cfg::EPI_ROWF, RES_, true, true>(                                      \
          a, acc_lo[1], acc_hi[1], rvn_, scratch, smem + Cfg::OFF_BIAS2_F, smem + Cfg::OFF_STAT2_F, ep_n, ep_cotile + 1, ep_ptile,  \
          ep_x0, ep_y0, ep_z0, wp, half, l32, lane, tid EMO_S_TSTAMP_ARG);                                                         \
    }
    if (epi_mode == 1) EMO_T_EPI_FAST(1)
    else if (epi_mode == 2) EMO_T_EPI_FAST(2)
    else EMO_T_EPI_FAST(0)
#if EMO_S_TIMING
    tstamp[8] = tstamp[10];          // (measurement builds: stamp 8 = first tile written, 9 = second tile written)
#endif
#undef EMO_T_EPI_FAST
  }
  if (a.sat_flag != nullptr && sat_m > 65504.0f) *a.sat_flag = 1;   // (every writer stores the same value)
#if EMO_S_TIMING
  EMO_S_STAMP(3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  EMO_S_STAMP(4)
  {
    const int ep_L_ = l_base + idx8;
    if (tid == 0 && ep_L_ < EMO_S_TLOG_N) {
      unsigned long long* t_ = emo_s_tlog + (long)ep_L_ * EMO_S_TLOG_W;
#pragma unroll
      for (int k = 0; k < 12; ++k) t_[k] = tstamp[k];
      t_[12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
      t_[13] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
      t_[14] = (unsigned long long)blockIdx.x;
    }
  }
#endif
  // the next prologue overwrites the tables and W[1]: every wave must be out of the epilogue first
  __syncthreads();
  // tile statistics, second half (conv_epilogue_rows_stats, for both tiles at once and behind the barrier above instead of one of
  // their own each): thread c of the first 128 combines the four waves' (mean, M2) of channel c with the equal-count update.
  // The exchange areas are next written by the NEXT item's epilogue, a K loop away
  EMO_P_COMBINE_STATS(Cfg::OFF_STAT_F, Cfg::OFF_STAT2_F, true)
  chained_in = chain_out;
  }
#undef EMO_T_WPTR
#undef EMO_T_LOAD_FRAGS_PLANE
#undef EMO_T_SET_STAGE_VARS
#undef EMO_T_ISSUE_BEGIN
#undef EMO_T_ISSUE_LOADS
#undef EMO_T_HALF_TABLE
#undef EMO_T_TOUCH_QUAD
#undef EMO_T_CONV_HALF
#undef EMO_T_DMA_PIECE
#undef EMO_T_WSRC
}

// Launches the channel-tile PAIRS of the layer on conv_igemm_bf16x3_ct2_kernel when that fills the chip; *rest_cot0 = the first
// channel tile NOT covered (0: nothing was launched, the caller runs the whole layer on the single-tile kernel; n_cotiles: all
// done; otherwise the odd last tile is the caller's, ConvArgs::cot0).  EMO_CONV_CT2=0 disables it (A/B).
template <int TR, int TW, bool UPS>
int conv_f16x2_ct2_launch(ConvArgs a, hipStream_t s, int* rest_cot0) {
  using Cfg = ConvCfgS2<TR, TW, UPS>;
  *rest_cot0 = 0;
  // (read at every launch: tests and A/B runs toggle them inside one process; a launch is microseconds)
  const char* const e_on = getenv("EMO_CONV_CT2");
  const char* const e_min = getenv("EMO_CONV_CT2_MIN_ITEMS");
  if ((e_on && atoi(e_on) == 0) || a.ksplit != 1 || a.run_if != nullptr) return EMO_OK;
  // the straight-line epilogue forms only (conv_igemm_bf16x3.h, epi_mode): final output, no activation, whole 64-channel tiles,
  // 16-byte aligned output, 16- / 8-byte aligned same-size / half-size residual, offsets inside 2^31 bytes per sample
  if (a.act != EMO_ACT_NONE || a.Cout % Cfg::BM != 0 || (a.Wl & 3) != 0 || (long)a.Dl * a.Hl * a.Wl > (1l << 23) ||
      (reinterpret_cast<unsigned long long>(a.out) & 15ull) != 0 ||
      (a.res != nullptr && (reinterpret_cast<unsigned long long>(a.res) & (a.res_ups ? 7ull : 15ull)) != 0)) return EMO_OK;
  if (a.Wl % TW || a.Hl % TR || a.Cin % 8) return EMO_OK;                 // (the single-tile launcher reports what is unsupported)
  if (a.scale && a.Cin > Cfg::SCT) return EMO_OK;
  if ((unsigned long long)a.Cin * a.D * a.H * a.W * 4ull >= (1ull << 32)) return EMO_OK;
  if ((reinterpret_cast<unsigned long long>(a.x) & 15ull) || (a.W & 3)) return EMO_OK;
  const int cot = (a.Cout + Cfg::BM - 1) / Cfg::BM;
  const int pairs = cot / 2;
  const long nt = (long)(a.Wl / TW) * (a.Hl / TR) * a.Dl;
  const int ncu = emo_cu_count();
  // enough pair items for two per CU: below that the single-tile kernel's twice as many, half as long items balance better
  const long min_items = e_min ? atol(e_min) : 2l * ncu;
  if (pairs < 1 || nt > 0x7fffffffL || a.N > 65535 || nt * pairs * a.N > 0x7fffffffL || nt * pairs * a.N < min_items) return EMO_OK;
  auto kern = conv_igemm_bf16x3_ct2_kernel<TR, TW, UPS>;
  const int rc = emo_raise_dynamic_lds(kern);
  if (rc != EMO_OK) return rc;
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl;
  a.n_cchunks = (a.Cin + Cfg::KC - 1) / Cfg::KC;
  a.stages_per_split = a.n_cchunks * a.KD;
  a.partial = nullptr;
  a.cot0 = 0;
  a.n_cotiles = pairs;                         // (pairs: EMO_P_DECODE)
  a.n_work = (int)(nt * pairs * a.N);
  const int grid = a.n_work > ncu ? ncu : a.n_work;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)Cfg::LDS_BYTES, s, a);
  *rest_cot0 = 2 * pairs;
  return emo_launch_status();
}
