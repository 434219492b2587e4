// POINTWISE (1x1) fp32 convolution on the fp16 matrix pipes: the fp16 split of conv_igemm_bf16x3.h (scaled operands as two fp16
// terms, three products, fp32 accumulation in two accumulator sets, device-checked operand range) with two 64-channel output
// tiles per work item on one converted patch, as conv_igemm_f16x2_ct2.h -- for the layers that have no taps to amortise the
// staging over: the decoder's 1536 -> 512 entry convolution and the 1x1 skip convolutions of its up-blocks (on the fp32 MFMA
// kernel they ran at 95-117 TF, two of them below stock MIOpen: profiles/r4_conv_microbench.jsonl).
//
// A stage is 32 input channels (two MFMA k-steps of 16) of a 256-position tile: 4 x 64 pixels, no halo.
//   LDS   P[2][plane][4 groups of 8 channels][slot][8]   the converted patch, double-buffered by stage parity (wave w stages
//                                                         group w: 8 quad loads, 8 conversion units per stage and thread)
//         W[2][tile][plane][k-step][half][64][8]           the weights of a stage for both channel tiles: 8 KB per tile and
//                                                         stage, host-packed in this order, copied by LDS-DMA (4 pieces per wave)
//   step gs = 0 .. 3 of a stage: channel tile h = gs >> 1, k-step gs & 1; 12 MFMAs each.  The patch fragments of a k-step are
//   read once and used by both tiles.
//   stage cg, parity par:  step 0  the quad loads of stage cg + 2 into qv[par]; steps 0 .. 2: the patch of stage cg + 1 converted
//                                   from qv[par ^ 1] into P[par ^ 1] (3 + 3 + 2 units)
//                          step 1  the weight pieces of stage cg + 1 into W[par ^ 1]
//                          step 3  ONE barrier (everything in flight is drained: vmcnt(0)); the first fragments of stage cg + 1
// (A stage is 1.5 k cycles of MFMA work -- shorter than a memory round trip: the kernel is latency-bound at ~ 4.4 k per stage.
// Running the loads three stages ahead (issued behind the barrier into the buffer whose conversion just ended) was built and
// measured, tools/session/r5_call14.sh: the loads then cross the loop's exit in flight, the compiler copies their destination
// registers on the exit edge -- the audit's epilogue walk flags it -- and a first version declared the buffer dead at the top of
// the item loop: every chained item converted garbage, raised its overflow word, and was silently recomputed by the guarded
// fp32 launch (correct results at a third of the speed).  Reverted; the fix is a peeled last stage.)
// Items of a persistent block are chained (even stage counts), the epilogue is conv_igemm_bf16x3.h's straight-line form, run
// per channel tile; it transposes through the patch buffer the last stage read.  A layer with an odd number of channel tiles
// (320 = 5 x 64) runs its last tile in a pair whose second half computes on zero weights and is not written.
// Launch form: final output (no K split), no activation, Cout % 64 == 0, 16-byte aligned tensors; anything else stays on the
// fp32 MFMA kernel (conv_igemm.h), which is also the guarded exact recomputation behind a raised overflow word.
#pragma once
#include "conv_igemm_bf16x3.h"

template <int TR, int TW>
struct ConvCfgP {
  static constexpr int BM = 64, BP = 256, TM = 2, TP = 2, WGP = 4, KC = 32, NPL = 2;
  static constexpr int NQ = TW / 4;                      // quads per tile row
  static constexpr int SUB = ((TR * NQ + 4 + 11) / 16) * 16 + 4;   // slots per sub-row: = 4 (mod 16), conv_igemm_bf16x3.h
  static constexpr int CHS = 4 * SUB;                    // slots per 8-channel group
  static constexpr int NG = 4;                           // 8-channel groups per stage: one per wave
  static constexpr int PPL = NG * CHS;                   // one plane of the patch
  static constexpr int PBUF = NPL * PPL;
  static constexpr int OFF_P = 0;
  static constexpr int WTILE = NPL * 2 * 2 * BM;         // slots of one channel tile's weights of a stage: [plane][k-step][half][BM]
  static constexpr int WTILE_BYTES = WTILE * 16;
  static constexpr int WSTAGE = 2 * WTILE;               // both tiles
  static constexpr int OFF_W = 2 * PBUF;
  static constexpr int OFF_SCT = OFF_W + 2 * WSTAGE;     // scale / shift tables (fp32)
  static constexpr int SCT = 1024;
  static constexpr int EPI_ROWF = 68;
  static constexpr int EPI_WAVE = 32 * EPI_ROWF;
  static constexpr int OFF_BIAS_F = OFF_SCT * 4 + 2 * SCT;
  static constexpr int OFF_STAT_F = OFF_BIAS_F + BM;
  static constexpr int OFF_BIAS2_F = OFF_STAT_F + 2 * WGP * BM;
  static constexpr int OFF_STAT2_F = OFF_BIAS2_F + BM;
  static constexpr int LDS_BYTES = (OFF_STAT2_F + 2 * WGP * BM) * 4;
  static_assert(TR * TW == BP && TR * NQ == 64, "256 positions = 64 quads: one per lane of the staging wave");
  static_assert(WTILE_BYTES == 8 * 1024, "8 DMA pieces per channel tile and stage");
  static_assert(PBUF * 4 >= WGP * EPI_WAVE, "the epilogue transposes through one patch buffer");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int TR, int TW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_igemm_bf16x3_p1_kernel(const ConvArgs a) {
  using Cfg = ConvCfgP<TR, TW>;
  using opx8 = halfx8;
  constexpr int SPLIT = 2, NPL = 2, NPROD = 3;
  constexpr int BM = Cfg::BM, TM = Cfg::TM, TP = Cfg::TP, WGP = Cfg::WGP, KC = Cfg::KC;
  constexpr int NQ = Cfg::NQ, SUB = Cfg::SUB, CHS = Cfg::CHS, PPL = Cfg::PPL, PBUF = Cfg::PBUF;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int wp = wave;
  const int p0 = wp * TP * 32;
  float sat_m = 0.0f;

  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const bool has_affine = a.scale != nullptr;
  const int epi_mode = __builtin_amdgcn_readfirstlane(a.res == nullptr ? 0 : (a.res_ups ? 2 : 1));   // (host: conv_f16x2_p1_launch)
  const float in_scale = a.in_scale;
  constexpr float CLAMP_HI = 65504.0f;
  const float clamp_lo = a.relu_in ? 0.0f : -CLAMP_HI;
  const int nst = a.n_cchunks;                           // stages of an item: 32-channel chunks
  const int nptiles = a.tiles_x * a.tiles_y * a.tiles_z;
  const int n_cot_all = (a.Cout + BM - 1) / BM;          // channel tiles of the layer (an odd count: the last pair is half empty)

  // ---- staging map: wave w stages the w-th 8-channel group of a stage, lane u the u-th quad of the tile (row u / NQ) ----
  const int q_r = lane / NQ, q_c = lane - q_r * NQ;
  int q_slb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) q_slb[i] = (wave * CHS + i * SUB + lane) * 16;

  floatx16 acc_lo[2][TM][TP], acc_hi[2][TM][TP];

  const int q8 = a.n_work >> 3, r8 = a.n_work & 7;
  const int xcd = blockIdx.x & 7;
  const int n_mine = q8 + (xcd < r8 ? 1 : 0);
  const int l_base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int l_stride = (gridDim.x + 7) >> 3;
  int it_cotile = 0, it_n = 0, it_ptile = 0, it_x0 = 0, it_y0 = 0, it_z0 = 0;
  unsigned lq_off = 0;
  int lq_z0 = 0;
#define EMO_P_DECODE(P_, L_)                                                                          \
  {                                                                                                   \
    const int l_ = (L_);                                                                              \
    const int cot_ = l_ % a.n_cotiles;                                                                \
    const int rest_ = l_ / a.n_cotiles;                                                               \
    const int n_ = rest_ / nptiles;                                                                   \
    int bx_ = rest_ - n_ * nptiles;                                                                   \
    P_##ptile = __builtin_amdgcn_readfirstlane(bx_);                                                  \
    const int tx_ = bx_ % a.tiles_x; bx_ /= a.tiles_x;                                                \
    const int ty_ = bx_ % a.tiles_y; bx_ /= a.tiles_y;                                                \
    P_##cotile = __builtin_amdgcn_readfirstlane(2 * cot_);                                            \
    P_##n = __builtin_amdgcn_readfirstlane(n_);                                                       \
    P_##x0 = __builtin_amdgcn_readfirstlane(tx_ * TW);                                                \
    P_##y0 = __builtin_amdgcn_readfirstlane(ty_ * TR);                                                \
    P_##z0 = __builtin_amdgcn_readfirstlane(bx_);                                                     \
  }
// packed weights of (channel tile c_, stage k_); the host pads the tiles to an even count (pack.pack_weight_f16x2_1x1)
#define EMO_P_WPTR(c_, k_) (reinterpret_cast<const char*>(a.wpk) + (long)((c_) * nst + (k_)) * Cfg::WTILE_BYTES)
#define EMO_P_CURSOR_OF(P_, off_) { off_ = (unsigned)((P_##y0 + q_r) * a.W + P_##x0 + 4 * q_c) * 4u; }

  // LDS byte offsets of the lane's operands: weight fragment (+ buffer / tile / plane / k-step immediates) and the patch slot of
  // the lane's output pixel (+ half * CHS: the lane's 8-channel group of the k-step; + buffer / plane / k-step immediates)
  const int a_off = (half * BM + l32) * 16;
  int b_off[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW, row = p / TW;
    b_off[j] = (half * CHS + (col & 3) * SUB + row * NQ + (col >> 2)) * 16;
  }
  const char* const lds_c = reinterpret_cast<const char*>(smem);
  char* const lds_w = reinterpret_cast<char*>(smem);
  opx8 fa_[2][NPL][TM];        // [register set: step parity][plane][tile row]
  opx8 fb_[2][NPL][TP];        // [k-step][plane][tile column]: read once per stage and k-step, used by both channel tiles
#define EMO_P_LOAD_A(set_, wb_, h_, ks_)                                                              \
  { _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) _Pragma("unroll") for (int i = 0; i < TM; ++i)  \
      fa_[set_][pl][i] = *reinterpret_cast<const opx8*>(lds_c + a_off + (Cfg::OFF_W + (wb_) * Cfg::WSTAGE + (h_) * Cfg::WTILE + ((pl * 2 + (ks_)) * 2) * BM + i * 32) * 16); }
#define EMO_P_LOAD_B(pb_, ks_)                                                                        \
  { _Pragma("unroll") for (int pl = 0; pl < NPL; ++pl) _Pragma("unroll") for (int j = 0; j < TP; ++j)  \
      fb_[ks_][pl][j] = *reinterpret_cast<const opx8*>(lds_c + b_off[j] + (Cfg::OFF_P + (pb_) * PBUF + pl * PPL + (ks_) * 2 * CHS) * 16); }

  float* const sct = smem + Cfg::OFF_SCT * 4;
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);
  const unsigned lane16 = (unsigned)lane * 16u;

  floatx4 qv[2][8];             // raw patch registers, double-buffered by stage parity
  float q_lo[2], q_hi[2];
  int q_tix[2];
  floatx4 t_sc[2], t_sh[2];     // scale / shift of the 8 channels of the group being converted
  opx8 cv_h, cv_m;
  emo_intx4 xrs = emo_raw_buffer(a.x);
  unsigned usoff[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) usoff[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)u * (unsigned)DHW * 4u));

  int ld_stage;                 // the stage whose patch is being loaded
  unsigned q_vo;
#define EMO_P_ISSUE_BEGIN(b_)                                                                         \
  {                                                                                                   \
    const int c0_ = ld_stage * KC + wave * 8;                                                         \
    const bool cv_ = c0_ < a.Cin;                                                                     \
    const int cs_ = cv_ ? c0_ : 0;                                                                    \
    q_lo[b_] = cv_ ? clamp_lo : 0.0f;                                                                 \
    q_hi[b_] = cv_ ? CLAMP_HI : 0.0f;                                                                 \
    q_vo = lq_off + ((unsigned)cs_ * (unsigned)DHW + (unsigned)(lq_z0 * HW)) * 4u;                    \
    q_tix[b_] = (has_affine ? cs_ : (cs_ & (Cfg::SCT - 1))) >> 2;                                     \
  }
#define EMO_P_ISSUE_LOADS(b_)                                                                         \
  { _Pragma("unroll") for (int u = 0; u < 8; u += 2) emo_bload4x2_pinned(xrs, q_vo, usoff[u], usoff[u + 1], qv[b_][u], qv[b_][u + 1]); }
#define EMO_P_TABLES(b_)                                                                              \
  {                                                                                                   \
    const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + q_tix[b_];                           \
    t_sc[0] = t4_[0]; t_sc[1] = t4_[1]; t_sh[0] = t4_[Cfg::SCT / 4]; t_sh[1] = t4_[Cfg::SCT / 4 + 1]; \
  }
#define EMO_P_TOUCH_QUAD(b_) { _Pragma("unroll") for (int u = 0; u < 8; ++u) emo_touch4(qv[b_][u]); }
// conversion unit: channels 4 * hf_ .. + 3 of pixel i_ of buffer b_ into patch buffer pb_ (conv_igemm_bf16x3.h, SPLIT = 2)
#define EMO_P_CONV_HALF(b_, pb_, i_, hf_)                                                             \
  {                                                                                                   \
    float t_[4];                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                     \
      t_[k] = __fmaf_rn(qv[b_][4 * (hf_) + k][i_], t_sc[hf_][k], t_sh[hf_][k]);                       \
    sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[0])), __builtin_fabsf(t_[1])); \
    sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[2])), __builtin_fabsf(t_[3])); \
    _Pragma("unroll") for (int k = 0; k < 4; k += 2)                                                  \
      emo_split_f16x2_pair(__builtin_amdgcn_fmed3f(t_[k], q_lo[b_], q_hi[b_]),                        \
                           __builtin_amdgcn_fmed3f(t_[k + 1], q_lo[b_], q_hi[b_]), cv_h, cv_m, 4 * (hf_) + k); \
    if ((hf_) == 1) {                                                                                 \
      char* d_ = lds_w + q_slb[i_] + (Cfg::OFF_P + (pb_) * PBUF) * 16;                                \
      *reinterpret_cast<opx8*>(d_) = cv_h;                                                            \
      *reinterpret_cast<opx8*>(d_ + PPL * 16) = cv_m;                                                 \
    }                                                                                                 \
  }
// piece k = 0 .. 3 of a stage's weights: wave w copies pieces w and w + 4 of tile k >> 1 (k & 1 selects which) into W[wb_]
#define EMO_P_DMA_PIECE(ptr0_, ptr1_, wb_, k_)                                                        \
  {                                                                                                   \
    const int j_ = wave + 4 * ((k_) & 1);                                                             \
    emo_dma16_pinned_s((((k_) >> 1) ? (ptr1_) : (ptr0_)) + j_ * 1024, lane16,                         \
                       smem_lds + (unsigned)((Cfg::OFF_W + (wb_) * Cfg::WSTAGE + ((k_) >> 1) * Cfg::WTILE) * 16 + j_ * 1024)); \
  }
#define EMO_P_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define EMO_P_BARRIER(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")

  constexpr int PA3[3] = {1, 0, 0}, PB3[3] = {0, 1, 0};
  constexpr int NTE = Cfg::SCT / 256;
  float te_sc[NTE], te_sh[NTE], te_b = 0.0f;

  bool chained_in = false;
  for (int idx8 = blockIdx.x >> 3; idx8 < n_mine; idx8 += l_stride) {
  EMO_P_DECODE(it_, l_base + idx8)
  int nx_cotile = 0, nx_n = 0, nx_ptile = 0, nx_x0 = 0, nx_y0 = 0, nx_z0 = 0;
  bool chain_out = false;
  unsigned nxq_off = 0;
  if (EMO_S_CHAIN && idx8 + l_stride < n_mine) {
    EMO_P_DECODE(nx_, l_base + idx8 + l_stride)
    chain_out = nx_n == it_n;            // (stage counts are even: the buffer parities line up)
    EMO_P_CURSOR_OF(nx_, nxq_off)
  }
  (void)nx_ptile;
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
  if (EMO_S_CHAIN && chained_in) {
    // P[0] holds the converted patch of stage 0, W[0] the weights of stage 0 (both tiles), qv[1] the landed loads of stage 1,
    // t_sc / t_sh its table entries.  What is left: the bias entries
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    if (tid < 2 * BM) {
      const int t2_ = tid & (BM - 1);
      smem[(tid < BM ? Cfg::OFF_BIAS_F : Cfg::OFF_BIAS2_F) + (t2_ >> 5) * 32 + (t2_ & 3) * 8 + ((t2_ & 31) >> 2)] = te_b;
    }
    EMO_P_BARRIER(0);
  } else {
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    EMO_P_CURSOR_OF(it_, lq_off)
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
      const char* const w0_ = EMO_P_WPTR(it_cotile, 0);
      const char* const w1_ = EMO_P_WPTR(it_cotile + 1, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) EMO_P_DMA_PIECE(w0_, w1_, 0, k)
    }
    ld_stage = 0;
    EMO_P_ISSUE_BEGIN(0)
    EMO_P_ISSUE_LOADS(0)
    if (nst > 1) ++ld_stage;             // (one-stage item: the same patch again, a dead re-stage)
    EMO_P_ISSUE_BEGIN(1)
    EMO_P_ISSUE_LOADS(1)
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
    EMO_P_TOUCH_QUAD(0)
    EMO_P_TOUCH_QUAD(1)
    __syncthreads();   // scale / shift tables visible
    EMO_P_TABLES(0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      EMO_P_CONV_HALF(0, 0, i, 0)
      EMO_P_CONV_HALF(0, 0, i, 1)
    }
    EMO_P_TABLES(1)                      // (what the first stage converts with)
    EMO_P_BARRIER(0);                    // (P[0] visible)
  }

  // ---- K loop, two stages per iteration (buffer parities and fragment sets are compile-time constants) ----
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_lo[c][i][j][r] = 0.0f; acc_hi[c][i][j][r] = 0.0f; }
  EMO_P_LOAD_A(0, 0, 0, 0)
  EMO_P_LOAD_B(0, 0)
  for (int cg0 = 0; cg0 < nst; cg0 += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int cg = cg0 + par;          // (nst is even -- conv_f16x2_p1_launch: no exit between the two stages of an iteration; a
                                         // branch there makes the compiler carry the 256 accumulators through ordinary registers)
      // stage cg + 1 (its weights are fetched during this stage) -- past the item's end the next item's stage 0 (chained) or the
      // last stage again (dead); pointers from selected indices, no branch
      const bool nx1_ = EMO_S_CHAIN && par == 1 && chain_out && cg + 1 >= nst;
      const int k1_ = nx1_ ? 0 : ((cg + 1) < nst ? cg + 1 : nst - 1);
      const int c1_ = nx1_ ? nx_cotile : it_cotile;
      const char* const dma0 = EMO_P_WPTR(c1_, k1_);
      const char* const dma1 = EMO_P_WPTR(c1_ + 1, k1_);
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int gs = 0; gs < 4; ++gs) {
        const int h = gs >> 1, ks = gs & 1;
        const int fcur = gs & 1, fnxt = fcur ^ 1;          // (weight fragment sets alternate by step; four steps per stage)
        if (gs == 3) { EMO_P_BARRIER(0); }
        if (gs == 3) EMO_P_TOUCH_QUAD(par)                  // (the loads of stage cg + 2 have landed behind the barrier)
        if (gs == 0) {
          // the patch loads of stage cg + 2; chained item: past its end the next item's stages 0 / 1 (the cursor moves on in a
          // stage of parity 0: stage counts of chained items are even)
          const bool sw_ = EMO_S_CHAIN && par == 0 && chain_out && cg + 2 == nst;
          const int tgt_ = (EMO_S_CHAIN && par == 1 && chain_out && cg + 2 > nst) ? 1 : ((cg + 2) < nst ? cg + 2 : nst - 1);
          lq_off = sw_ ? nxq_off : lq_off;
          lq_z0 = sw_ ? nx_z0 : lq_z0;
          ld_stage = sw_ ? 0 : tgt_;
          EMO_P_ISSUE_BEGIN(par)
        }
        __builtin_amdgcn_sched_barrier(0);
        // fragments of the next step: weights of (h', k') = step gs + 1 -- of the next stage's step 0 behind the barrier -- and, in
        // steps 0 and 3, the patch fragments of the k-step that comes next
        if (gs < 3) { EMO_P_LOAD_A(fnxt, par, (gs + 1) >> 1, (gs + 1) & 1) } else { EMO_P_LOAD_A(fnxt, par ^ 1, 0, 0) }
        if (gs == 0) { EMO_P_LOAD_B(par, 1) }
        if (gs == 3) { EMO_P_LOAD_B(par ^ 1, 0) }
        if (gs == 0) EMO_P_ISSUE_LOADS(par)
        if (gs == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) EMO_P_DMA_PIECE(dma0, dma1, par ^ 1, k)
        }
        // conversion of the patch of stage cg + 1: units 0 .. 7 = (pixel, half), 3 + 3 + 2 over steps 0 .. 2
        if (gs == 0) { EMO_P_CONV_HALF(par ^ 1, par ^ 1, 0, 0) EMO_P_CONV_HALF(par ^ 1, par ^ 1, 0, 1) EMO_P_CONV_HALF(par ^ 1, par ^ 1, 1, 0) }
        if (gs == 1) { EMO_P_CONV_HALF(par ^ 1, par ^ 1, 1, 1) EMO_P_CONV_HALF(par ^ 1, par ^ 1, 2, 0) EMO_P_CONV_HALF(par ^ 1, par ^ 1, 2, 1) }
        if (gs == 2) { EMO_P_CONV_HALF(par ^ 1, par ^ 1, 3, 0) EMO_P_CONV_HALF(par ^ 1, par ^ 1, 3, 1) }
        if (gs == 3) { EMO_P_TABLES(par) }                  // (what the next stage converts with: the loads issued in this one)
#pragma unroll
        for (int p = 0; p < NPROD; ++p) {
          const int pa = PA3[p], pb = PB3[p];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
              floatx16& acc_ = (pa == 0 && pb == 0) ? acc_lo[h][i][j] : acc_hi[h][i][j];
              acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[ks][pb][j], fa_[fcur][pa][i], acc_, 0, 0, 0);
            }
        }
        if (EMO_S_PIN) {
          // { MFMA, LDS read, <= 9 VALU } for the step's fragment reads, then { MFMA, <= 10 VALU, LDS store }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
          }
#pragma unroll
          for (int k = 8; k < 4 * NPROD; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
  }
  {
    // ---- epilogue, once per channel tile of the pair; it transposes through the patch buffer the last stage read ----
    const int last_par = (nst - 1) & 1;
    float* const scratch = smem + (Cfg::OFF_P + last_par * PBUF) * 4 + wave * Cfg::EPI_WAVE;
    const int ep_n = it_n, ep_cotile = it_cotile, ep_ptile = it_ptile, ep_x0 = it_x0, ep_y0 = it_y0, ep_z0 = it_z0;
    const bool second = ep_cotile + 1 < n_cot_all;          // (an odd count of channel tiles: the last pair's second half is padding)
    EMO_P_WAIT(0);
    __syncthreads();
    if (EMO_S_CHAIN && chain_out && tid < 2 * BM && a.bias != nullptr) {
      const int co_ = nx_cotile * BM + tid;
      te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];
    }
#define EMO_P_EPI_FAST(RES_)                                                                                                      \
    {                                                                                                                              \
      floatx4 rv_[8], rvn_[8];                                                                                                     \
      conv_epilogue_fast_issue<TW, TP, BM, RES_, 0, true>(a, rv_, ep_n, ep_cotile, ep_x0, ep_y0, ep_z0, wp, lane);                 \
      if (second) conv_epilogue_fast_issue<TW, TP, BM, RES_, 0, true>(a, rvn_, ep_n, ep_cotile + 1, ep_x0, ep_y0, ep_z0, wp, lane); \
      conv_epilogue_fast_finish<TW, TM, TP, WGP, BM, SPLIT, Cfg::EPI_ROWF, RES_, true, true>(                                      \
          a, acc_lo[0], acc_hi[0], rv_, scratch, smem + Cfg::OFF_BIAS_F, smem + Cfg::OFF_STAT_F, ep_n, ep_cotile, ep_ptile, ep_x0,  \
          ep_y0, ep_z0, wp, half, l32, lane, tid);                                                                                 \
      if (second)                                                                                                                  \
        conv_epilogue_fast_finish<TW, TM, TP, WGP, BM, SPLIT, Cfg::EPI_ROWF, RES_, true, true>(                                    \
            a, acc_lo[1], acc_hi[1], rvn_, scratch, smem + Cfg::OFF_BIAS2_F, smem + Cfg::OFF_STAT2_F, ep_n, ep_cotile + 1,          \
            ep_ptile, ep_x0, ep_y0, ep_z0, wp, half, l32, lane, tid);                                                              \
    }
    if (epi_mode == 1) EMO_P_EPI_FAST(1)
    else if (epi_mode == 2) EMO_P_EPI_FAST(2)
    else EMO_P_EPI_FAST(0)
#undef EMO_P_EPI_FAST
  }
  if (a.sat_flag != nullptr && sat_m > 65504.0f) *a.sat_flag = 1;
  __syncthreads();
  // tile statistics, second half, for both tiles behind the barrier above (conv_igemm_f16x2_ct2.h)
  if (a.gn_stats != nullptr && tid < 2 * BM && it_cotile * BM + tid < a.Cout) {
    const int c_ = tid & (BM - 1);
    const float* const st_ = smem + (tid < BM ? Cfg::OFF_STAT_F : Cfg::OFF_STAT2_F);
    float mean = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int w = 0; w < WGP; ++w) mean += st_[(w * BM + c_) * 2 + 0];
    mean *= 1.0f / (float)WGP;
#pragma unroll
    for (int w = 0; w < WGP; ++w) {
      const float d = st_[(w * BM + c_) * 2 + 0] - mean;
      m2 += st_[(w * BM + c_) * 2 + 1] + (float)(TP * 32) * d * d;
    }
    // Written as TWO entries of 128 positions each, (mean, M2 / 2): gn_stats of a pointwise layer is laid out for 128-position
    // tiles, the tile of the fp32 MFMA kernel that recomputes the layer behind a raised overflow word (both fill the same
    // buffer).  Two equal halves with the same mean combine to exactly (mean, M2) over 256 positions
    float2* dst = reinterpret_cast<float2*>(a.gn_stats) + ((long)it_n * nptiles + it_ptile) * 2 * a.Cout + it_cotile * BM + tid;
    dst[0] = make_float2(mean, 0.5f * m2);
    dst[a.Cout] = make_float2(mean, 0.5f * m2);
  }
  chained_in = chain_out;
  }
#undef EMO_P_DECODE
#undef EMO_P_WPTR
#undef EMO_P_CURSOR_OF
#undef EMO_P_LOAD_A
#undef EMO_P_LOAD_B
#undef EMO_P_ISSUE_BEGIN
#undef EMO_P_ISSUE_LOADS
#undef EMO_P_TABLES
#undef EMO_P_TOUCH_QUAD
#undef EMO_P_CONV_HALF
#undef EMO_P_DMA_PIECE
#undef EMO_P_WAIT
#undef EMO_P_BARRIER
}

// Host side.  EMO_ERR_UNSUPPORTED for anything but the launch form of the header comment: the caller (conv_api.hip) then reports it,
// the Python planner (pack.PackedConv.plan_for) does not route such a launch here.
template <int TR, int TW>
int conv_f16x2_p1_launch(ConvArgs a, hipStream_t s) {
  using Cfg = ConvCfgP<TR, TW>;
  if (a.KD != 1 || a.ksplit != 1 || a.run_if != nullptr) return EMO_ERR_UNSUPPORTED;
  if (a.Wl % TW || a.Hl % TR || a.Wl != a.W || a.Hl != a.H) return EMO_ERR_UNSUPPORTED;          // (no fused upsample)
  if (a.scale && a.Cin > Cfg::SCT) return EMO_ERR_UNSUPPORTED;
  if ((unsigned long long)a.Cin * a.D * a.H * a.W * 4ull >= (1ull << 32)) return EMO_ERR_UNSUPPORTED;
  if ((reinterpret_cast<unsigned long long>(a.x) & 15ull) || (a.W & 3)) return EMO_ERR_UNSUPPORTED;
  if (a.act != EMO_ACT_NONE || a.Cout % Cfg::BM != 0 || (long)a.Dl * a.Hl * a.Wl > (1l << 23) ||
      (reinterpret_cast<unsigned long long>(a.out) & 15ull) != 0 ||
      (a.res != nullptr && (reinterpret_cast<unsigned long long>(a.res) & (a.res_ups ? 7ull : 15ull)) != 0)) return EMO_ERR_UNSUPPORTED;
  const int cot = a.Cout / Cfg::BM;
  const int pairs = (cot + 1) / 2;
  const long nt = (long)(a.Wl / TW) * (a.Hl / TR) * a.Dl;
  if (nt > 0x7fffffffL || a.N > 65535 || nt * pairs * a.N > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  auto kern = conv_igemm_bf16x3_p1_kernel<TR, TW>;
  const int rc = emo_raise_dynamic_lds(kern);
  if (rc != EMO_OK) return rc;
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl;
  a.n_cchunks = (a.Cin + Cfg::KC - 1) / Cfg::KC;
  if (a.n_cchunks & 1) return EMO_ERR_UNSUPPORTED;       // (the K loop runs two stages per iteration)
  a.stages_per_split = a.n_cchunks;
  a.partial = nullptr;
  a.cot0 = 0;
  a.n_cotiles = pairs;
  a.n_work = (int)(nt * pairs * a.N);
  const int ncu = emo_cu_count();
  const int grid = a.n_work > ncu ? ncu : a.n_work;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)Cfg::LDS_BYTES, s, a);
  return emo_launch_status();
}
