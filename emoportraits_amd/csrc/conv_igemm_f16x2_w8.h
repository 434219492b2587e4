// The two-tile fp16-split kernel of conv_igemm_f16x2_ct2.h with TWO WAVES PER SIMD -- "W8": 512 threads, eight waves of 256 registers.
//
// Why.  conv_igemm_bf16x3_ct2_kernel runs one wave per SIMD with 512 registers, all 256 accumulation registers in use.  Whatever
// that wave waits for, its SIMD's matrix pipe waits with it: in the K loop the in-order issue of 13 vector-memory instructions
// per half-stage (~45 cycles each with their wait states), the patch conversion and the barrier put 41 cycles on every 32-cycle
// MFMA slot, and the epilogue -- 15 % of a pair item (profiles/r5_conv_phase_timing_final.jsonl) -- is a chain of dependent
// LDS round trips and global stores that one wave walks alone, with nothing beside it.  Here the same work item (a channel-tile
// PAIR x 256 positions, the same LDS image: one converted patch for both tiles, W[0] / W[1], one barrier per half-stage, chained
// items) is computed by eight waves: wave w owns position group w & 3 (64 positions, as before) and CHANNEL HALF w >> 2 -- rows
// 32 (w >> 2) .. + 31 of BOTH 64-channel tiles.  Its accumulators are 2 tiles x (32 channels x 64 positions) x 2 sets = 128
// registers, and waves w and w + 4 share a SIMD (workgroup waves go to SIMDs round robin): while one of them issues a load,
// converts its share of the patch or sits in an LDS round trip of the epilogue, the other one's MFMAs keep the pipe busy.
//   * per step and wave: 1 weight fragment + 2 patch fragments per plane (6 ds_read_b128) feed 6 MFMAs (2 position tiles x 3
//     products); per SIMD that is 12 MFMAs per step as before, per CU 48 fragment reads instead of 32 (the two waves of a
//     position group read the same patch fragments): 192 of the step's 384 cycles of LDS read bandwidth.
//   * staging is split eight ways: a thread converts ONE pixel quad of FOUR channels per stage (waves 0-3: channels 0-3 of the
//     8-channel groups, waves 4-7: channels 4-7; two paired 16-byte loads, four conversion units, eight ds_write_b64 into the
//     halves of the slots) and copies five of the 36 weight chunks of a half-stage (chunk w + 8 m, m < 4; the fifth is one of
//     chunks 32 .. 35 -- waves 4-7 copy them a second time, same bytes to the same place, so that every wave issues the same
//     instructions: no branch in the loop).
//   * epilogue: a wave transposes 32 channels x 32 positions at a time through its 4.5 KB of the dead weight stage (8 x 4.5 KB =
//     the stage exactly) and stores rows of 8 channels x 128 contiguous bytes; bias, residual and the tile statistics as before.
// Same arithmetic per output element as the single-tile and the two-tile kernel -- same products in the same order into the same
// two accumulator sets, same epilogue operations -- and the same statistics: a wave's 64 positions of a channel are reduced by
// an 8-lane butterfly per 32-position pass, which is the first three levels of the 16-lane butterfly of conv_epilogue_fast_finish,
// and the two passes are added as its fourth level adds the halves: the three kernels are BIT-IDENTICAL, output, statistics and
// overflow word (tests/test_conv_bf16x3_gpu.py, tests/test_conv_split_emul.py).
// Launch: conv_f16x2_w8_launch() under the conditions of conv_f16x2_ct2_launch(); EMO_CONV_W8=0 leaves the pairs to the
// one-wave-per-SIMD kernel (A/B).
#pragma once
#include "conv_igemm_bf16x3.h"
#include "conv_split_pair_common.h"

#ifndef EMO_W8_EARLY_LOADS
#define EMO_W8_EARLY_LOADS 1   /* 0: A/B builds (loads of stage cg + 2 in half-stage 1, as the fp16 split) */
#endif
#ifndef EMO_W8_EARLY_STEPS
#define EMO_W8_EARLY_STEPS 4
#endif

typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

template <int TR, int TW, bool UPS>
struct ConvCfgW8 : ConvCfgS<TR, TW, UPS, 2> {
  using Base = ConvCfgS<TR, TW, UPS, 2>;
  static_assert(Base::NWB == 2, "two whole weight stages in LDS");
  static constexpr int NWV = 8;                          // waves per block
  static constexpr int NTH = 64 * NWV;
  // second channel tile: its bias table and its (mean, M2) exchange, behind the first tile's
  static constexpr int OFF_BIAS2_F = Base::OFF_EPI_F;
  static constexpr int OFF_STAT2_F = OFF_BIAS2_F + Base::BM;
  static constexpr int LDS_BYTES = (OFF_STAT2_F + 2 * Base::WGP * Base::BM) * 4;
  // epilogue scratch: per wave [32 channels][32 positions + 4] floats (the + 4 spreads the b128 stores over the banks); the eight
  // of them fill the weight stage buffer the item's last half-stage read
  static constexpr int EPI_ROWF8 = 36;
  static constexpr int EPI_WAVE8 = 32 * EPI_ROWF8;
  static_assert(Base::WSTAGE * 4 >= NWV * EPI_WAVE8, "the epilogue transposes through one weight stage buffer");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(Base::PBUF * 16 >= 8 * 1024, "the dead weight chunks of an item's last step are dumped into a patch buffer");
  static_assert(Base::WSTAGE * 16 == 36 * 1024, "36 chunks of 1 KiB per half-stage");
};

// sum over the 8 lanes of a half DPP row, every lane ends with the total: the first three levels of emo_row16_sum_n
template <int N>
__device__ __forceinline__ void emo_row8_sum_n(float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0xB1, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x4E, 0xf, 0xf, true));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), 0x141, 0xf, 0xf, true));
}

// residual loads of one 32-position pass J of one channel tile for this wave: 32 channels in ROW layout -- lane (g8 = lane >> 3,
// t8 = lane & 7) holds positions 32 J + 4 t8 .. + 3 of channel 8 it + g8 (of the wave's 32).  cot32 = the wave's 32-row tile
// (2 x channel tile + channel half).  RES as in conv_epilogue_fast_issue; volume sizes laundered (see there).
template <int TW, int RES, int J>
__device__ __forceinline__ void conv_w8_res_issue(const ConvArgs& a, floatx4 (&rv)[4], int n, int cot32, int x0, int y0, int z0,
                                                  int wp, int lane) {
  if constexpr (RES != 0) {
    const int g8 = lane >> 3, t8 = lane & 7;
    const unsigned Hr = RES == 2 ? a.Hl >> 1 : a.Hl, Wr = RES == 2 ? a.Wl >> 1 : a.Wl;
    unsigned rvol = (unsigned)a.Dl * Hr * Wr;
    asm volatile("" : "+s"(rvol));
    const float* rbase = a.res + ((long)n * a.Cout + (long)cot32 * 32) * rvol;      // wave-uniform
    const int p = wp * 64 + J * 32 + 4 * t8;
    const int y = y0 + p / TW, x = x0 + p % TW;
    const unsigned rsp = RES == 2 ? ((unsigned)z0 * Hr + (y >> 1)) * Wr + (x >> 1) : ((unsigned)z0 * Hr + y) * Wr + x;
    const unsigned roff = (unsigned)g8 * rvol + rsp;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float* rp = rbase + (roff + (unsigned)(8 * it) * rvol);
      if constexpr (RES == 1) rv[it] = *reinterpret_cast<const floatx4*>(rp);
      else { const float2 r2 = *reinterpret_cast<const float2*>(rp); rv[it] = floatx4{r2.x, r2.y, 0.0f, 0.0f}; }
    }
  }
}

// epilogue of one channel tile for this wave (see the header comment): per element the operations of conv_epilogue_fast_finish in
// its order -- (leading + small accumulator) * out_scale, + bias, + residual -- and its statistics.
//   scratch  this wave's [32][EPI_ROWF8] floats     sbias  the wave's 32 bias entries, entry g8 * 4 + it = channel 8 it + g8
//   st_lds   the tile's [WGP][BM][2] (mean, M2) exchange; the caller combines the position groups behind its closing barrier
// NEXT: the residual loads of the pair's SECOND tile (32-row tile cot32 + 2) are issued from here, pass J of them behind pass J of
// this tile -- into the registers this tile's residual has just left (with both tiles' 64 residual registers live from the top the
// compiler spilled the chained item's raw patch across the epilogue).
// HI: the kernel keeps a second accumulator set (the fp16 SPLIT's small products); false for plain fp16 operands (NPROD = 1).
// issue_next (wave-uniform, with NEXT): false for the half-empty last pair of a layer with an odd number of channel tiles.
template <int TW, int BM, int ROWF, int RES, bool NEXT, bool HI>
__device__ __forceinline__ void conv_w8_epilogue_tile(const ConvArgs& a, floatx16 (&acc_lo)[2], floatx16 (&acc_hi)[2],
                                                      floatx4 (&rv)[2][4], floatx4 (&rvn)[2][4], float* scratch, const float* sbias,
                                                      float* st_lds, int n, int cot32, int x0, int y0, int z0, int wp, int ch,
                                                      int half, int l32, int lane, bool issue_next) {
  const int g8 = lane >> 3, t8 = lane & 7;
  const bool want_stats = a.gn_stats != nullptr;
  const unsigned plane = (unsigned)a.Hl * a.Wl;
  unsigned ovol = (unsigned)a.Dl * plane;
  asm volatile("" : "+s"(ovol));
  float* const obase = a.out + ((long)n * a.Cout + (long)cot32 * 32) * ovol;          // wave-uniform
  const floatx4 b4 = *reinterpret_cast<const floatx4*>(sbias + g8 * 4);
  floatx4 v[2][4];
#pragma unroll
  for (int J = 0; J < 2; ++J) {
    // accumulator layout -> LDS: position tile J, register quad q of lane (half, l32) = channel l32, positions 32 J + 8 q + 4 half .. + 3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      floatx2 c01 = floatx2{emo_acc_read(acc_lo[J][4 * q + 0]), emo_acc_read(acc_lo[J][4 * q + 1])};
      floatx2 c23 = floatx2{emo_acc_read(acc_lo[J][4 * q + 2]), emo_acc_read(acc_lo[J][4 * q + 3])};
      if constexpr (HI) {
        c01 = c01 + floatx2{emo_acc_read(acc_hi[J][4 * q + 0]), emo_acc_read(acc_hi[J][4 * q + 1])};
        c23 = c23 + floatx2{emo_acc_read(acc_hi[J][4 * q + 2]), emo_acc_read(acc_hi[J][4 * q + 3])};
      }
      const floatx2 sc2 = floatx2{a.out_scale, a.out_scale};
      c01 = c01 * sc2;
      c23 = c23 * sc2;
      *reinterpret_cast<floatx4*>(scratch + l32 * ROWF + 8 * q + 4 * half) = floatx4{c01[0], c01[1], c23[0], c23[1]};
    }
    // (LDS operations of one wave execute in order: the row reads below see the stores above, and the next pass's stores come
    // behind these reads)
#pragma unroll
    for (int it = 0; it < 4; ++it) v[J][it] = *reinterpret_cast<const floatx4*>(scratch + (8 * it + g8) * ROWF + 4 * t8);
    const int p = wp * 64 + J * 32 + 4 * t8;
    const int y = y0 + p / TW, x = x0 + p % TW;
    const unsigned off = (unsigned)g8 * ovol + (unsigned)z0 * plane + (unsigned)y * a.Wl + x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const float bs = b4[it];
      const floatx2 bs2 = floatx2{bs, bs};
      floatx2 u01 = floatx2{v[J][it][0], v[J][it][1]} + bs2, u23 = floatx2{v[J][it][2], v[J][it][3]} + bs2;
      if constexpr (RES == 1) { u01 = u01 + floatx2{rv[J][it][0], rv[J][it][1]}; u23 = u23 + floatx2{rv[J][it][2], rv[J][it][3]}; }
      if constexpr (RES == 2) { u01 = u01 + floatx2{rv[J][it][0], rv[J][it][0]}; u23 = u23 + floatx2{rv[J][it][1], rv[J][it][1]}; }
      v[J][it] = floatx4{u01[0], u01[1], u23[0], u23[1]};
      float* const op = obase + (off + (unsigned)(8 * it) * ovol);
      if (EMO_CONV_NT_STORE) __builtin_nontemporal_store(v[J][it], reinterpret_cast<floatx4*>(op));
      else *reinterpret_cast<floatx4*>(op) = v[J][it];
    }
    if constexpr (NEXT) {
      if (issue_next) {
        if (J == 0) conv_w8_res_issue<TW, RES, 0>(a, rvn[0], n, cot32 + 2, x0, y0, z0, wp, lane);
        else conv_w8_res_issue<TW, RES, 1>(a, rvn[1], n, cot32 + 2, x0, y0, z0, wp, lane);
      }
    }
  }
  if (want_stats) {
    // mean over the wave's 64 positions of a channel, M2 centred at it: per 32-position pass an 8-lane butterfly (levels 1 - 3 of
    // the 16-lane one), the passes added as its level 4 adds the row halves -- the bits of conv_epilogue_fast_finish
    constexpr float inv_cnt = 1.0f / 64.0f;
    float s[8], m2[8], mean[4];
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const floatx2 p2 = floatx2{v[J][it][0], v[J][it][2]} + floatx2{v[J][it][1], v[J][it][3]};      // (v0 + v1, v2 + v3)
        s[J * 4 + it] = p2[0] + p2[1];
      }
    emo_row8_sum_n<8>(s);
#pragma unroll
    for (int it = 0; it < 4; ++it) mean[it] = (s[it] + s[4 + it]) * inv_cnt;
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const floatx2 mean2 = floatx2{mean[it], mean[it]};
        const floatx2 d01 = floatx2{v[J][it][0], v[J][it][1]} - mean2, d23 = floatx2{v[J][it][2], v[J][it][3]} - mean2;
        float q = 0.0f;
        q = __fmaf_rn(d01[0], d01[0], q);
        q = __fmaf_rn(d01[1], d01[1], q);
        q = __fmaf_rn(d23[0], d23[0], q);
        q = __fmaf_rn(d23[1], d23[1], q);
        m2[J * 4 + it] = q;
      }
    emo_row8_sum_n<8>(m2);
    if (t8 == 0) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        *reinterpret_cast<float2*>(st_lds + (wp * BM + ch * 32 + 8 * it + g8) * 2) = make_float2(mean[it], m2[it] + m2[4 + it]);
    }
  }
}

// NPROD = 3: the fp16 SPLIT (fp32 results).  NPROD = 1 -- BASELINE configs[4], "fp16 MFMA convs", emo_conv_igemm_f16w8: plain fp16
// operands, ONE product per operand pair, one accumulator set, operands saturate at +-65504 as in conv_igemm_f16.h (no range
// word).  The two operand PLANES of the split layout become two 16-channel K BLOCKS: a stage is 32 input channels, the LDS image,
// the fragment reads, the weight chunks (36 per half-stage) and the schedule are those of the split -- a step is 6 fragment reads
// and FOUR MFMAs (2 k-blocks x 2 position tiles) instead of six, a thread stages its quad of 4 channels of BOTH k-blocks (8
// channel planes).  (A first build kept 16-channel stages with one plane: 18 MFMAs per half-stage between two barriers left the
// latency of the weight chunks and patch loads exposed -- 535-850 TF, a tie with conv_igemm_f16.h: tools/session/r6_call3.sh.)
// A layer with an odd number of channel tiles runs its last tile in a pair whose second half recomputes the same tile and is not
// written (the single-tile kernel has no such mode): the host plans such layers onto conv_igemm_f16.h (pack.f16w8_launch_fits).
template <int TR, int TW, bool UPS, int NPROD = 3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_igemm_f16x2_w8_kernel(const ConvArgs a) {
  using Cfg = ConvCfgW8<TR, TW, UPS>;
  using opx8 = halfx8;
  static_assert(NPROD == 3 || NPROD == 1, "the split's three products, or the leading one alone");
  constexpr int NPL = 2;                                 // operand planes (NPROD = 3) / 16-channel k-blocks of a stage (NPROD = 1)
  constexpr int NCHK = 36;                               // 1 KiB weight chunks per half-stage
  constexpr int NDM = 5;                                 // chunk copies per wave and half-stage
  constexpr int NKB = NPROD == 1 ? 2 : 1;                // k-blocks a thread stages
  constexpr int KCE = 16 * NKB;                          // input channels per stage
  constexpr int BM = Cfg::BM, TP = Cfg::TP, WGP = Cfg::WGP, KC = Cfg::KC, NTH = Cfg::NTH;
  constexpr int PR = Cfg::PR, NQ = Cfg::NQ, NQ1 = Cfg::NQ1, SUB = Cfg::SUB, CHS = Cfg::CHS, QPG = Cfg::QPG;
  constexpr int NHQ = Cfg::NHQ, TWS = Cfg::TWS, WPLANE = Cfg::WPLANE, WROW = Cfg::WROW, PPL = Cfg::PPL, PBUF = Cfg::PBUF;
  static_assert(TP == 2 && BM == 64, "wave tile: 32 channels x 64 positions of each of the two channel tiles");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int wp = wave & 3;                               // position group
  const int ch = wave >> 2;                              // channel half of both tiles; also the 4-channel half this wave stages
  const int p0 = wp * TP * 32;
  float sat_m = 0.0f;                                    // largest |scaled staged value| this thread has seen

  // ---- constants of the launch and of the thread (conv_igemm_f16x2_ct2.h) ----
  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const bool has_affine = a.scale != nullptr;
  const int epi_mode = __builtin_amdgcn_readfirstlane(a.res == nullptr ? 0 : (a.res_ups ? 2 : 1));
  const float in_scale = a.in_scale;
  const int padD = a.KD >> 1;
  constexpr float CLAMP_HI = 65504.0f;
  const float clamp_lo = a.relu_in ? 0.0f : -CLAMP_HI;
  const int nst = a.n_cchunks * a.KD;                    // stages of an item (no K split)
  const int nptiles = a.tiles_x * a.tiles_y * a.tiles_z;
  const int n_cot_all = a.Cout / BM;                     // channel tiles of the layer (an odd count only with NPROD = 1)

  // ---- staging map: the 256 threads of a channel half (waves 0-3 / 4-7) are two 8-channel groups of 128 lanes, mapped onto the
  //      interior quads and halo pixels of the patch exactly as the 256 threads of conv_igemm_bf16x3.h; a thread stages the four
  //      channels 4 ch .. 4 ch + 3 of its group's eight: bytes 8 ch .. 8 ch + 7 of the 16-byte slots ----
  const int q_u = tid % QPG;
  const int q_g = __builtin_amdgcn_readfirstlane((tid / QPG) & 1);
  const bool is_quad = q_u < PR * NQ;
  const int hq = q_u - PR * NQ;
  const bool is_halo = !is_quad && hq < NHQ;
  const int h_side = hq & 1;
  const int q_r = is_quad ? q_u / NQ : (is_halo ? hq >> 1 : 0);
  const int q_c = is_quad ? q_u - q_r * NQ : 0;
  int q_slb[4];                                           // byte offsets of the lane's four staging half-slots inside a patch buffer
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int dump = q_g * CHS + i * SUB + PR * NQ1 + (q_u & 3);
    const int own = is_quad ? q_g * CHS + i * SUB + q_r * NQ1 + q_c : q_g * CHS + h_side * SUB + q_r * NQ1 + NQ;
    q_slb[i] = ((is_quad || (is_halo && i == (h_side ? 0 : 3))) ? own : dump) * 16 + ch * 8;
  }

  floatx16 acc_lo[2][TP], acc_hi[2][TP];                 // [channel tile of the pair][position tile]: 128 accumulation registers

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
// byte address of the packed kernel rows of (channel tile c_, stage k_): NCHK contiguous chunks of 1 KiB (a tile past the layer's
// last one -- the second half of an odd last pair -- reads the last tile's)
#define EMO_W_WPTR(c_, k_) (reinterpret_cast<const char*>(a.wpk) + \
                            (long)(((c_) < n_cot_all ? (c_) : n_cot_all - 1) * nst + (k_)) * (NCHK * 1024))

  // LDS byte offsets of the lane's operands (conv_igemm_f16x2_ct2.h): the wave's weight rows are ch * 32 + l32 of the tile's 64
  const int a_off = (half * BM + ch * 32 + l32) * 16;
  EMO_P_DECLARE_B_OFF()

  const char* const lds_c = reinterpret_cast<const char*>(smem);
  char* const lds_w = reinterpret_cast<char*>(smem);
  opx8 fa_[2][NPL], fb_[2][NPL][TP];         // [register set: this step / the next][plane]([position tile])
// (wbase_: slots, compile-time; pbyte_: byte offset of the patch buffer, run-time)
// (a kernel row of a stage buffer holds NPL planes: rows are NPL * WPLANE slots apart)
#define EMO_W_LOAD_FRAGS_PLANE(set_, pl_, wbase_, pbyte_, r_, s_)                                      \
  {                                                                                                   \
    fa_[set_][pl_] = *reinterpret_cast<const opx8*>(lds_c + a_off + ((wbase_) + (pl_) * WPLANE + (s_) * 2 * BM) * 16); \
    _Pragma("unroll") for (int j = 0; j < TP; ++j)                                                    \
      fb_[set_][pl_][j] = *reinterpret_cast<const opx8*>(lds_c + (EMO_P_B_OFF(j, r_, s_) + (pbyte_)) + ((pl_) * PPL) * 16); \
  }

  float* const sct = smem + Cfg::OFF_SCT * 4;
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);
  const unsigned lane16 = (unsigned)lane * 16u;

  // raw patch registers: ONE buffer of four channel planes -- converted during half-stage 0, reloaded during half-stage 1
  floatx4 qv[NKB][4];
  float q_lo[NKB], q_hi[NKB];
  int q_tix[NKB];
  floatx4 q_sc[NKB], q_sh[NKB];
  emo_intx4 xrs = emo_raw_buffer(a.x);
  unsigned usoff[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) usoff[u] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(4 * ch + u) * (unsigned)DHW * 4u * (EMO_CT2_X_CONST ? 0u : 1u)));

  int n_ci0, n_zu;
  bool n_zv;
  int ld_stage, ld_cc, ld_kd;   // the stage whose patch is being loaded, stepped (no division in the loop)
#define EMO_W_SET_STAGE_VARS()                                                                        \
  {                                                                                                   \
    n_ci0 = ld_cc * KCE;                                                                              \
    n_zu = lq_z0 + ld_kd - padD;                                                                      \
    n_zv = (unsigned)n_zu < (unsigned)a.D;                                                            \
  }
  unsigned q_vo[NKB];
#define EMO_W_ISSUE_BEGIN()                                                                           \
  {                                                                                                   \
    _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb) {                                              \
      const int c0_ = n_ci0 + 16 * kb + q_g * 8;                                                      \
      const bool cv_ = c0_ < a.Cin;                                                                   \
      const int cs_ = cv_ ? c0_ : 0;                                                                  \
      const bool keep_ = lq_ok && cv_ && n_zv;                                                        \
      q_lo[kb] = keep_ ? clamp_lo : 0.0f;                                                             \
      q_hi[kb] = keep_ ? CLAMP_HI : 0.0f;                                                             \
      q_vo[kb] = lq_off + (EMO_CT2_X_CONST ? 0u : ((unsigned)cs_ * (unsigned)DHW + (unsigned)((n_zv ? n_zu : 0) * HW)) * 4u);  \
      q_tix[kb] = ((has_affine ? cs_ : (cs_ & (Cfg::SCT - 1))) >> 2) + ch;                            \
    }                                                                                                 \
  }
// paired load v_ = 0 .. 2 NKB - 1 of a stage: channel planes 2 (v_ & 1), 2 (v_ & 1) + 1 of k-block v_ >> 1
#define EMO_W_ISSUE_LOADS(v_)                                                                         \
  { emo_bload4x2_pinned(xrs, q_vo[(v_) >> 1], usoff[2 * ((v_) & 1)], usoff[2 * ((v_) & 1) + 1], qv[(v_) >> 1][2 * ((v_) & 1)],  \
                        qv[(v_) >> 1][2 * ((v_) & 1) + 1]); }
#define EMO_W_TABLE()                                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb) {                                              \
      const floatx4* t4_ = reinterpret_cast<const floatx4*>(sct) + q_tix[kb];                         \
      q_sc[kb] = t4_[0]; q_sh[kb] = t4_[Cfg::SCT / 4];                                                \
    }                                                                                                 \
  }
#define EMO_W_TOUCH_QUAD() { _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb) _Pragma("unroll") for (int u = 0; u < 4; ++u) emo_touch4(qv[kb][u]); }
// conversion of the lane's four channels (of k-block kb_) of pixel i_ (conv_igemm_bf16x3.h, SPLIT = 2); pbyte_: byte offset of the
// target patch buffer.  NPROD = 3: both planes of the one k-block; NPROD = 1: the fp16 value into plane kb_
#define EMO_W_CONV_UNIT(pbyte_, i_, kb_)                                                              \
  {                                                                                                   \
    float t_[4];                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                     \
      t_[k] = __fmaf_rn(qv[kb_][k][i_], q_sc[kb_][k], q_sh[kb_][k]);                                  \
    if constexpr (NPROD == 3) {                                                                       \
      sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[0])), __builtin_fabsf(t_[1])); \
      sat_m = __builtin_fmaxf(__builtin_fmaxf(sat_m, __builtin_fabsf(t_[2])), __builtin_fabsf(t_[3])); \
    }                                                                                                 \
    halfx4 cvh_, cvm_;                                                                                \
    _Pragma("unroll") for (int k = 0; k < 4; k += 2)                                                  \
      emo_split_f16x2_pair(__builtin_amdgcn_fmed3f(t_[k], q_lo[kb_], q_hi[kb_]),                      \
                           __builtin_amdgcn_fmed3f(t_[k + 1], q_lo[kb_], q_hi[kb_]), cvh_, cvm_, k);  \
    char* d_ = lds_w + (q_slb[i_] + (pbyte_));                                                        \
    if constexpr (NPROD == 3) {                                                                       \
      *reinterpret_cast<halfx4*>(d_) = cvh_;                                                          \
      *reinterpret_cast<halfx4*>(d_ + PPL * 16) = cvm_;                                               \
    } else {                                                                                          \
      *reinterpret_cast<halfx4*>(d_ + (kb_) * PPL * 16) = cvh_;                                       \
    }                                                                                                 \
  }
// chunk m_ = 0 .. NDM - 1 of this wave's share of a half-stage's NCHK weight chunks (header comment), to the LDS byte address
// dst_ + chunk KiB
#define EMO_W_CHUNK_OF(m_) ((m_) < 4 ? wave + 8 * (m_) : 32 + wp)
#define EMO_W_DMA_CHUNK(ptr_, dst_, m_)                                                               \
  {                                                                                                   \
    const int c_ = EMO_W_CHUNK_OF(m_);                                                                \
    emo_dma16_pinned_s(EMO_CT2_W_CONST ? reinterpret_cast<const char*>(a.wpk) : (ptr_) + c_ * 1024, lane16, \
                       (dst_) + (unsigned)(c_ * 1024));   /* (EMO_CT2_W_CONST / _X_CONST: measurement builds, conv_igemm_f16x2_ct2.h) */ \
  }
#define EMO_W_WBUF(wb_) (smem_lds + (unsigned)((Cfg::OFF_W + (wb_) * Cfg::WSTAGE) * 16))

  // the partial products, smallest first: (weight plane, patch plane); the last one is the leading product
  // (NPROD = 1: "products" = the two k-blocks of a stage, weight block x patch block, both into the one accumulator set)
  constexpr int NMM = NPROD == 1 ? 2 : 3;                // MFMAs per position tile and step
  constexpr int PA3[3] = {NPROD == 1 ? 0 : 1, NPROD == 1 ? 1 : 0, 0}, PB3[3] = {0, 1, 0};
  constexpr int WROWU = NPL * WPLANE;                    // slots between the kernel rows of a stage buffer
  constexpr int NTE = Cfg::SCT / NTH;
  float te_sc[NTE], te_sh[NTE], te_b = 0.0f;
// bias table entry of channel t2_ of a tile (conv_w8_epilogue_tile: entry ch * 32 + g8 * 4 + it = channel ch * 32 + 8 it + g8)
#define EMO_W_BIAS_SLOT(t2_) (((t2_) >> 5) * 32 + ((t2_) & 7) * 4 + (((t2_) & 31) >> 3))

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
      asm volatile("" : "=v"(fa_[st_][pl]));
#pragma unroll
      for (int j = 0; j < TP; ++j) asm volatile("" : "=v"(fb_[st_][pl][j]));
    }
  // (the scale / shift entries of the patch under conversion are NOT carried from item to item: a chained item reads them again --
  // two LDS reads -- instead of holding 8 / 16 registers through the epilogue, where the plain-fp16 build spilled landed residual
  // values to scratch behind a vmcnt(0) each: epilogue 28 k cycles per item against 11.5 k, profiles/r6_conv_phase_timing_f16w8_first.jsonl)
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) { asm volatile("" : "=v"(q_sc[kb])); asm volatile("" : "=v"(q_sh[kb])); }
  if (EMO_S_CHAIN && chained_in) {
    // P[pp] holds the converted patch of stage 0, W[0] the kernel rows of (c0, stage 0), qv the landed loads of stage 1, q_sc /
    // q_sh its table entries; the tables are the sample's.  What is left: bias entries, the first chunk of (c0 + 1, 0)
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    if (tid < 2 * BM) {
      const int t2_ = tid & (BM - 1);
      smem[(tid < BM ? Cfg::OFF_BIAS_F : Cfg::OFF_BIAS2_F) + EMO_W_BIAS_SLOT(t2_)] = te_b;
    }
    const char* const w1_ = EMO_W_WPTR(it_cotile + 1, 0);
    EMO_W_DMA_CHUNK(w1_, EMO_W_WBUF(1), 0)
    EMO_W_TABLE()
    EMO_P_BARRIER(1);
  } else {
    // ---- full prologue: tables, the chunks of (c0, stage 0), the patch of stage 0 converted into P[0], the loads of stage 1 ----
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "=v"(qv[kb][u]));
    xrs = emo_raw_buffer(a.x + (long)it_n * a.Cin * DHW);
    EMO_P_CURSOR_OF(it_, lq_ok, lq_off)
    lq_z0 = it_z0;
#pragma unroll
    for (int k = 0; k < NTE; ++k) {
      const int c = tid + NTH * k;
      const bool real = has_affine && c < a.Cin;
      te_sc[k] = real ? a.scale[(long)it_n * a.Cin + c] : 1.0f;
      te_sh[k] = real ? a.shift[(long)it_n * a.Cin + c] : 0.0f;
    }
    if (tid < 2 * BM && a.bias != nullptr) {
      const int co_ = it_cotile * BM + tid;
      te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];
    }
    {
      const char* const w0_ = EMO_W_WPTR(it_cotile, 0);
#pragma unroll
      for (int m = 0; m < NDM; ++m) EMO_W_DMA_CHUNK(w0_, EMO_W_WBUF(0), m)
    }
    ld_stage = 0; ld_cc = 0; ld_kd = 0;
    EMO_W_SET_STAGE_VARS()
    EMO_W_ISSUE_BEGIN()
#pragma unroll
    for (int v = 0; v < 2 * NKB; ++v) EMO_W_ISSUE_LOADS(v)
#pragma unroll
    for (int k = 0; k < NTE; ++k) {       // (without an affine the index wraps at SCT: identity entries)
      const int c = tid + NTH * k;
      if (c < min(a.Cin, Cfg::SCT)) {
        sct[c] = te_sc[k] * in_scale;
        sct[Cfg::SCT + c] = te_sh[k] * in_scale;
      }
    }
    if (tid < 2 * BM) {
      const int t2_ = tid & (BM - 1);
      smem[(tid < BM ? Cfg::OFF_BIAS_F : Cfg::OFF_BIAS2_F) + EMO_W_BIAS_SLOT(t2_)] = te_b;
    }
    EMO_P_WAIT(0);
    EMO_W_TOUCH_QUAD()
    __syncthreads();   // scale / shift tables visible
    EMO_W_TABLE()
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) EMO_W_CONV_UNIT(Cfg::OFF_P * 16, i, kb)
    if (nst > 1) {                        // (one-stage item: the same patch again, a dead re-stage)
      ++ld_stage;
      if (++ld_kd == a.KD) { ld_kd = 0; ++ld_cc; }
    }
    EMO_W_SET_STAGE_VARS()
    EMO_W_ISSUE_BEGIN()
#pragma unroll
    for (int v = 0; v < 2 * NKB; ++v) EMO_W_ISSUE_LOADS(v)
    EMO_W_TABLE()
    {
      const char* const w1_ = EMO_W_WPTR(it_cotile + 1, 0);
      EMO_W_DMA_CHUNK(w1_, EMO_W_WBUF(1), 0)
    }
    EMO_P_BARRIER(0);                    // (P[0] visible, W[0] and the loads of stage 1 landed)
    EMO_W_TOUCH_QUAD()
    pp = 0;
  }

  // ---- K loop: one stage = two half-stages (conv_igemm_f16x2_ct2.h, header comment there) ----
  // EARLY (plain fp16 operands): a half-stage is a third of the split's matrix work, so the raw patch loads of stage cg + 2 --
  // issued in the first steps of half-stage 1 and awaited at its end -- had five to eight steps (1.5-2 k cycles) to come back:
  // less than a loaded L2 miss.  The conversion of stage cg + 1 (12 VALU per unit here) runs two units per step in steps 0-3 of
  // half-stage 0, the loads go out in its steps 4-7 and fly across its barrier: 10-13 steps of cover, same registers, same
  // invariants at the stage boundary (prologue, chaining and epilogue unchanged)
  constexpr bool EARLY = NPROD == 1 && EMO_W8_EARLY_LOADS;
  constexpr int ESTEPS = EMO_W8_EARLY_STEPS;              // steps the conversion takes (4: one pixel per step, 2: two)
  EMO_S_STAMP(1)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_lo[c][j][r] = 0.0f; acc_hi[c][j][r] = 0.0f; }
  {
    const int pb0_ = (Cfg::OFF_P + pp * PBUF) * 16;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) EMO_W_LOAD_FRAGS_PLANE(0, pl, Cfg::OFF_W, pb0_, 0, 0)      // (first half-stage: W[0], P[pp])
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
      const char* const dma_ptr = EMO_W_WPTR(c1_, k1_);
      const char* const dma_ptr2 = EMO_W_WPTR(c2_, k2_);
      if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int gs = 0; gs < 9; ++gs) {
        const int fcur = (h * 9 + gs) & 1, fnxt = fcur ^ 1;
        if (gs == 8) {
          // (EARLY: the patch loads issued in this half-stage 0 -- 2 NKB pairs, younger than its weight chunks -- stay in flight)
          if (EARLY && h == 0) { EMO_P_BARRIER(4 * NKB); } else { EMO_P_BARRIER(0); }
        }
        if (gs == 8 && h == 1) EMO_W_TOUCH_QUAD()          // (the loads of stage cg + 2 have landed behind the barrier)
        if (EARLY ? (gs == ESTEPS && h == 0) : (gs == 0 && h == 1)) {
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
          EMO_W_SET_STAGE_VARS()
          EMO_W_ISSUE_BEGIN()
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const int rn = gs < 8 ? (gs + 1) / 3 : 0, sn = gs < 8 ? (gs + 1) % 3 : 0;
          const int wbn = Cfg::OFF_W + (gs < 8 ? h : h ^ 1) * Cfg::WSTAGE + rn * WROWU;
          const int pbn = (gs == 8 && h == 1) ? pnxt_b : pcur_b;       // (half-stage 1 reads the same patch as half-stage 0)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            EMO_W_LOAD_FRAGS_PLANE(fnxt, pl, wbn, pbn, rn, sn)
            if (gs == 8 && pl == 0) {
              // chunk 0 of half-stage t + 2 into W[h] (free behind the barrier).  Last stage, h = 1: W[1] is the epilogue's
              // scratch -- the chunk (dead, or the next item's, which its short prologue fetches) goes to the idle patch buffer
              const unsigned dst_ = (h == 1 && last_) ? smem_lds + (unsigned)pcur_b : EMO_W_WBUF(h);
              EMO_W_DMA_CHUNK(dma_ptr2, dst_, 0)
            }
            // chunks 1 .. NDM - 1 of the rows of half-stage t + 1 into W[h ^ 1], one per step
            if (gs < NDM - 1 && pl == 0) EMO_W_DMA_CHUNK(dma_ptr, EMO_W_WBUF(h ^ 1), 1 + gs)
            if (!EARLY && h == 1 && gs < 2 * NKB && pl == NPL - 1) EMO_W_ISSUE_LOADS(gs)
            if (EARLY && h == 0 && gs >= ESTEPS && gs < ESTEPS + 2 * NKB && pl == NPL - 1) EMO_W_ISSUE_LOADS(gs - ESTEPS)
          }
        }
        // the patch of stage cg + 1: one conversion unit per step (four channels of one pixel, of one k-block)
        if (!EARLY && h == 0 && gs < 4 * NKB) EMO_W_CONV_UNIT(pnxt_b, gs / NKB, gs % NKB)
        if (EARLY && h == 0 && gs < ESTEPS) {              // (all of it in the first steps: the registers are free for the loads)
#pragma unroll
          for (int i = gs * (4 / ESTEPS); i < (gs + 1) * (4 / ESTEPS); ++i)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) EMO_W_CONV_UNIT(pnxt_b, i, kb)
        }
        if (h == 1 && gs == 7) { EMO_W_TABLE() }           // (what the next stage's units convert with)
#pragma unroll
        for (int p = 0; p < NMM; ++p) {
          const int pa = PA3[p], pb = PB3[p];
#pragma unroll
          for (int j = 0; j < TP; ++j) {
            floatx16& acc_ = (NPROD == 1 || (pa == 0 && pb == 0)) ? acc_lo[h][j] : acc_hi[h][j];
            acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[fcur][pb][j], fa_[fcur][pa], acc_, 0, 0, 0);
          }
        }
        if (EMO_S_PIN) {
          // the step's six fragment reads and its other work spread between its MFMAs (six; plain fp16 operands: four, the first
          // two with two reads each)
          if constexpr (NPROD == 1) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
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
    float* const scratch = smem + (Cfg::OFF_W + Cfg::WSTAGE) * 4 + wave * Cfg::EPI_WAVE8;
    const int ep_n = it_n, ep_cot32 = 2 * it_cotile + ch, ep_x0 = it_x0, ep_y0 = it_y0, ep_z0 = it_z0;
    const bool has_t1 = NPROD == 3 || it_cotile + 1 < n_cot_all;      // (false: the half-empty last pair of an odd tile count)
    EMO_P_WAIT(0);
    EMO_S_STAMP(5)
    __syncthreads();
    EMO_S_STAMP(6)
    if (EMO_S_CHAIN && chain_out && tid < 2 * BM && a.bias != nullptr) {   // the next item's bias entries
      const int co_ = nx_cotile * BM + tid;
      te_b = a.bias[co_ < a.Cout ? co_ : a.Cout - 1];
    }
// both tiles of the pair; the second tile's residual loads go out from inside the first tile's epilogue (conv_w8_epilogue_tile)
#define EMO_W_EPI(RES_)                                                                                                           \
    {                                                                                                                              \
      floatx4 rv_[2][4], rvn_[2][4];                                                                                               \
      conv_w8_res_issue<TW, RES_, 0>(a, rv_[0], ep_n, ep_cot32, ep_x0, ep_y0, ep_z0, wp, lane);                                    \
      conv_w8_res_issue<TW, RES_, 1>(a, rv_[1], ep_n, ep_cot32, ep_x0, ep_y0, ep_z0, wp, lane);                                    \
      EMO_S_STAMP(7)                                                                                                               \
      conv_w8_epilogue_tile<TW, BM, Cfg::EPI_ROWF8, RES_, NPROD == 3, NPROD == 3>(                                                 \
          a, acc_lo[0], acc_hi[0], rv_, rvn_, scratch, smem + Cfg::OFF_BIAS_F + ch * 32, smem + Cfg::OFF_STAT_F, ep_n, ep_cot32,    \
          ep_x0, ep_y0, ep_z0, wp, ch, half, l32, lane, has_t1);                                                                   \
      EMO_S_STAMP(8)                                                                                                               \
      if (has_t1) {                                                                                                                \
        /* (plain fp16 operands: more raw patch registers live across the epilogue -- the second tile's residual goes out here) */ \
        if constexpr (NPROD == 1) {                                                                                                \
          conv_w8_res_issue<TW, RES_, 0>(a, rvn_[0], ep_n, ep_cot32 + 2, ep_x0, ep_y0, ep_z0, wp, lane);                           \
          conv_w8_res_issue<TW, RES_, 1>(a, rvn_[1], ep_n, ep_cot32 + 2, ep_x0, ep_y0, ep_z0, wp, lane);                           \
        }                                                                                                                          \
        conv_w8_epilogue_tile<TW, BM, Cfg::EPI_ROWF8, RES_, false, NPROD == 3>(                                                    \
            a, acc_lo[1], acc_hi[1], rvn_, rv_, scratch, smem + Cfg::OFF_BIAS2_F + ch * 32, smem + Cfg::OFF_STAT2_F, ep_n,          \
            ep_cot32 + 2, ep_x0, ep_y0, ep_z0, wp, ch, half, l32, lane, false);                                                    \
      }                                                                                                                            \
      EMO_S_STAMP(9)                                                                                                               \
    }
    if (epi_mode == 1) EMO_W_EPI(1)
    else if (epi_mode == 2) EMO_W_EPI(2)
    else EMO_W_EPI(0)
#undef EMO_W_EPI
  }
  if (NPROD == 3 && a.sat_flag != nullptr && sat_m > 65504.0f) *a.sat_flag = 1;   // (every writer stores the same value)
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
  // tile statistics, second half (conv_igemm_f16x2_ct2.h): thread c of the first 128 combines the four position groups' (mean, M2)
  // of channel c with the equal-count update
  EMO_P_COMBINE_STATS(Cfg::OFF_STAT_F, Cfg::OFF_STAT2_F, it_cotile * BM + tid < a.Cout)
  chained_in = chain_out;
  }
#undef EMO_W_WPTR
#undef EMO_W_LOAD_FRAGS_PLANE
#undef EMO_W_SET_STAGE_VARS
#undef EMO_W_ISSUE_BEGIN
#undef EMO_W_ISSUE_LOADS
#undef EMO_W_TABLE
#undef EMO_W_TOUCH_QUAD
#undef EMO_W_CONV_UNIT
#undef EMO_W_CHUNK_OF
#undef EMO_W_DMA_CHUNK
#undef EMO_W_WBUF
#undef EMO_W_BIAS_SLOT
}

// Launches the channel-tile PAIRS of the layer on conv_igemm_f16x2_w8_kernel under the conditions of conv_f16x2_ct2_launch (same
// contract: *rest_cot0 = the first channel tile NOT covered; 0: nothing was launched).
// NPROD = 3 (fp16 split): only with EMO_CONV_W8=1 -- measured on one box, A B A B, the eight-wave kernel needs 4-7 % fewer cycles
// per item than the four-wave one and the chip answers with a clock 8-10 % lower: the same frames per second (DESIGN.md
// section 3.0b, profiles/r6_bench_w8_ab_same_box.jsonl); the four-wave kernel stays the default.
// NPROD = 1 (plain fp16 operands): the whole layer, an odd last tile in a half-empty pair; EMO_F16_W8=0 turns it off.
template <int TR, int TW, bool UPS, int NPROD = 3>
int conv_f16x2_w8_launch(ConvArgs a, hipStream_t s, int* rest_cot0) {
  using Cfg = ConvCfgW8<TR, TW, UPS>;
  *rest_cot0 = 0;
  const char* const e_on = getenv(NPROD == 3 ? "EMO_CONV_W8" : "EMO_F16_W8");
  const char* const e_ct2 = getenv("EMO_CONV_CT2");
  const char* const e_min = getenv("EMO_CONV_CT2_MIN_ITEMS");
  if (NPROD == 3 && (!(e_on && atoi(e_on) == 1) || (e_ct2 && atoi(e_ct2) == 0))) return EMO_OK;
  if (NPROD == 1 && e_on && atoi(e_on) == 0) return EMO_OK;
  if (a.ksplit != 1 || a.run_if != nullptr) return EMO_OK;
  if (a.act != EMO_ACT_NONE || a.Cout % Cfg::BM != 0 || (a.Wl & 3) != 0 || (long)a.Dl * a.Hl * a.Wl > (1l << 23) ||
      (reinterpret_cast<unsigned long long>(a.out) & 15ull) != 0 ||
      (a.res != nullptr && (reinterpret_cast<unsigned long long>(a.res) & (a.res_ups ? 7ull : 15ull)) != 0)) return EMO_OK;
  if (a.Wl % TW || a.Hl % TR || a.Cin % 8) return EMO_OK;
  if (a.scale && a.Cin > Cfg::SCT) return EMO_OK;
  if ((unsigned long long)a.Cin * a.D * a.H * a.W * 4ull >= (1ull << 32)) return EMO_OK;
  if ((reinterpret_cast<unsigned long long>(a.x) & 15ull) || (a.W & 3)) return EMO_OK;
  const int cot_all = (a.Cout + Cfg::BM - 1) / Cfg::BM;
  if (a.cot_end != 0 && (NPROD != 1 || a.cot_end < 2 || a.cot_end > cot_all || (a.cot_end & 1))) return EMO_ERR_BAD_ARG;
  const int cot = a.cot_end != 0 ? a.cot_end : cot_all;          // (cot_end: whole pairs only, the rest is the caller's)
  const int pairs = NPROD == 1 ? (cot + 1) / 2 : cot / 2;
  const long nt = (long)(a.Wl / TW) * (a.Hl / TR) * a.Dl;
  const int ncu = emo_cu_count();
  const long min_items = e_min ? atol(e_min) : 2l * ncu;
  if (pairs < 1 || nt > 0x7fffffffL || a.N > 65535 || nt * pairs * a.N > 0x7fffffffL || nt * pairs * a.N < min_items) return EMO_OK;
  auto kern = conv_igemm_f16x2_w8_kernel<TR, TW, UPS, NPROD>;
  const int rc = emo_raise_dynamic_lds(kern);
  if (rc != EMO_OK) return rc;
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl;
  constexpr int KCE = NPROD == 1 ? 32 : Cfg::KC;   // (plain fp16 operands: 32 input channels per stage)
  a.n_cchunks = (a.Cin + KCE - 1) / KCE;
  a.stages_per_split = a.n_cchunks * a.KD;
  a.partial = nullptr;
  a.cot0 = 0;
  a.n_cotiles = pairs;                         // (pairs: EMO_P_DECODE)
  a.n_work = (int)(nt * pairs * a.N);
  const int grid = a.n_work > ncu ? ncu : a.n_work;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NTH), (size_t)Cfg::LDS_BYTES, s, a);
  *rest_cot0 = NPROD == 1 ? cot : 2 * pairs;
  return emo_launch_status();
}
