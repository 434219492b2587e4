// The stage machinery that the two-tile kernels of the fp16 split share -- conv_igemm_f16x2_ct2.h (four waves, one per SIMD) and
// conv_igemm_f16x2_w8.h (eight waves, two per SIMD; also the plain-fp16 mode) -- factored out of the two files (round 5's verdict:
// three hand-scheduled copies of one schedule mean every fix lands three times).  These are MACROS over the kernels' local names
// (a, nptiles, TW, TR, UPS, TWS, NQ, NQ1, SUB, CHS, BM, TP, WGP, q_r, q_c, is_quad, is_halo, h_side, half, l32, p0, smem, tid, ...):
// textual sharing, so both kernels compile to exactly the code they had.  What differs between them -- the staging of the patch
// (8 resp. 4 channels per thread), the weight pieces (9 per wave resp. 5 chunks), the fragment tiles and the epilogue -- stays in
// their own files.
#pragma once

// measurement builds: 1 = every weight piece re-reads the FIRST KiB of the packed weights (resident in the CU's L1: WRONG
// results) -- what the weight stream out of the L2 costs a power-managed chip (both paired kernels;
// tools/session/r6_call16.sh, r6_call19.sh)
#ifndef EMO_CT2_W_CONST
#define EMO_CT2_W_CONST 0
#endif
// ... and 1 = every stage of an item re-reads the patch of input channel 0 (L1 / L2 hits after the first stage: WRONG results)
#ifndef EMO_CT2_X_CONST
#define EMO_CT2_X_CONST 0
#endif


// work item l -> (sample, position tile, channel-tile PAIR): XCD-contiguous order, pair fastest.  Every result through
// readfirstlane (conv_igemm_bf16x3.h); n_cotiles counts PAIRS here, cot0 is the first tile of the launch
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
    P_##cotile = __builtin_amdgcn_readfirstlane(a.cot0 + 2 * cot_);                                   \
    P_##n = __builtin_amdgcn_readfirstlane(n_);                                                       \
    P_##x0 = __builtin_amdgcn_readfirstlane(tx_ * TW);                                                \
    P_##y0 = __builtin_amdgcn_readfirstlane(ty_ * TR);                                                \
    P_##z0 = __builtin_amdgcn_readfirstlane(bx_);                                                     \
  }

// the lane's 16-byte patch load for the tile of item P_ (its quad, or the aligned quad that contains its halo pixel) and whether
// it lies inside the image
#define EMO_P_CURSOR_OF(P_, ok_, off_)                                                                \
  {                                                                                                   \
    const int x0s_ = UPS ? P_##x0 >> 1 : P_##x0, y0s_ = UPS ? P_##y0 >> 1 : P_##y0;                   \
    const int q_y_ = y0s_ - 1 + q_r;                                                                  \
    const int q_x_ = is_quad ? x0s_ + 4 * q_c : (h_side ? x0s_ + TWS : x0s_ - 4);                     \
    ok_ = (is_quad || is_halo) && (unsigned)q_y_ < (unsigned)a.H && q_x_ >= 0 && q_x_ < a.W;          \
    off_ = ok_ ? (unsigned)(q_y_ * a.W + q_x_) * 4u : 0u;                                             \
  }

// LDS byte offsets of the lane's patch fragments: b_off[position tile][kernel-row class][tap column] (conv_igemm_bf16x3.h)
#define EMO_P_DECLARE_B_OFF()                                                                         \
  constexpr int NBR = UPS ? 2 : 1;                                                                    \
  int b_off[TP][NBR][3];                                                                              \
  _Pragma("unroll") for (int j = 0; j < TP; ++j) {                                                    \
    const int p = p0 + j * 32 + l32;                                                                  \
    const int col = p % TW, row = p / TW;                                                             \
    _Pragma("unroll") for (int r = 0; r < NBR; ++r)                                                   \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                                   \
      const int pr = UPS ? ((row + r - 1 + 2) >> 1) - 1 + 1 : row + r;                                \
      const int pc = UPS ? ((col + s - 1 + 2) >> 1) - 1 : col + s - 1;                                \
      const int slot = pc < 0 ? pr * NQ1 + NQ : (pc >= TWS ? SUB + pr * NQ1 + NQ : (pc & 3) * SUB + pr * NQ1 + (pc >> 2)); \
      b_off[j][r][s] = (half * CHS + slot) * 16;                                                      \
    }                                                                                                 \
  }
#define EMO_P_B_OFF(j_, r_, s_) (UPS ? ((r_) == 2 ? b_off[j_][0][s_] + NQ1 * 16 : b_off[j_][(r_) < NBR ? (r_) : 0][s_]) \
                                     : b_off[j_][0][s_] + (r_) * NQ1 * 16)

#define EMO_P_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define EMO_P_BARRIER(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")

// tile statistics, second half (conv_epilogue_rows_stats, for both tiles of the pair at once, behind the barrier that ends the
// item): thread c of the first 2 BM combines the WGP position groups' (mean, M2) of channel c with the equal-count update.  The
// exchange areas are next written by the NEXT item's epilogue, a K loop away
// (real_: false for the unwritten second half of a half-empty last pair -- conv_igemm_f16x2_w8.h with plain fp16 operands only)
#define EMO_P_COMBINE_STATS(st1_f_, st2_f_, real_)                                                    \
  if (a.gn_stats != nullptr && tid < 2 * BM && (real_)) {                                             \
    const int c_ = tid & (BM - 1);                                                                    \
    const float* const st_ = smem + (tid < BM ? (st1_f_) : (st2_f_));                                 \
    float mean = 0.0f, m2 = 0.0f;                                                                     \
    _Pragma("unroll") for (int w = 0; w < WGP; ++w) mean += st_[(w * BM + c_) * 2 + 0];               \
    mean *= 1.0f / (float)WGP;                                                                        \
    _Pragma("unroll") for (int w = 0; w < WGP; ++w) {                                                 \
      const float d = st_[(w * BM + c_) * 2 + 0] - mean;                                              \
      m2 += st_[(w * BM + c_) * 2 + 1] + (float)(TP * 32) * d * d;                                    \
    }                                                                                                 \
    float2* dst = reinterpret_cast<float2*>(a.gn_stats) + ((long)it_n * nptiles + it_ptile) * a.Cout + it_cotile * BM + tid; \
    *dst = make_float2(mean, m2);                                                                     \
  }
