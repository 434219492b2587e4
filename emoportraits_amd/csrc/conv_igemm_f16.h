// Implicit-GEMM convolution with fp16 MFMA operands and fp32 accumulation for gfx950 -- the reduced-precision mode of
// BASELINE.json configs[4] ("stage_2 refinement ... fp16 MFMA convs").  Opt-in per layer (PackedConv(precision="f16"));
// the exact-fp32 kernel of conv_igemm.h stays the default everywhere and is what bench.py measures.
//
// Same tiling, staging map, block order, K split and epilogue as conv_igemm.h; what changes is the operand format:
//   * activations stay fp32 NC(D)HW in HBM; the staging applies the producer's norm + ReLU in fp32 exactly as before and
//     converts to fp16 (round-to-nearest-even) on the way into LDS;
//   * v_mfma_f32_32x32x8_f16: the 8 k-values of one MFMA are 8 input CHANNELS at one tap; lane (l&31, l>>5) holds the four
//     consecutive channels 4*(l>>5) .. +3 of its row / column, so both operands are one aligned 8-byte LDS read:
//       patch   Ph[channel group of 4][patch element][4]   (wave w writes channel 4g + w: ds_write_b16)
//       weights Ah[k-group of 8][tap][half][BM][4]          (packed like that on the host, copied by LDS-DMA)
//   * a stage is KC = 8 (3x3) or 32 (1x1) channels: per wave 18 / 36 MFMAs of 32 cycles instead of 36 / 72 of 64.
// Rounding: operands carry 11 significand bits (activations saturate at +-65504 instead of overflowing to inf), products
// and sums are exact fp32 MFMA accumulation; measured error on the decoder layer shapes ~3e-4 of max|out|
// (tests/test_kernels_gpu.py).
#pragma once
#include "conv_igemm.h"

typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
struct ConvCfgH {
  static constexpr int BM = WGM * TM * 32;
  static constexpr int BP = WGP * TP * 32;
  static constexpr int TAPS = KH * KW;
  static constexpr int PR = TR + KH - 1;
  static constexpr int PW = TW + KW - 1;
  static constexpr int CHS = TZ * PR * PW;               // patch elements per input channel
  static constexpr int KG = KC / 8;                      // k-groups (one MFMA per k-group, tap and 32x32 tile)
  static constexpr int CPW = KC / 4;                     // channels staged per wave: c = 4 g + wave
  static constexpr int EPC = (CHS + 63) / 64;
  static constexpr int NPE = CPW * EPC;
  static constexpr int ASZ_H = KC * TAPS * BM;           // halfs of one stage's weight tile
  static constexpr int PATCH_H = KC * CHS;               // halfs of one stage's patch
  static constexpr int ASZ = ASZ_H / 2;                  // in floats
  static constexpr int BUF = ASZ + (((PATCH_H + 1) / 2 + 3) & ~3);
  static_assert(WGM * WGP == 4, "4 waves per block");
  static_assert(TZ * TR * TW == BP, "position tile must equal BP");
  static_assert(KC % 8 == 0, "whole k-groups of 8 channels");
  static_assert(ASZ_H % 8 == 0, "weight tile must be 16-byte copyable");
  static_assert(TM * TP <= 4, "accumulator budget");
};

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
void conv_igemm_f16_kernel(const ConvArgs a) {
  using Cfg = ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  constexpr int BM = Cfg::BM, TAPS = Cfg::TAPS, PR = Cfg::PR, PW = Cfg::PW, CHS = Cfg::CHS, KG = Cfg::KG;
  constexpr int ASZ = Cfg::ASZ, ASZ_H = Cfg::ASZ_H, BUF = Cfg::BUF, NPE = Cfg::NPE, CPW = Cfg::CPW, EPC = Cfg::EPC;

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
  const int tx = bx % a.tiles_x; bx /= a.tiles_x;
  const int ty = bx % a.tiles_y; bx /= a.tiles_y;
  const int tz = bx;
  const int x0 = tx * TW, y0 = ty * TR, z0 = tz * TZ;

  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const float* xn = a.x + (long)n * a.Cin * DHW;
  const bool has_affine = a.scale != nullptr;
  const int padD = a.KD >> 1;
  const float* scale_n = has_affine ? a.scale + (long)n * a.Cin : a.x;
  const float* shift_n = has_affine ? a.shift + (long)n * a.Cin : a.x;
  const float relu_floor = a.relu_in ? 0.0f : -__builtin_huge_valf();

  unsigned p_off[EPC];
  int p_pz[EPC];
  bool p_ok[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    const int e = lane + i * 64;
    const int pz = e / (PR * PW);
    const int rem2 = e - pz * (PR * PW);
    const int pr = rem2 / PW;
    const int pc = rem2 - pr * PW;
    const int yl = y0 + pr - (KH >> 1);
    const int xl = x0 + pc - (KW >> 1);
    const bool ok = (e < CHS) && ((unsigned)yl < (unsigned)a.Hl) && ((unsigned)xl < (unsigned)a.Wl);
    const int ys = UPS ? (yl >> 1) : yl;
    const int xs = UPS ? (xl >> 1) : xl;
    p_ok[i] = ok;
    p_off[i] = ok ? (unsigned)(ys * a.W + xs) * 4u : 0u;
    p_pz[i] = pz;
  }

  const int nstages_all = a.n_cchunks * a.KD;
  const int st_begin = ks * a.stages_per_split;
  const int st_end = min(nstages_all, st_begin + a.stages_per_split);
  const char* wsrc = reinterpret_cast<const char*>(a.wpk) + ((long)cotile * nstages_all) * (ASZ_H * 2);

  _Float16* const dumph = reinterpret_cast<_Float16*>(smem + 2 * BUF) + lane;   // per-lane dump slot (written, never read)

  float pv[NPE];
  bool pvz[NPE];
  bool sv[CPW];
  float sc[CPW], sh[CPW];

#define EMO_H_ISSUE_PATCH(stage_)                                                                     \
  {                                                                                                   \
    const int cc_ = (stage_) / a.KD;                                                                  \
    const int t_ = (stage_) - cc_ * a.KD;                                                             \
    const int ci0_ = cc_ * KC;                                                                        \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) {                                                 \
      const int c_ = ci0_ + g * 4 + wave;                                                             \
      const bool cv_ = c_ < a.Cin;                                                                    \
      const int cs_ = cv_ ? c_ : 0;                                                                   \
      const int zu_ = z0 + t_ - padD;                                                                 \
      const bool zv_ = (unsigned)zu_ < (unsigned)a.D;                                                 \
      const float* base_ = xn + (long)cs_ * DHW + (long)((TZ == 1 && zv_) ? zu_ : 0) * HW;            \
      sv[g] = cv_ && (TZ > 1 || zv_);                                                                 \
      { const float s1_ = scale_n[cs_], s0_ = shift_n[cs_];                                           \
        sc[g] = has_affine ? s1_ : 1.0f; sh[g] = has_affine ? s0_ : 0.0f; }                           \
      _Pragma("unroll") for (int i = 0; i < EPC; ++i) {                                               \
        if (TZ == 1) {                                                                                \
          pvz[g * EPC + i] = true;                                                                    \
          pv[g * EPC + i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_) + p_off[i]); \
        } else {                                                                                      \
          const int zi = zu_ + p_pz[i];                                                               \
          const bool zok = (unsigned)zi < (unsigned)a.D;                                              \
          pvz[g * EPC + i] = zok;                                                                     \
          pv[g * EPC + i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_) + (p_off[i] + (unsigned)((zok ? zi : 0) * HW) * 4u)); \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
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

// transform in fp32, round to fp16, one ds_write_b16 per element: Ph[(g * CHS + e) * 4 + wave]
#define EMO_H_STORE_PATCH(buf_)                                                                       \
  {                                                                                                   \
    _Float16* Ph_ = reinterpret_cast<_Float16*>((buf_) + ASZ);                                        \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) {                                                 \
      _Pragma("unroll") for (int i = 0; i < EPC; ++i) {                                               \
        const int e = lane + i * 64;                                                                  \
        float v = fmaxf(__fmaf_rn(pv[g * EPC + i], sc[g], sh[g]), relu_floor);                        \
        v = (p_ok[i] && sv[g] && pvz[g * EPC + i]) ? v : 0.0f;                                        \
        v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);   /* saturate instead of overflowing to inf */ \
        _Float16* d_ = ((i + 1) * 64 <= CHS || e < CHS) ? Ph_ + ((g * CHS + e) * 4 + wave) : dumph;   \
        *d_ = (_Float16)v;                                                                            \
      }                                                                                               \
    }                                                                                                 \
  }

  EMO_H_ISSUE_PATCH(st_begin);
  EMO_H_ISSUE_WEIGHTS(st_begin, smem);
  EMO_H_STORE_PATCH(smem);
  {
    const int st1_ = (st_begin + 1) < st_end ? (st_begin + 1) : st_begin;
    EMO_H_ISSUE_PATCH(st1_);
  }
  __syncthreads();

  floatx16 acc[TM][TP];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int a_base = half * BM + m0 + l32;   // in units of 4 halfs
  int b_base[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW;
    const int row = (p / TW) % TR;
    const int pz = p / (TW * TR);
    b_base[j] = half * CHS + pz * (PR * PW) + row * PW + col;
  }

  for (int st = st_begin; st < st_end; ++st) {
    float* cur = smem + ((st - st_begin) & 1) * BUF;
    float* nxt = smem + ((st - st_begin + 1) & 1) * BUF;
    const int stn = (st + 1) < st_end ? (st + 1) : st;
    const halfx4* Ah = reinterpret_cast<const halfx4*>(cur);
    const halfx4* Ph = reinterpret_cast<const halfx4*>(cur + ASZ);
    EMO_H_ISSUE_WEIGHTS(stn, nxt);
#pragma unroll
    for (int q = 0; q < KG; ++q) {
#pragma unroll
      for (int r = 0; r < KH; ++r) {
#pragma unroll
        for (int s = 0; s < KW; ++s) {
          const int tap = r * KW + s;
          halfx4 av_[TM], bv_[TP];
#pragma unroll
          for (int i = 0; i < TM; ++i) av_[i] = Ah[a_base + ((q * TAPS + tap) * 2) * BM + i * 32];
#pragma unroll
          for (int j = 0; j < TP; ++j) bv_[j] = Ph[b_base[j] + (q * 2) * CHS + r * PW + s];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(av_[i], bv_[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    EMO_H_STORE_PATCH(nxt);   // stage st+1 (loaded during the previous stage)
    {
      const int stn2 = (st + 2) < st_end ? (st + 2) : (st_end - 1);
      EMO_H_ISSUE_PATCH(stn2);
    }
    __syncthreads();
  }
#undef EMO_H_ISSUE_PATCH
#undef EMO_H_ISSUE_WEIGHTS
#undef EMO_H_STORE_PATCH

  // ---- epilogue (same as conv_igemm.h): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
  const long plane = (long)a.Hl * a.Wl;
  const long ovol = (long)a.Dl * plane;
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW;
    const int row = (p / TW) % TR;
    const int pz = p / (TW * TR);
    const int z = z0 + pz, y = y0 + row, x = x0 + col;
    const long sp = (long)z * plane + (long)y * a.Wl + x;
    long rsp = sp;
    long rvol = ovol;
    if (a.res_ups) {
      const int Wr = a.Wl >> 1, Hr = a.Hl >> 1;
      rsp = ((long)z * Hr + (y >> 1)) * Wr + (x >> 1);
      rvol = (long)a.Dl * Hr * Wr;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cotile * BM + m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.Cout && a.partial) {
          a.partial[(((long)ks * a.N + n) * a.Cout + co) * ovol + sp] = acc[i][j][r];
        } else if (co < a.Cout) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[co];
          if (a.res) v += a.res[((long)n * a.Cout + co) * rvol + rsp];
          v = emo_act(v, a.act);
          a.out[((long)n * a.Cout + co) * ovol + sp] = v;
        }
      }
    }
  }
}

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
int conv_igemm_f16_launch(ConvArgs a, hipStream_t s) {
  using Cfg = ConvCfgH<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (a.Wl % TW || a.Hl % TR || a.Dl % TZ) return EMO_ERR_UNSUPPORTED;
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl / TZ;
  a.n_cchunks = (a.Cin + KC - 1) / KC;
  const long nt = (long)a.tiles_x * a.tiles_y * a.tiles_z;
  if (nt > 0x7fffffffL || a.N > 65535) return EMO_ERR_UNSUPPORTED;
  const int cot = (a.Cout + Cfg::BM - 1) / Cfg::BM;
  const size_t lds = (size_t)(2 * Cfg::BUF + 64) * sizeof(float);   // two stage buffers + 64 dump slots
  if (lds > 160 * 1024) return EMO_ERR_UNSUPPORTED;
  auto kern = conv_igemm_f16_kernel<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      raised = true;
    }
  }
  a.n_cotiles = cot;
  if (a.ksplit < 1 || (a.ksplit > 1 && !a.partial)) return EMO_ERR_BAD_ARG;
  if (a.ksplit == 1) { a.stages_per_split = a.n_cchunks * a.KD; a.partial = nullptr; }
  if (nt * cot * a.N * a.ksplit > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nt * cot * a.N * a.ksplit)), dim3(256), lds, s, a);
  return emo_launch_status();
}
