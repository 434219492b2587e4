// C ABI of the implicit-GEMM convolution (include/emo_hip.h: emo_conv_igemm_f32, emo_conv_pack_info).
#include "conv_dispatch.h"

conv_launch_fn conv_lookup_3x3_A(int, int);
conv_launch_fn conv_lookup_3x3_B(int, int);
conv_launch_fn conv_lookup_3x3_C(int, int);
conv_launch_fn conv_lookup_3x3_D(int, int);
conv_launch_fn conv_lookup_3x3_E(int, int);
conv_launch_fn conv_lookup_3x3_F(int, int);
conv_launch_fn conv_lookup_1x1_A(int, int);
conv_launch_fn conv_lookup_1x1_B(int, int);
conv_launch_fn conv_lookup_1x1_C(int, int);
conv_launch_fn conv_lookup_1x7_A(int, int);
conv_launch_fn conv_lookup_1x7_B(int, int);
conv_launch_fn conv_lookup_f16_3x3_D(int, int);
conv_launch_fn conv_lookup_f16_1x1_D(int, int);
conv_launch_fn conv_lookup_f16_3x3_G(int, int);
conv_launch_fn conv_lookup_bf16x3_3x3(int, int);
conv_launch_fn conv_lookup_f16x2_3x3(int, int);
conv_launch_fn conv_lookup_f16x2_1x1(int, int);
conv_launch_fn conv_lookup_f16x2_3x3_bm32(int, int);
conv_launch_fn conv_lookup_f16w8_3x3(int, int);

// MFMA operand format (accumulation and all tensors in HBM are fp32 either way).  PREC_S: every fp32 operand as the exact sum
// of three bf16 terms, six partial products (conv_igemm_bf16x3.h) -- fp32 results on the bf16 pipes
// PREC_S2: the scaled operand as two fp16 terms, three partial products (same kernel, SPLIT = 2; opt-in)
// PREC_H1: plain fp16 operands on the eight-wave two-tile kernel (conv_igemm_f16x2_w8.h, NPROD = 1): weights = the first plane
// of the split layout
enum { PREC_F32 = 0, PREC_F16 = 1, PREC_S = 2, PREC_S2 = 3, PREC_H1 = 4 };

static int shape_of_width(int Wl) {
  if (Wl >= 128 && Wl % 128 == 0) return SHAPE_W128;
  if (Wl == 64) return SHAPE_W64;
  if (Wl == 32) return SHAPE_W32;
  if (Wl == 16) return SHAPE_W16;
  if (Wl == 8) return SHAPE_W8;
  return -1;
}

static int kc_of(int KH, int KW, int cfg, int prec = PREC_F32) {
  if (prec == PREC_H1) return (cfg == CFG_D && KH == 3 && KW == 3) ? 32 : 0;   // two 16-channel k-blocks per stage
  if (prec == PREC_S2 && cfg == CFG_D && KH == 1 && KW == 1) return 32;   // conv_igemm_f16x2_p1.h: pointwise, 32 channels per stage
  if (prec == PREC_S2 && cfg == CFG_F && KH == 3 && KW == 3) return 16;   // fp16 split on 32-row channel tiles (conv_igemm_bf16x3.h, BMT = 32)
  if (prec == PREC_S || prec == PREC_S2) return (cfg == CFG_D && KH == 3 && KW == 3) ? 16 : 0;
  if (prec == PREC_F16) {
    if (cfg == CFG_G) return (KH == 3 && KW == 3) ? EMO_CONV_KC_F16_3X3 : 0;   // 128 x 256 tile: 3x3 only
    if (cfg != CFG_D) return 0;   // otherwise the fp16-operand kernels exist for the 64 x 256 tile only
    return (KH == 3 && KW == 3) ? EMO_CONV_KC_F16_3X3 : (KH == 1 && KW == 1) ? EMO_CONV_KC_F16_1X1 : 0;
  }
  if (cfg == CFG_G) return 0;                                   // fp16 operands only
  if (KH == 3 && KW == 3) return cfg == CFG_A ? EMO_CONV_KC_3X3_A : EMO_CONV_KC_3X3;
  if (cfg == CFG_D || cfg == CFG_E || cfg == CFG_F) return 0;   // 3x3 (x3) only
  if (KH == 1 && KW == 1) return EMO_CONV_KC_1X1;
  if (KH == 1 && KW == 7) return EMO_CONV_KC_1X7;
  return 0;
}

extern "C" int emo_conv_pack_info(int KH, int KW, int cfg, int* BM, int* KC) {
  if (!BM || !KC) return EMO_ERR_BAD_ARG;
  if (cfg == CFG_A) *BM = 128; else if (cfg == CFG_B || cfg == CFG_D || cfg == CFG_E) *BM = 64; else if (cfg == CFG_C || cfg == CFG_F) *BM = 32; else return EMO_ERR_BAD_ARG;
  *KC = kc_of(KH, KW, cfg);
  return *KC ? EMO_OK : EMO_ERR_UNSUPPORTED;
}

// output positions per block of a config (the tile the GroupNorm statistics of gn_stats are reduced over)
extern "C" int emo_conv_tile_positions(int cfg) {
  if (cfg == CFG_A || cfg == CFG_B || cfg == CFG_C) return 128;
  if (cfg == CFG_D || cfg == CFG_F || cfg == CFG_G) return 256;
  if (cfg == CFG_E) return 512;
  return EMO_ERR_BAD_ARG;
}

// second half of a split-K launch: out = act(sum_ks partial[ks] + bias + residual), ks ascending (deterministic)
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const float* __restrict__ partial,
                                                                   const float* __restrict__ bias, const float* res,
                                                                   float* out, long total, int ksplit, int Cout, int Dl,
                                                                   int Hl, int Wl, int act, int res_ups, const int* run_if) {
  if (run_if != nullptr && *run_if == 0) return;   // second half of a guarded fallback launch (ConvArgs::run_if)
  const long ovol = (long)Dl * Hl * Wl;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float v = partial[i];
    for (int k = 1; k < ksplit; ++k) v += partial[(long)k * total + i];
    const long nc = i / ovol;
    if (bias) v += bias[nc % Cout];
    if (res) {
      if (res_ups) {
        const long sp = i - nc * ovol;
        const int x = (int)(sp % Wl);
        const long r = sp / Wl;
        const int y = (int)(r % Hl), z = (int)(r / Hl);
        const int Wr = Wl >> 1, Hr = Hl >> 1;
        v += res[nc * ((long)Dl * Hr * Wr) + ((long)z * Hr + (y >> 1)) * Wr + (x >> 1)];
      } else {
        v += res[i];
      }
    }
    out[i] = emo_act(v, act);
  }
}


// launch heuristic: split the K loop until the launch has >= 2 blocks per CU, keeping >= 8 stages per split
extern "C" int emo_conv_pack_info_f16(int KH, int KW, int cfg, int* BM, int* KC) {
  if (!BM || !KC) return EMO_ERR_BAD_ARG;
  if (cfg == CFG_D) *BM = 64; else if (cfg == CFG_G) *BM = 128; else return EMO_ERR_UNSUPPORTED;
  *KC = kc_of(KH, KW, cfg, PREC_F16);
  return *KC ? EMO_OK : EMO_ERR_UNSUPPORTED;
}

extern "C" int emo_conv_pack_info_bf16x3(int KH, int KW, int cfg, int* BM, int* KC) {
  if (!BM || !KC) return EMO_ERR_BAD_ARG;
  if (cfg != CFG_D) return EMO_ERR_UNSUPPORTED;
  *BM = 64;
  *KC = kc_of(KH, KW, cfg, PREC_S);
  return *KC ? EMO_OK : EMO_ERR_UNSUPPORTED;
}

extern "C" int emo_conv_igemm_ksplit(int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW, int ups,
                                     int cfg) {
  const int kc = kc_of(KH, KW, cfg);
  if (!kc || N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0 || cfg < 0 || cfg >= N_CFGS) return EMO_ERR_BAD_ARG;
  const int bm = (cfg == CFG_A || cfg == CFG_G) ? 128 : (cfg == CFG_B || cfg == CFG_D || cfg == CFG_E) ? 64 : 32;
  const int bp = (cfg == CFG_D || cfg == CFG_F || cfg == CFG_G) ? 256 : cfg == CFG_E ? 512 : 128;
  const long pos = (long)N * D * (ups ? 4 : 1) * H * W;
  const long blocks = ((pos + bp - 1) / bp) * ((Cout + bm - 1) / bm);
  const int nstages = ((Cin + kc - 1) / kc) * KD;
  long want = blocks >= 512 ? 1 : (512 + blocks - 1) / blocks;
  if (want > nstages / 8) want = nstages / 8;
  if (want > 16) want = 16;
  return want < 1 ? 1 : (int)want;
}

static int conv_igemm_dispatch(int prec, const float* x, const void* wpk, const float* bias, const float* scale,
                               const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D, int H,
                               int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups, int cfg,
                               int ksplit, float* workspace, float* gn_stats, void* stream, float in_scale = 1.0f,
                               float w_scale = 1.0f, int* sat_flag = nullptr, const int* run_if = nullptr, int cot0 = 0,
                               int cot_end = 0) {
  if (!x || !wpk || !out) return EMO_ERR_BAD_ARG;
  if (gn_stats && ksplit > 1) return EMO_ERR_UNSUPPORTED;   // tile statistics come from the single-pass epilogue
  if ((long)D * H * W >= (1L << 30)) return EMO_ERR_UNSUPPORTED;                 // 32-bit byte offsets inside one channel
  if (ksplit < 1 || (ksplit > 1 && !workspace)) return EMO_ERR_BAD_ARG;
  if (N <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return EMO_ERR_BAD_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMO_ERR_BAD_ARG;
  if (KD != 1 && KD != 3 && KD != 7) return EMO_ERR_UNSUPPORTED;
  if (KD == 3 && !(KH == 3 && KW == 3)) return EMO_ERR_UNSUPPORTED;
  if ((KD == 7) != (KH == 1 && KW == 7)) return EMO_ERR_UNSUPPORTED;   // 7x7 2-D conv = KD 7 over rows x 1x7 taps
  if (cfg < 0 || cfg >= N_CFGS) return EMO_ERR_BAD_ARG;
  if (!emo_aligned16(wpk)) return EMO_ERR_ALIGN;
  if (ups && D != 1) return EMO_ERR_UNSUPPORTED;
  if ((long)Cin * D * H * W >= (1L << 31)) return EMO_ERR_UNSUPPORTED;   // 32-bit per-sample element offsets
  ConvArgs a;
  a.x = x; a.wpk = reinterpret_cast<const float*>(wpk); a.bias = bias; a.scale = scale; a.shift = shift; a.res = res; a.out = out;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
  a.Dl = D; a.Hl = ups ? 2 * H : H; a.Wl = ups ? 2 * W : W;
  a.KD = KD; a.relu_in = relu_in; a.act = act; a.res_ups = res_ups;
  a.gn_stats = gn_stats;
  a.n_cchunks = 0; a.tiles_x = a.tiles_y = a.tiles_z = 0; a.n_cotiles = 0; a.n_work = 0; a.cot0 = cot0; a.cot_end = cot_end;
  if ((cot0 != 0 && !(prec == PREC_F16 && cfg == CFG_D && ksplit == 1)) || (cot_end != 0 && prec != PREC_H1)) return EMO_ERR_BAD_ARG;
  a.in_scale = in_scale; a.out_scale = 1.0f / (in_scale * w_scale);
  a.sat_flag = sat_flag; a.run_if = run_if;
  const int shape = shape_of_width(a.Wl);
  if (shape < 0) return EMO_ERR_UNSUPPORTED;
  conv_launch_fn fn = nullptr;
  if (prec == PREC_H1) {
    if (!(KH == 3 && KW == 3 && (KD == 1 || KD == 3)) || cfg != CFG_D || Cin % 8 || ksplit != 1 || run_if != nullptr) return EMO_ERR_UNSUPPORTED;
    if (!(in_scale > 0.0f && w_scale > 0.0f)) return EMO_ERR_BAD_ARG;
    fn = conv_lookup_f16w8_3x3(a.Wl, ups);
  } else if (prec == PREC_S2 && KH == 1 && KW == 1) {
    // pointwise layers on the fp16 split (conv_igemm_f16x2_p1.h): one launch form, no K split
    if (KD != 1 || cfg != CFG_D || Cin % 8 || ksplit != 1 || run_if != nullptr) return EMO_ERR_UNSUPPORTED;
    if (!(in_scale > 0.0f && w_scale > 0.0f)) return EMO_ERR_BAD_ARG;
    fn = conv_lookup_f16x2_1x1(a.Wl, ups);
  } else if (prec == PREC_S2 && cfg == CFG_F) {
    // the fp16 split on 32-row channel tiles (weights packed for BM = 32: emoportraits_amd.pack.pack_weight_f16x2(w, bm=32))
    if (!(KH == 3 && KW == 3 && (KD == 1 || KD == 3)) || Cin % 8) return EMO_ERR_UNSUPPORTED;
    if (!(in_scale > 0.0f && w_scale > 0.0f)) return EMO_ERR_BAD_ARG;
    fn = conv_lookup_f16x2_3x3_bm32(a.Wl, ups);
  } else if (prec == PREC_S || prec == PREC_S2) {
    if (!(KH == 3 && KW == 3 && (KD == 1 || KD == 3)) || cfg != CFG_D || Cin % 8) return EMO_ERR_UNSUPPORTED;
    if (prec == PREC_S2 && !(in_scale > 0.0f && w_scale > 0.0f)) return EMO_ERR_BAD_ARG;
    fn = prec == PREC_S ? conv_lookup_bf16x3_3x3(a.Wl, ups) : conv_lookup_f16x2_3x3(a.Wl, ups);
  } else if (prec == PREC_F16) {
    if (KD != 1 && !(KD == 3 && KH == 3)) return EMO_ERR_UNSUPPORTED;
    if ((cfg != CFG_D && cfg != CFG_G) || Cin % 8) return EMO_ERR_UNSUPPORTED;
    if (cfg == CFG_G) fn = (KH == 3 && KW == 3) ? conv_lookup_f16_3x3_G(shape, ups) : nullptr;
    else if (KH == 3 && KW == 3) fn = conv_lookup_f16_3x3_D(shape, ups);
    else if (KH == 1 && KW == 1) fn = conv_lookup_f16_1x1_D(shape, ups);
    else return EMO_ERR_UNSUPPORTED;
  } else if (cfg == CFG_G) {
    return EMO_ERR_UNSUPPORTED;   // fp16 operands only
  } else if (cfg == CFG_D || cfg == CFG_E || cfg == CFG_F) {
    // position tiles of one depth slice (TZ = 1): 2-D layers, and 3-D layers whose depth taps run as K stages
    if (!(KH == 3 && KW == 3 && (KD == 1 || (KD == 3 && cfg != CFG_E)))) return EMO_ERR_UNSUPPORTED;
    fn = cfg == CFG_D ? conv_lookup_3x3_D(shape, ups) : cfg == CFG_E ? conv_lookup_3x3_E(shape, ups) : conv_lookup_3x3_F(shape, ups);
  } else if (KH == 3 && KW == 3) {
    fn = cfg == CFG_A ? conv_lookup_3x3_A(shape, ups) : cfg == CFG_B ? conv_lookup_3x3_B(shape, ups) : conv_lookup_3x3_C(shape, ups);
  } else if (KH == 1 && KW == 1) {
    fn = cfg == CFG_A ? conv_lookup_1x1_A(shape, ups) : cfg == CFG_B ? conv_lookup_1x1_B(shape, ups) : conv_lookup_1x1_C(shape, ups);
  } else if (KH == 1 && KW == 7) {
    fn = cfg == CFG_A ? conv_lookup_1x7_A(shape, ups) : cfg == CFG_B ? conv_lookup_1x7_B(shape, ups) : nullptr;
  } else {
    return EMO_ERR_UNSUPPORTED;
  }
  if (!fn) return EMO_ERR_UNSUPPORTED;
  const int kc = kc_of(KH, KW, cfg, prec);
  if (!kc) return EMO_ERR_UNSUPPORTED;
  const int nstages = ((Cin + kc - 1) / kc) * KD;
  if (ksplit > nstages) ksplit = nstages;
  a.stages_per_split = (nstages + ksplit - 1) / ksplit;
  a.ksplit = (nstages + a.stages_per_split - 1) / a.stages_per_split;   // no empty split
  a.partial = a.ksplit > 1 ? workspace : nullptr;
  const int rc = fn(a, (hipStream_t)stream);
  if (rc != EMO_OK || a.ksplit == 1) return rc;
  const long total = (long)N * Cout * a.Dl * a.Hl * a.Wl;
  long blocks = (total + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, workspace, bias,
                     res, out, total, a.ksplit, Cout, a.Dl, a.Hl, a.Wl, act, res_ups, run_if);
  return emo_launch_status();
}

extern "C" int emo_conv_igemm_f32(const float* x, const float* wpk, const float* bias, const float* scale,
                                  const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D,
                                  int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups,
                                  int cfg, int ksplit, float* workspace, float* gn_stats, void* stream) {
  return conv_igemm_dispatch(PREC_F32, x, wpk, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups,
                             relu_in, act, res_ups, cfg, ksplit, workspace, gn_stats, stream);
}

extern "C" int emo_conv_igemm_f32_guarded(const float* x, const float* wpk, const float* bias, const float* scale,
                                          const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D,
                                          int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups,
                                          int cfg, int ksplit, float* workspace, float* gn_stats, void* stream,
                                          const int* run_if) {
  return conv_igemm_dispatch(PREC_F32, x, wpk, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups,
                             relu_in, act, res_ups, cfg, ksplit, workspace, gn_stats, stream, 1.0f, 1.0f, nullptr, run_if);
}

extern "C" int emo_conv_igemm_bf16x3(const float* x, const void* wpk3, const float* bias, const float* scale,
                                     const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D,
                                     int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups,
                                     int cfg, int ksplit, float* workspace, float* gn_stats, void* stream,
                                     const int* run_if) {
  return conv_igemm_dispatch(PREC_S, x, wpk3, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups,
                             relu_in, act, res_ups, cfg, ksplit, workspace, gn_stats, stream, 1.0f, 1.0f, nullptr, run_if);
}

extern "C" int emo_conv_igemm_f16x2(const float* x, const void* wpk2, const float* bias, const float* scale,
                                    const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D,
                                    int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups,
                                    int cfg, int ksplit, float* workspace, float* gn_stats, void* stream, float in_scale,
                                    float w_scale, int* overflow_flag) {
  return conv_igemm_dispatch(PREC_S2, x, wpk2, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups,
                             relu_in, act, res_ups, cfg, ksplit, workspace, gn_stats, stream, in_scale, w_scale,
                             overflow_flag, nullptr);
}

extern "C" int emo_conv_igemm_f16acc32(const float* x, const void* wpk16, const float* bias, const float* scale,
                                       const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D,
                                       int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups,
                                       int cfg, int ksplit, float* workspace, float* gn_stats, void* stream) {
  return conv_igemm_dispatch(PREC_F16, x, wpk16, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups,
                             relu_in, act, res_ups, cfg, ksplit, workspace, gn_stats, stream);
}

// ABI 10.  A layer with an ODD number of 64-channel tiles in the plain-fp16 mode: its whole pairs on the eight-wave two-tile
// kernel (wpk1), the last tile on the older fp16-operand kernel (wpk16, that kernel's layout for block config 3) -- two
// launches, one call, every output element written once: see include/emo_hip.h
extern "C" int emo_conv_igemm_f16w8_rest(const float* x, const void* wpk1, const void* wpk16, const float* bias,
                                         const float* scale, const float* shift, const float* res, float* out, int N, int Cin,
                                         int Cout, int D, int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act,
                                         int res_ups, int cfg, int ksplit, float* workspace, float* gn_stats, void* stream,
                                         float w_scale) {
  const int cot = (Cout + 63) / 64;
  if (!wpk16 || Cout % 64 || cot < 3 || !(cot & 1)) return EMO_ERR_UNSUPPORTED;
  // the plane must be in the OLDER kernel's form too (its 2 x 128 / 4 x 64 position tiles) -- checked before anything is launched
  // (today shape_of_width admits no other width that the eight-wave kernel takes; this keeps the two launches all-or-nothing if
  // that changes): whatever else that launcher refuses, the eight-wave launcher in front of it refuses first
  const int Wl = ups ? 2 * W : W, Hl = ups ? 2 * H : H;
  if (!((Wl % 128 == 0 && Hl % 2 == 0) || (Wl == 64 && Hl % 4 == 0))) return EMO_ERR_UNSUPPORTED;
  int rc = conv_igemm_dispatch(PREC_H1, x, wpk1, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups, relu_in,
                               act, res_ups, cfg, ksplit, workspace, gn_stats, stream, 1.0f, w_scale, nullptr, nullptr, 0, cot - 1);
  if (rc != EMO_OK) return rc;
  return conv_igemm_dispatch(PREC_F16, x, wpk16, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups, relu_in,
                             act, res_ups, cfg, ksplit, workspace, gn_stats, stream, 1.0f, 1.0f, nullptr, nullptr, cot - 1, 0);
}

// ABI 9.  Plain fp16 operands (BASELINE configs[4]) on the eight-wave two-tile kernel: see include/emo_hip.h
extern "C" int emo_conv_igemm_f16w8(const float* x, const void* wpk1, const float* bias, const float* scale,
                                    const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D,
                                    int H, int W, int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups,
                                    int cfg, int ksplit, float* workspace, float* gn_stats, void* stream, float w_scale) {
  return conv_igemm_dispatch(PREC_H1, x, wpk1, bias, scale, shift, res, out, N, Cin, Cout, D, H, W, KD, KH, KW, ups,
                             relu_in, act, res_ups, cfg, ksplit, workspace, gn_stats, stream, 1.0f, w_scale, nullptr, nullptr);
}
