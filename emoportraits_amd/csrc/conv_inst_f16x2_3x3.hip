// instantiations of conv_igemm_bf16x3_kernel with SPLIT = 2 (scaled fp32 operands as two fp16 terms, three products, fp32
// accumulation: opt-in, emo_conv_igemm_f16x2): 3x3 taps, 64 x 256 tiles of 4 x 64, 8 x 32 or 16 x 16 pixels
#include "conv_dispatch.h"
#include "conv_igemm_bf16x3.h"
int conv_f16x2_ct2_4x64(ConvArgs, hipStream_t, int ups, int* rest_cot0);    // conv_inst_f16x2_ct2.hip
int conv_f16x2_w8_4x64(ConvArgs, hipStream_t, int ups, int* rest_cot0);     // conv_inst_f16x2_w8.hip
// 4 x 64 tiles: the layer's channel-tile pairs on a two-tile kernel where that fills the chip -- the one with two waves per SIMD
// (conv_igemm_f16x2_w8.h; EMO_CONV_W8=0: the one-wave-per-SIMD kernel of conv_igemm_f16x2_ct2.h) --, the rest (an odd last tile,
// or everything) on the single-tile one
template <bool UPS>
static int conv_f16x2_4x64(ConvArgs a, hipStream_t s) {
  int rest = 0;
  int rc = conv_f16x2_w8_4x64(a, s, UPS, &rest);
  if (rc != EMO_OK) return rc;
  if (rest == 0) rc = conv_f16x2_ct2_4x64(a, s, UPS, &rest);
  if (rc != EMO_OK) return rc;
  if (rest * ConvCfgS<4, 64, UPS, 2>::BM >= a.Cout) return EMO_OK;
  a.cot0 = rest;
  return conv_igemm_bf16x3_launch<4, 64, UPS, 2>(a, s);
}
// 32-row channel tiles (block config F as the tile id: 32 channels x 256 positions; weights packed for BM = 32): layers with at
// most 32 output channels per tile row -- the WarpGenerator's last 3-D block, stage 2's 32-channel ResBlocks
conv_launch_fn conv_lookup_f16x2_3x3_bm32(int Wl, int ups) {
  if (Wl % 64 == 0 && !ups) return &conv_igemm_bf16x3_launch<4, 64, false, 2, 32>;
  return nullptr;
}
conv_launch_fn conv_lookup_f16x2_3x3(int Wl, int ups) {
  if (Wl % 64 == 0) return ups ? &conv_f16x2_4x64<true> : &conv_f16x2_4x64<false>;
  if (Wl == 32 && !ups) return &conv_igemm_bf16x3_launch<8, 32, false, 2>;
  if (Wl == 16 && !ups) return &conv_igemm_bf16x3_launch<16, 16, false, 2>;     // (the 16-wide 3-D maps of the WarpGenerator)
  return nullptr;
}

#if EMO_S_TIMING
// measurement builds only: the per-work-item phase stamps of the last launches (conv_igemm_bf16x3.h, EMO_S_TIMING)
extern "C" int emo_debug_conv_timing_f16x2(unsigned long long* host_out, int n_items) {
  if (!host_out || n_items < 0 || n_items > EMO_S_TLOG_N) return EMO_ERR_BAD_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return EMO_ERR_BAD_ARG;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(emo_s_tlog), (size_t)n_items * EMO_S_TLOG_W * sizeof(unsigned long long)) == hipSuccess ? EMO_OK : EMO_ERR_BAD_ARG;
}
#endif
