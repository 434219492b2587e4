// instantiations of conv_igemm_f16x2_w8_kernel (conv_igemm_f16x2_w8.h: the two-tile fp16 split with two waves per SIMD), 4 x 64
// pixel tiles with and without the fused nearest x2 upsample -- the decoder's layers
#include "conv_dispatch.h"
#include "conv_igemm_f16x2_w8.h"
int conv_f16x2_w8_4x64(ConvArgs a, hipStream_t s, int ups, int* rest_cot0) {
  return ups ? conv_f16x2_w8_launch<4, 64, true>(a, s, rest_cot0) : conv_f16x2_w8_launch<4, 64, false>(a, s, rest_cot0);
}

// plain fp16 operands (NPROD = 1; emo_conv_igemm_f16w8): the whole layer in one launch, or EMO_ERR_UNSUPPORTED when the launch is
// not in the kernel's form (the caller -- emoportraits_amd/pack.py mirrors the conditions -- then runs conv_igemm_f16.h)
template <bool UPS>
static int conv_f16_w8_4x64(ConvArgs a, hipStream_t s) {
  int rest = 0;
  const int rc = conv_f16x2_w8_launch<4, 64, UPS, 1>(a, s, &rest);
  if (rc != EMO_OK) return rc;
  return rest > 0 ? EMO_OK : EMO_ERR_UNSUPPORTED;
}
conv_launch_fn conv_lookup_f16w8_3x3(int Wl, int ups) {
  if (Wl % 64 == 0) return ups ? &conv_f16_w8_4x64<true> : &conv_f16_w8_4x64<false>;
  return nullptr;
}

#if EMO_S_TIMING
// measurement builds only: the per-work-item phase stamps of the last launch (conv_igemm_f16x2_w8.h, EMO_S_TIMING)
extern "C" int emo_debug_conv_timing_w8(unsigned long long* host_out, int n_items) {
  if (!host_out || n_items < 0 || n_items > EMO_S_TLOG_N) return EMO_ERR_BAD_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return EMO_ERR_BAD_ARG;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(emo_s_tlog), (size_t)n_items * EMO_S_TLOG_W * sizeof(unsigned long long)) == hipSuccess ? EMO_OK : EMO_ERR_BAD_ARG;
}
#endif
