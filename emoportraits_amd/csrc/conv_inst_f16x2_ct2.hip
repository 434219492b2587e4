// instantiations of conv_igemm_bf16x3_ct2_kernel (conv_igemm_f16x2_ct2.h: the fp16 split with two 64-channel output tiles per
// work item on one converted patch), 4 x 64 pixel tiles with and without the fused nearest x2 upsample -- the decoder's layers
#include "conv_dispatch.h"
#include "conv_igemm_f16x2_ct2.h"
int conv_f16x2_ct2_4x64(ConvArgs a, hipStream_t s, int ups, int* rest_cot0) {
  return ups ? conv_f16x2_ct2_launch<4, 64, true>(a, s, rest_cot0) : conv_f16x2_ct2_launch<4, 64, false>(a, s, rest_cot0);
}
