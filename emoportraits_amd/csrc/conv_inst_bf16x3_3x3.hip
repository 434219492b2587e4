// instantiations of conv_igemm_bf16x3_kernel (fp32 operands as three bf16 terms, six products, fp32 accumulation): 3x3 taps,
// 64 x 256 tiles of 4 x 64 pixels (map widths that are multiples of 64) or 8 x 32 pixels (32-wide maps, no upsample)
#include "conv_dispatch.h"
#include "conv_igemm_bf16x3.h"
conv_launch_fn conv_lookup_bf16x3_3x3(int Wl, int ups) {
  if (Wl % 64 == 0) return ups ? &conv_igemm_bf16x3_launch<4, 64, true> : &conv_igemm_bf16x3_launch<4, 64, false>;
  if (Wl == 32 && !ups) return &conv_igemm_bf16x3_launch<8, 32, false>;
  return nullptr;
}
