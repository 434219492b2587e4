// instantiations of conv_igemm_bf16x3_kernel with SPLIT = 2 (scaled fp32 operands as two fp16 terms, three products, fp32
// accumulation: opt-in, emo_conv_igemm_f16x2): 3x3 taps, 64 x 256 tiles of 4 x 64 or 8 x 32 pixels
#include "conv_dispatch.h"
#include "conv_igemm_bf16x3.h"
conv_launch_fn conv_lookup_f16x2_3x3(int Wl, int ups) {
  if (Wl % 64 == 0) return ups ? &conv_igemm_bf16x3_launch<4, 64, true, 2> : &conv_igemm_bf16x3_launch<4, 64, false, 2>;
  if (Wl == 32 && !ups) return &conv_igemm_bf16x3_launch<8, 32, false, 2>;
  return nullptr;
}
