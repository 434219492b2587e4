// instantiation of conv_igemm_bf16x3_p1_kernel (conv_igemm_f16x2_p1.h: pointwise convolutions on the fp16 split, two channel tiles
// per work item), 4 x 64 pixel tiles: the decoder's 1536 -> 512 entry convolution and the 1x1 skips of its up-blocks
#include "conv_dispatch.h"
#include "conv_igemm_f16x2_p1.h"
conv_launch_fn conv_lookup_f16x2_1x1(int Wl, int ups) {
  if (Wl % 64 == 0 && !ups) return &conv_f16x2_p1_launch<4, 64>;
  return nullptr;
}
