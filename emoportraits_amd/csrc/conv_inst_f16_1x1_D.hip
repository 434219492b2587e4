// instantiations of conv_igemm_f16_kernel (fp16 MFMA operands, fp32 accumulation): 1x1 taps, 64 x 256 tiles
#include "conv_dispatch.h"
#include "conv_igemm_f16.h"
conv_launch_fn conv_lookup_f16_1x1_D(int shape, int ups) {
  return CONV_FOR_SHAPE_F16(1, 1, EMO_CONV_KC_F16_1X1, shape, ups);
}
