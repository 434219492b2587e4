// instantiations of conv_igemm_f16_kernel (fp16 MFMA operands, fp32 accumulation): 3x3 taps, 128 x 256 tiles
#include "conv_dispatch.h"
#include "conv_igemm_f16.h"
conv_launch_fn conv_lookup_f16_3x3_G(int shape, int ups) {
  return CONV_FOR_SHAPE_F16_G(3, 3, EMO_CONV_KC_F16_3X3, shape, ups);
}
