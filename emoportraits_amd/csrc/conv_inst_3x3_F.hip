// instantiations of conv_igemm_kernel: 3x3 taps, block config F (32 output channels x 256 positions)
#include "conv_dispatch.h"
conv_launch_fn conv_lookup_3x3_F(int shape, int ups) {
  return CONV_FOR_SHAPE_F(3, 3, EMO_CONV_KC_3X3, shape, ups);
}
