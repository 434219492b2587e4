// instantiations of conv_igemm_kernel: 1x1 taps, block config A
#include "conv_dispatch.h"
conv_launch_fn conv_lookup_1x1_A(int shape, int ups) {
  return CONV_FOR_SHAPE(1, 1, EMO_CONV_KC_1X1, 2, 2, 2, 2, shape, ups);
}
