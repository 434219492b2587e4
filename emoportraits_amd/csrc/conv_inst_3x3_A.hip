// instantiations of conv_igemm_kernel: 3x3 taps, block config A
#include "conv_dispatch.h"
conv_launch_fn conv_lookup_3x3_A(int shape, int ups) {
  return CONV_FOR_SHAPE(3, 3, EMO_CONV_KC_3X3_A, 2, 2, 2, 2, shape, ups);
}
