// instantiations of conv_igemm_kernel: 1x7 taps per depth slice (a 7x7 2-D conv run as KD=7 over the image rows),
// block config B.  LocalEncoder.from_rgb (local_encoder.py:64-73): image widths that are multiples of 128, or 64.
#include "conv_dispatch.h"
conv_launch_fn conv_lookup_1x7_B(int shape, int ups) {
  if (ups) return nullptr;
  if (shape == SHAPE_W128) return &conv_igemm_launch<1, 7, EMO_CONV_KC_1X7, 1, 1, 128, 2, 1, 1, 4, false>;
  if (shape == SHAPE_W64) return &conv_igemm_launch<1, 7, EMO_CONV_KC_1X7, 2, 1, 64, 2, 1, 1, 4, false>;
  return nullptr;
}
