// LDS-staged 3-D grid_sample (gs3d_tile.h), padding_mode 'zeros': instantiations of one padding mode per translation unit.
#include "gs3d_tile_launch.h"
EMO_GS3D_TILE_PAD_SIGNATURE(emo_gs3d_tile_zeros) {
  return gs3d::launch_tile_pad<EMO_PAD_ZEROS>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                  in_layout, out_layout, variant, grid_kind, s);
}
