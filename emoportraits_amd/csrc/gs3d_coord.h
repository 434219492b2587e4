// Coordinate and weight arithmetic of the 3-D trilinear grid_sample (align_corners=False), shared by every sampler
// kernel of this library -- SURVEY.md section 8 rows a1 + a2.
//
// The arithmetic restated here is ATen's CPU grid_sampler_3d, which is what the reference's
//   F.grid_sample(inputs.float(), grid.float(), padding_mode=...)      (models/stage_1/volumetric_avatar/va.py:264-265)
// executes (ATen/native/GridSampler.h: grid_sampler_unnormalize :27-36, clip_coordinates :58-60, reflect_coordinates
// :89-106).  It is kept bit-identical: fp32 with explicit round-to-nearest operations and no FMA contraction, corner
// weights (wx*wy)*wz, corners in the order tnw,tne,tsw,tse,bnw,bne,bsw,bse.
//
// The header compiles in two worlds: device code under hipcc (the product), and plain host C++ when GS3D_HOST_EMULATION is
// defined -- tests/emul/gs3d_tile_emul.cpp runs the tile kernel's phases thread by thread on the CPU (test infrastructure:
// the index logic of the LDS-staged kernel is checked against the oracle without a GPU).  The host build must be compiled
// with -ffp-contract=off.
#pragma once

#if defined(GS3D_HOST_EMULATION)
#include <math.h>
#include <stdint.h>
#include <string.h>
#define GS_FN static inline
#define GS_MFN inline          /* member functions */
#define gs_fmul(a, b) ((float)(a) * (float)(b))
#define gs_fadd(a, b) ((float)(a) + (float)(b))
#define gs_fsub(a, b) ((float)(a) - (float)(b))
#define gs_fdiv(a, b) ((float)(a) / (float)(b))
#define gs_fma(a, b, c) fmaf((a), (b), (c))
#ifndef EMO_PAD_ZEROS
#define EMO_PAD_ZEROS 0
#define EMO_PAD_BORDER 1
#define EMO_PAD_REFLECTION 2
#endif
#else
#include "common.h"
#define GS_FN __device__ __forceinline__
#define GS_MFN __device__ __forceinline__
#define gs_fmul(a, b) __fmul_rn((a), (b))
#define gs_fadd(a, b) __fadd_rn((a), (b))
#define gs_fsub(a, b) __fsub_rn((a), (b))
#define gs_fdiv(a, b) __fdiv_rn((a), (b))
#define gs_fma(a, b, c) __fmaf_rn((a), (b), (c))
#endif

namespace gs3d {

// where the sampling coordinate of an output voxel comes from
enum { MODE_GRID = 0,    // explicit grid [N,Do,Ho,Wo,3]
       MODE_THETA = 1,   // head-pose affine of the identity lattice (a2: notebooks/infer.py:441-444, :583-588)
       MODE_DELTA = 2 }; // identity lattice + planar deltas [N,3,Do,Ho,Wo]: WarpGenerator's
                         // warp = (identity_grid + deltas).permute(0,2,3,4,1)  (warp_generator_resnet.py:178)

template <int PAD>
GS_FN float source_index(float g, int size) {
  // grid_sampler_unnormalize, align_corners=False: ((coord + 1) * size - 1) / 2
  float c = gs_fdiv(gs_fsub(gs_fmul(gs_fadd(g, 1.0f), (float)size), 1.0f), 2.0f);
  if (PAD == EMO_PAD_BORDER) {
    const float lim = (float)(size - 1);
    c = (c < 0.0f) ? 0.0f : c;           // std::max(in, 0)
    c = (c < lim) ? c : lim;             // std::min(lim, .)
  } else if (PAD == EMO_PAD_REFLECTION) {
    // reflect_coordinates(c, twice_low=-1, twice_high=2*size-1)
    const float mn = -0.5f;
    const float span = (float)size;
    float in = fabsf(gs_fsub(c, mn));
    float extra = fmodf(in, span);
    int flips = (int)floorf(gs_fdiv(in, span));
    c = (flips % 2 == 0) ? gs_fadd(extra, mn) : gs_fadd(gs_fsub(span, extra), mn);
    const float lim = (float)(size - 1);
    c = (c < 0.0f) ? 0.0f : c;
    c = (c < lim) ? c : lim;
  }
  return c;
}

// identity_grid_3d.bmm(theta[:, :3]^T): k-ordered fma chain starting from 0 (what the reference's GEMM does)
GS_FN float affine_row(const float* __restrict__ t, float u, float v, float w) {
  float acc = gs_fmul(u, t[0]);
  acc = gs_fma(v, t[1], acc);
  acc = gs_fma(w, t[2], acc);
  acc = gs_fma(1.0f, t[3], acc);
  return acc;
}

// coordinate of output voxel (x, y, z) = linear index vox of sample n
template <int MODE>
GS_FN void load_coord_xyz(const float* __restrict__ grid, const float* __restrict__ theta,
                          const float* __restrict__ lin_x, const float* __restrict__ lin_y,
                          const float* __restrict__ lin_z, int n, int x, int y, int z, int vox, int nvox,
                          float& gx, float& gy, float& gz) {
  if (MODE == MODE_GRID) {
    const float* g = grid + ((long)n * nvox + vox) * 3;
    gx = g[0]; gy = g[1]; gz = g[2];
  } else {
    const float u = lin_x[x], v = lin_y[y], w = lin_z[z];
    if (MODE == MODE_THETA) {
      const float* t = theta + (long)n * 12;
      gx = affine_row(t + 0, u, v, w);
      gy = affine_row(t + 4, u, v, w);
      gz = affine_row(t + 8, u, v, w);
    } else {
      const float* d = grid + (long)n * 3 * nvox + vox;
      gx = gs_fadd(u, d[0]);
      gy = gs_fadd(v, d[nvox]);
      gz = gs_fadd(w, d[2L * nvox]);
    }
  }
}

template <int MODE>
GS_FN void load_coord(const float* __restrict__ grid, const float* __restrict__ theta,
                      const float* __restrict__ lin_x, const float* __restrict__ lin_y,
                      const float* __restrict__ lin_z, int n, int vox, int nvox, int Ho, int Wo,
                      float& gx, float& gy, float& gz) {
  const int x = vox % Wo;
  const int y = (vox / Wo) % Ho;
  const int z = vox / (Wo * Ho);
  load_coord_xyz<MODE>(grid, theta, lin_x, lin_y, lin_z, n, x, y, z, vox, nvox, gx, gy, gz);
}

// Floor corner (x0, y0, z0) and the 8 corner weights of one sample point.  Non-finite / absurdly large coordinates are
// mapped to a point far outside the volume: every corner out of range (ATen's integer cast would be UB there).
template <int PAD>
GS_FN void corner_weights(float gx, float gy, float gz, int D, int H, int W, int& x0, int& y0, int& z0, float (&w)[8]) {
  float ix = source_index<PAD>(gx, W);
  float iy = source_index<PAD>(gy, H);
  float iz = source_index<PAD>(gz, D);
  const bool sane = (fabsf(ix) < 1.0e9f) && (fabsf(iy) < 1.0e9f) && (fabsf(iz) < 1.0e9f);
  if (!sane) { ix = -100.0f; iy = -100.0f; iz = -100.0f; }
  const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
  const float x1f = gs_fadd(x0f, 1.0f), y1f = gs_fadd(y0f, 1.0f), z1f = gs_fadd(z0f, 1.0f);
  const float wx0 = gs_fsub(x1f, ix), wx1 = gs_fsub(ix, x0f);
  const float wy0 = gs_fsub(y1f, iy), wy1 = gs_fsub(iy, y0f);
  const float wz0 = gs_fsub(z1f, iz), wz1 = gs_fsub(iz, z0f);
  x0 = (int)x0f; y0 = (int)y0f; z0 = (int)z0f;
  const float wxy00 = gs_fmul(wx0, wy0), wxy10 = gs_fmul(wx1, wy0);
  const float wxy01 = gs_fmul(wx0, wy1), wxy11 = gs_fmul(wx1, wy1);
  w[0] = gs_fmul(wxy00, wz0); w[1] = gs_fmul(wxy10, wz0);
  w[2] = gs_fmul(wxy01, wz0); w[3] = gs_fmul(wxy11, wz0);
  w[4] = gs_fmul(wxy00, wz1); w[5] = gs_fmul(wxy10, wz1);
  w[6] = gs_fmul(wxy01, wz1); w[7] = gs_fmul(wxy11, wz1);
}

// XCD-aware block order (cdna_hip_programming.md T1).  The dispatcher places block b on XCD b % 8 and each XCD has a private
// 4 MiB L2: remapped, XCD k walks one contiguous eighth of the linear work order.  Bijective for any total; a statement
// about speed only, never about correctness.
GS_FN int xcd_remap(int b, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

}  // namespace gs3d
