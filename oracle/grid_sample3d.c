/* TEST INFRASTRUCTURE ONLY -- plain-C restatement (the oracle) of the 3-D trilinear grid_sample the reference
 * calls through torch (models/stage_1/volumetric_avatar/va.py:264-265 -> F.grid_sample -> ATen grid_sampler_3d,
 * CPU path; helper semantics from ATen/native/GridSampler.h:27-36 unnormalize, :58-60 clip, :89-106 reflect;
 * align_corners=False, mode bilinear/trilinear).  Scalar, single-threaded, fp32, compiled with
 * -ffp-contract=off so that no multiply-add is fused (ATen's scalar CPU kernel does not fuse either).
 * Pinned bit-exactly against torch's CPU F.grid_sample by tests/test_oracle.py.
 *
 * Also restates the reference's rotation-warp construction (va.py:101-105 identity_grid_3d +
 * notebooks/infer.py:583-588 grid.bmm(theta[:, :3]^T)) as a k-ordered fma chain (what the CPU GEMM computes).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

enum { PAD_ZEROS = 0, PAD_BORDER = 1, PAD_REFLECTION = 2 };

static float clip_coord(float in, int size) {
  const float lim = (float)(size - 1);
  float m = (in < 0.0f) ? 0.0f : in;      /* std::max(in, 0)   */
  return (m < lim) ? m : lim;             /* std::min(lim, m)  */
}

static float reflect_coord(float in, int twice_low, int twice_high) {
  if (twice_low == twice_high) return 0.0f;
  float mn = (float)twice_low / 2;
  float span = (float)(twice_high - twice_low) / 2;
  in = fabsf(in - mn);
  float extra = fmodf(in, span);
  int flips = (int)floorf(in / span);
  if (flips % 2 == 0) return extra + mn;
  return span - extra + mn;
}

static float source_index(float g, int size, int pad) {
  float c = ((g + 1) * size - 1) / 2;
  if (pad == PAD_BORDER) c = clip_coord(c, size);
  else if (pad == PAD_REFLECTION) c = clip_coord(reflect_coord(c, -1, 2 * size - 1), size);
  return c;
}

static int inb(long z, long y, long x, int D, int H, int W) {
  return z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
}

/* vol [Nv,C,D,H,W] (vol_batch_stride elements between volumes, 0 = shared), grid [N,Do,Ho,Wo,3], out [N,C,Do,Ho,Wo] */
int oracle_grid_sample3d_f32(const float* vol, const float* grid, float* out, int N, int C, int D, int H, int W,
                             int Do, int Ho, int Wo, int64_t vol_batch_stride, int pad) {
  const long DHW = (long)D * H * W, nvox = (long)Do * Ho * Wo;
  for (int n = 0; n < N; ++n) {
    const float* v = vol + (long)n * vol_batch_stride;
    for (long p = 0; p < nvox; ++p) {
      const float* g = grid + ((long)n * nvox + p) * 3;
      float ix = source_index(g[0], W, pad), iy = source_index(g[1], H, pad), iz = source_index(g[2], D, pad);
      if (!(fabsf(ix) < 1.0e9f) || !(fabsf(iy) < 1.0e9f) || !(fabsf(iz) < 1.0e9f)) { ix = iy = iz = -100.0f; }
      long x0 = (long)floorf(ix), y0 = (long)floorf(iy), z0 = (long)floorf(iz);
      long x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
      float wx0 = (float)x1 - ix, wx1 = ix - (float)x0;
      float wy0 = (float)y1 - iy, wy1 = iy - (float)y0;
      float wz0 = (float)z1 - iz, wz1 = iz - (float)z0;
      float tnw = wx0 * wy0 * wz0, tne = wx1 * wy0 * wz0, tsw = wx0 * wy1 * wz0, tse = wx1 * wy1 * wz0;
      float bnw = wx0 * wy0 * wz1, bne = wx1 * wy0 * wz1, bsw = wx0 * wy1 * wz1, bse = wx1 * wy1 * wz1;
      for (int c = 0; c < C; ++c) {
        const float* vc = v + (long)c * DHW;
        float acc = 0.0f;
        if (inb(z0, y0, x0, D, H, W)) acc += vc[(z0 * H + y0) * W + x0] * tnw;
        if (inb(z0, y0, x1, D, H, W)) acc += vc[(z0 * H + y0) * W + x1] * tne;
        if (inb(z0, y1, x0, D, H, W)) acc += vc[(z0 * H + y1) * W + x0] * tsw;
        if (inb(z0, y1, x1, D, H, W)) acc += vc[(z0 * H + y1) * W + x1] * tse;
        if (inb(z1, y0, x0, D, H, W)) acc += vc[(z1 * H + y0) * W + x0] * bnw;
        if (inb(z1, y0, x1, D, H, W)) acc += vc[(z1 * H + y0) * W + x1] * bne;
        if (inb(z1, y1, x0, D, H, W)) acc += vc[(z1 * H + y1) * W + x0] * bsw;
        if (inb(z1, y1, x1, D, H, W)) acc += vc[(z1 * H + y1) * W + x1] * bse;
        out[((long)n * C + c) * nvox + p] = acc;
      }
    }
  }
  return 0;
}

/* rotation warp: grid[n,z,y,x,j] = fma-chain_k( lattice_k * theta[n,j,k] ) with lattice = (lin_x[x], lin_y[y], lin_z[z], 1) */
int oracle_affine_grid3d_f32(const float* theta /*[N,3,4]*/, const float* lin_x, const float* lin_y, const float* lin_z,
                             float* grid /*[N,Do,Ho,Wo,3]*/, int N, int Do, int Ho, int Wo) {
  for (int n = 0; n < N; ++n)
    for (int z = 0; z < Do; ++z)
      for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x) {
          float* g = grid + ((((long)n * Do + z) * Ho + y) * Wo + x) * 3;
          for (int j = 0; j < 3; ++j) {
            const float* t = theta + ((long)n * 3 + j) * 4;
            float acc = lin_x[x] * t[0];
            acc = fmaf(lin_y[y], t[1], acc);
            acc = fmaf(lin_z[z], t[2], acc);
            acc = fmaf(1.0f, t[3], acc);
            g[j] = acc;
          }
        }
  return 0;
}
