// Resampling + pointwise helpers of the hot path for gfx950 (all HBM-bound, one pass each).
//
//   emo_upsample_trilinear_f32  F.interpolate(x, scale_factor=(sd,sh,sw), mode='trilinear') with align_corners=False,
//                               scale factors 1 or 2 per axis: WarpGenerator.forward
//                               (networks/volumetric_avatar/warp_generator_resnet.py:163-166) and Unet3D.forward
//                               (unet_3d.py:223,269-272).
//   emo_avgpool_f32             nn.AvgPool3d / AvgPool2d with kernel == stride in 1..16 per axis (also the integer-window
//                               AdaptiveAvgPool2d of the embedders: identity_embedder.py:33, expression_embedder.py:394,408)
//                               (downsampling_layers['avgpool'(_3d)], utils.py:962-967; warp_generator_resnet.py:118,
//                               unet_3d.py:84-86,192-193, local_encoder.py via ResBlock stride 2).
//   emo_add_f32                 out = (a + b[i % period]) * alpha  (Unet3D skip sum unet_3d.py:281; embed mix va.py:857).
#include "common.h"

namespace {

// ATen area_pixel_compute_source_index(scale = 1/scale_factor, dst, align_corners=False, cubic=False):
//   src = scale * (dst + 0.5) - 0.5, clamped below at 0
__device__ __forceinline__ void lin_coeff(int o, int in_size, int factor, int& i0, int& i1, float& l0, float& l1) {
  if (factor == 1) { i0 = o; i1 = o; l0 = 1.0f; l1 = 0.0f; return; }
  float src = 0.5f * ((float)o + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void upsample_trilinear_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                  long NC, int D, int H, int W, int fd, int fh, int fw) {
  const int Do = D * fd, Ho = H * fh, Wo = W * fw;
  const long total = NC * Do * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    long r = i / Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long nc = r / Do;
    int z0, z1, y0, y1, x0, x1;
    float lz0, lz1, ly0, ly1, lx0, lx1;
    lin_coeff(zo, D, fd, z0, z1, lz0, lz1);
    lin_coeff(yo, H, fh, y0, y1, ly0, ly1);
    lin_coeff(xo, W, fw, x0, x1, lx0, lx1);
    const float* p = x + nc * (long)D * H * W;
    const long HW = (long)H * W;
    const float v000 = p[z0 * HW + y0 * W + x0], v001 = p[z0 * HW + y0 * W + x1];
    const float v010 = p[z0 * HW + y1 * W + x0], v011 = p[z0 * HW + y1 * W + x1];
    const float v100 = p[z1 * HW + y0 * W + x0], v101 = p[z1 * HW + y0 * W + x1];
    const float v110 = p[z1 * HW + y1 * W + x0], v111 = p[z1 * HW + y1 * W + x1];
    // ATen upsample_trilinear3d: t0 * (h0 * (w0 * v000 + w1 * v001) + h1 * (w0 * v010 + w1 * v011)) + t1 * (...)
    const float a = lz0 * (ly0 * (lx0 * v000 + lx1 * v001) + ly1 * (lx0 * v010 + lx1 * v011));
    const float b = lz1 * (ly0 * (lx0 * v100 + lx1 * v101) + ly1 * (lx0 * v110 + lx1 * v111));
    out[i] = a + b;
  }
}

// The same for a width factor of 2 and an even input width (every call of the driver pass).  The generic kernel spends three
// 64-bit divisions, eight 4-byte loads and a 4-byte store on every output and is bound by the CU's vector-memory instruction
// rate at a fifth of the HBM rate of its bytes.  Here one thread produces a BLOCK of outputs that share their inputs: four
// consecutive columns 4m .. 4m + 3 (input columns 2m - 1 .. 2m + 2) x the output rows 2k - 1, 2k (both interpolate the input
// rows k - 1, k; a factor-1 axis: one row) x the same pairing in depth -- up to 16 outputs from 16 loads (8 where the depth or
// the height factor is 1: both taps of that axis are the same row), written with four 16-byte stores.  Same coefficients and
// order of operations per output as the generic kernel (ATen's).
//   * First / last quad of a row: the input columns are clamped into the row, which IS the tap of the generic kernel there
//     (last column: x0 = x1 = W - 1); the first output of a row (source index clamped to 0: taps 0 and 1, weights 1 and 0) takes
//     its two values one register further right.  Round 4 sent those two quads of every row through the one-output-at-a-time
//     code -- 2 lanes in 16, so EVERY wave ran both paths, the second one with 128 scalar loads: 1.7 TB/s of the kernel's bytes.
//   * The first / last pair of an axis has one valid output (2k - 1 = -1, or 2k = 2 * in): its taps are used; two valid outputs
//     of a pair always share theirs (src = k - 0.75 and k - 0.25: both between the rows k - 1 and k).
//   * Work is laid out as runs of `cr` consecutive (sample, channel) volumes x `split` slices of a run's thread blocks
//     (grid = runs x split).  STATS: a run is a GroupNorm group (cr = C / G channels: one contiguous reduction domain of the
//     OUTPUT), and every block leaves the fp64 (sum, sum of squares) of the outputs it produced in partial[run][slice] -- the
//     layout gn_partial_kernel (groupnorm.hip) writes, so the norm in front of the next convolution needs no pass of its own
//     over the upsampled tensor (WarpGenerator: warp_generator_resnet.py:163-166 -> ResBlock3d's first norm).
constexpr int UPS_MAX_SPLIT = 64;         // == GN_MAX_SPLIT (groupnorm.hip): the partial-sum layout [run][64][2]

template <bool STATS>
__global__ __launch_bounds__(256) void upsample_trilinear_w2_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                     unsigned qrun, unsigned per, unsigned cr, int D, int H,
                                                                     int W, int fd, int fh, double* __restrict__ partial) {
  const int Do = D * fd, Ho = H * fh;
  const unsigned Wq = (unsigned)W >> 1;                      // quads per output row: 2 W / 4
  const unsigned Ky = fh == 2 ? H + 1 : H, Kz = fd == 2 ? D + 1 : D;   // pairs (2k - 1, 2k), k = 0 .. in; or single rows
  const long HW = (long)H * W;
  const unsigned run = blockIdx.x, sp = blockIdx.y;
  const unsigned lo = sp * per, hi = lo + per < qrun ? lo + per : qrun;
  double s_sum = 0.0, s_sq = 0.0;
  for (unsigned q = lo + threadIdx.x; q < hi; q += 256u) {
    const unsigned xq = q % Wq;
    unsigned r = q / Wq;
    const unsigned ky = r % Ky; r /= Ky;
    const unsigned kz = r % Kz;
    const long nc = (long)run * cr + r / Kz;
    // the (up to two) outputs of the pair along y and z: first = 2k - 1 (factor 2) or k (factor 1)
    const int ya = fh == 2 ? 2 * (int)ky - 1 : (int)ky, za = fd == 2 ? 2 * (int)kz - 1 : (int)kz;
    const int ny = fh == 2 ? 2 : 1, nz = fd == 2 ? 2 : 1;
    const float* p = x + nc * D * HW;
    float* const o = out + nc * Do * Ho * (2l * W) + 4 * xq;
    int yi0[2], yi1[2], zi0[2], zi1[2];
    float yl0[2], yl1[2], zl0[2], zl1[2];
    bool yv[2], zv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int yo = ya + e, zo = za + e;
      yv[e] = e < ny && yo >= 0 && yo < Ho;
      zv[e] = e < nz && zo >= 0 && zo < Do;
      lin_coeff(yv[e] ? yo : 0, H, fh, yi0[e], yi1[e], yl0[e], yl1[e]);
      lin_coeff(zv[e] ? zo : 0, D, fd, zi0[e], zi1[e], zl0[e], zl1[e]);
    }
    const int xb = 2 * (int)xq - 1;
    const int ey = yv[0] ? 0 : 1, ez = zv[0] ? 0 : 1;         // (a pair has at least one valid output)
    int col[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int c = xb + k; col[k] = c < 0 ? 0 : (c > W - 1 ? W - 1 : c); }
    const float* r00 = p + zi0[ez] * HW + (long)yi0[ey] * W;
    float a00[4], a01[4], a10[4], a11[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a00[k] = r00[col[k]];
    if (fh == 2) {
      const float* r01 = p + zi0[ez] * HW + (long)yi1[ey] * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) a01[k] = r01[col[k]];
    } else {
#pragma unroll
      // (factor 1 along y: the generic kernel / ATen form 1 * v[i0] + 0 * v[i1] with i1 = i0 + 1; here the second tap is a copy of
      // the first, i.e. v[i1] is never loaded.  Same bits for FINITE data; an Inf / NaN in v[i1] would propagate there (0 * Inf =
      // NaN) and does not here -- the "same bits as ATen" statement of this kernel holds for finite inputs, which is what a
      // GroupNorm'ed activation tensor is)
      for (int k = 0; k < 4; ++k) a01[k] = a00[k];
    }
    if (fd == 2) {
      const float* r10 = p + zi1[ez] * HW + (long)yi0[ey] * W;
#pragma unroll
      for (int k = 0; k < 4; ++k) a10[k] = r10[col[k]];
      if (fh == 2) {
        const float* r11 = p + zi1[ez] * HW + (long)yi1[ey] * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) a11[k] = r11[col[k]];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) a11[k] = a10[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { a10[k] = a00[k]; a11[k] = a01[k]; }
    }
    float lx0[4], lx1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      lin_coeff(4 * (int)xq + j, W, 2, x0, x1, lx0[j], lx1[j]);
    }
    // the first output of a row: taps (0, 1) = registers 1, 2 (register 0 holds the clamped column -1)
    const bool le = xq == 0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!(zv[c] && yv[e])) continue;
        float res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int u = (j + 1) >> 1;
          float p00 = a00[u], q00 = a00[u + 1], p01 = a01[u], q01 = a01[u + 1];
          float p10 = a10[u], q10 = a10[u + 1], p11 = a11[u], q11 = a11[u + 1];
          if (j == 0) {
            p00 = le ? a00[1] : p00; q00 = le ? a00[2] : q00; p01 = le ? a01[1] : p01; q01 = le ? a01[2] : q01;
            p10 = le ? a10[1] : p10; q10 = le ? a10[2] : q10; p11 = le ? a11[1] : p11; q11 = le ? a11[2] : q11;
          }
          const float a = zl0[c] * (yl0[e] * (lx0[j] * p00 + lx1[j] * q00) + yl1[e] * (lx0[j] * p01 + lx1[j] * q01));
          const float b = zl1[c] * (yl0[e] * (lx0[j] * p10 + lx1[j] * q10) + yl1[e] * (lx0[j] * p11 + lx1[j] * q11));
          res[j] = a + b;
        }
        *reinterpret_cast<float4*>(o + ((long)(za + c) * Ho + (ya + e)) * (2l * W)) = make_float4(res[0], res[1], res[2], res[3]);
        if constexpr (STATS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double v = (double)res[j];
            s_sum += v;
            s_sq += v * v;
          }
        }
      }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_sum += __shfl_down(s_sum, o, 64); s_sq += __shfl_down(s_sq, o, 64); }
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s_sum; red[1][wave] = s_sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* pp = partial + ((long)run * UPS_MAX_SPLIT + sp) * 2;
      pp[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      pp[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
  }
}

// launch plan of the block kernel: runs x slices, a slice >= 2048 thread blocks of outputs where the run is that long
inline bool ups_w2_plan(long runs, long qrun, unsigned& split, unsigned& per) {
  if (runs < 1 || runs > 0x7fffffffL || qrun < 1 || qrun >= (1l << 31)) return false;
  long sp = (qrun + 2047) / 2048;
  sp = sp < 1 ? 1 : (sp > UPS_MAX_SPLIT ? UPS_MAX_SPLIT : sp);
  split = (unsigned)sp;
  per = (unsigned)((qrun + sp - 1) / sp);
  return true;
}

__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ out, long NC,
                                                      int D, int H, int W, int kd, int kh, int kw) {
  const int Do = D / kd, Ho = H / kh, Wo = W / kw;
  const long total = NC * Do * Ho * Wo;
  const float inv = 1.0f / (float)(kd * kh * kw);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    long r = i / Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long nc = r / Do;
    const float* p = x + nc * (long)D * H * W;
    float s = 0.0f;
    for (int a = 0; a < kd; ++a)
      for (int b = 0; b < kh; ++b)
        for (int c = 0; c < kw; ++c) s += p[((long)(zo * kd + a) * H + (yo * kh + b)) * W + (xo * kw + c)];
    out[i] = s * inv;
  }
}

// The same for a window width of 1 or 2 on rows of whole quads (every call of the driver pass: the depth pooling (2, 1, 1) of the
// WarpGenerator's last block, 268 MB in, and the (1, 2, 2) / (2, 2, 2) poolings of the source pass): one thread produces four
// consecutive outputs from 16-byte loads and stores them with one 16-byte store, 32-bit index arithmetic.  The sum of an output
// runs over (depth, row, column) of its window in that order, as in the one-output-per-thread kernel: the same bits.
template <int KW>
__global__ __launch_bounds__(256) void avgpool_x4_kernel(const float* __restrict__ x, float* __restrict__ out, unsigned quads,
                                                         int D, int H, int W, int kd, int kh) {
  const unsigned Do = D / kd, Ho = H / kh, Wq = (unsigned)(W / KW) >> 2;
  const float inv = 1.0f / (float)(kd * kh * KW);
  const long vol = (long)D * H * W;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < quads; i += gridDim.x * 256u) {
    const unsigned xq = i % Wq;
    unsigned r = i / Wq;
    const unsigned yo = r % Ho; r /= Ho;
    const unsigned zo = r % Do;
    const unsigned nc = r / Do;
    const float* p = x + nc * vol + 4 * KW * xq;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int a = 0; a < kd; ++a)
      for (int b = 0; b < kh; ++b) {
        const float4* row = reinterpret_cast<const float4*>(p + ((long)(zo * kd + a) * H + (yo * kh + b)) * W);
        const float4 v0 = row[0];
        if (KW == 1) {
          s[0] += v0.x; s[1] += v0.y; s[2] += v0.z; s[3] += v0.w;
        } else {
          const float4 v1 = row[1];
          s[0] += v0.x; s[0] += v0.y; s[1] += v0.z; s[1] += v0.w;
          s[2] += v1.x; s[2] += v1.y; s[3] += v1.z; s[3] += v1.w;
        }
      }
    reinterpret_cast<float4*>(out)[i] = make_float4(s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv);
  }
}


__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n, long period, float alpha) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = (a[i] + b[i % period]) * alpha;
}

// F.interpolate(x, size=(Ho, Wo), mode='bilinear' | 'bicubic', align_corners=False) on 4-D tensors (ATen
// UpSampleBilinear2d / UpSampleBicubic2d: area_pixel_compute_source_index + cubic convolution, A = -0.75).
// Wrapper glue of the reference: bicubic resize of source / driver crops to image_size (notebooks/infer.py:399-401,
// :548-552), bilinear resize to output_size_s2 (notebooks/infer_s2.py:360-362).
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }

// one output element of F.interpolate(size=(Ho, Wo), align_corners=False) from the H x W window at p (row stride in floats)
__device__ __forceinline__ float resize2d_at(const float* __restrict__ p, long row_stride, int H, int W, float sh, float sw,
                                             int yo, int xo, int bicubic, int clamp01) {
  float sy = sh * ((float)yo + 0.5f) - 0.5f, sx = sw * ((float)xo + 0.5f) - 0.5f;
  if (!bicubic) {
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float v = ly0 * (lx0 * p[y0 * row_stride + x0] + lx1 * p[y0 * row_stride + x1]) +
                    ly1 * (lx0 * p[y1 * row_stride + x0] + lx1 * p[y1 * row_stride + x1]);
    return clamp01 ? fminf(fmaxf(v, 0.0f), 1.0f) : v;
  }
  const float fy = floorf(sy), fx = floorf(sx);
  const int iy = (int)fy, ix = (int)fx;
  const float ty = sy - fy, tx = sx - fx;
  const float A = -0.75f;
  const float wy[4] = {cubic2(ty + 1.0f, A), cubic1(ty, A), cubic1(1.0f - ty, A), cubic2(2.0f - ty, A)};
  const float wx[4] = {cubic2(tx + 1.0f, A), cubic1(tx, A), cubic1(1.0f - tx, A), cubic2(2.0f - tx, A)};
  float acc = 0.0f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int yy = iy - 1 + a;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    float row = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int xx = ix - 1 + b;
      xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
      row += p[yy * row_stride + xx] * wx[b];
    }
    acc += row * wy[a];
  }
  return clamp01 ? fminf(fmaxf(acc, 0.0f), 1.0f) : acc;   // bicubic overshoots: crop_image clips (infer.py:350)
}

__global__ __launch_bounds__(256) void resize2d_kernel(const float* __restrict__ x, long plane_stride, long row_stride,
                                                       float* __restrict__ out, long NC,
                                                       int H, int W, int Ho, int Wo, int bicubic, int clamp01) {
  const long total = NC * Ho * Wo;
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;   // scale = in/out when only `size` is given
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    const long r = i / Wo;
    const int yo = (int)(r % Ho);
    const long nc = r / Ho;
    out[i] = resize2d_at(x + nc * plane_stride, row_stride, H, W, sh, sw, yo, xo, bicubic, clamp01);
  }
}

// The same with one crop window PER SAMPLE (ABI 9; animate_frames' `windows`: the reference crops every frame around its own
// face box, notebooks/infer.py:301-352): win[n] = (x0, y0, w, h) of sample n inside its Hf x Wf frame, read from device memory,
// so that a batch of B frames is ONE launch instead of B (round 5: a Python loop of single-frame launches).  Per element the
// arithmetic is resize2d_kernel's on the window's first pixel: the two are bit-identical.
__global__ __launch_bounds__(256) void resize2d_windows_kernel(const float* __restrict__ x, long plane_stride, long row_stride,
                                                               const int* __restrict__ win, float* __restrict__ out, long N,
                                                               int C, int Ho, int Wo, int bicubic, int clamp01) {
  const long total = N * C * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    const long r = i / Wo;
    const int yo = (int)(r % Ho);
    const long nc = r / Ho;
    const int* const w4 = win + 4 * (nc / C);
    const int wx0 = w4[0], wy0 = w4[1], ww = w4[2], wh = w4[3];
    const float sh = (float)wh / (float)Ho, sw = (float)ww / (float)Wo;
    out[i] = resize2d_at(x + nc * plane_stride + (long)wy0 * row_stride + wx0, row_stride, wh, ww, sh, sw, yo, xo, bicubic, clamp01);
  }
}

// stage-2 glue (notebooks/infer_s2.py:365-375)
__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ img, const float* __restrict__ mask,
                                                       float* __restrict__ out, long N, int C, long HW) {
  const long total = N * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i % HW;
    const long n = i / (HW * C);
    out[i] = img[i] * mask[n * HW + p];
  }
}

__global__ __launch_bounds__(256) void stage2_compose_kernel(const float* __restrict__ img, const float* __restrict__ add,
                                                             const float* __restrict__ mask,
                                                             const float* __restrict__ face_mask,
                                                             float* __restrict__ out, long N, int C, long HW) {
  const long total = N * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i % HW;
    const long n = i / (HW * C);
    const float m = mask[n * HW + p] * face_mask[n * HW + p];
    float v = img[i] + add[i] * m;
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    out[i] = v;
  }
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int emo_upsample_trilinear_f32(const float* x, float* out, int64_t NC, int D, int H, int W, int fd, int fh,
                                          int fw, void* stream) {
  if (!x || !out || NC <= 0 || D <= 0 || H <= 0 || W <= 0) return EMO_ERR_BAD_ARG;
  if ((fd != 1 && fd != 2) || (fh != 1 && fh != 2) || (fw != 1 && fw != 2)) return EMO_ERR_UNSUPPORTED;
  const long total = NC * D * fd * H * fh * W * fw;
  const long qvol = (long)(fd == 2 ? D + 1 : D) * (fh == 2 ? H + 1 : H) * (W / 2);      // thread blocks of outputs per volume
  long cr = 1;                                                                         // volumes per run: small ones in groups
  while (qvol * cr < 2048 && NC % (2 * cr) == 0) cr *= 2;
  unsigned split, per;
  if (fw == 2 && (W & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ups_w2_plan(NC / cr, qvol * cr, split, per)) {
    hipLaunchKernelGGL(upsample_trilinear_w2_kernel<false>, dim3((unsigned)(NC / cr), split), dim3(256), 0, (hipStream_t)stream,
                       x, out, (unsigned)(qvol * cr), per, (unsigned)cr, D, H, W, fd, fh, (double*)nullptr);
    return emo_launch_status();
  }
  hipLaunchKernelGGL(upsample_trilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out,
                     (long)NC, D, H, W, fd, fh, fw);
  return emo_launch_status();
}

// The upsampling with the GroupNorm statistics of its OUTPUT reduced on the way: out as emo_upsample_trilinear_f32, and in
// `partial` ([N * G][64][2] doubles: emo_groupnorm_workspace_bytes(N, G)) the (sum, sum of squares) slices of every (sample,
// group) in the layout emo_groupnorm_affine_from_sums_f32 (groupnorm.hip) finishes; *split_out = the slices written per group.
// Width factor 2, an even input width and a 16-byte aligned output only (EMO_ERR_UNSUPPORTED otherwise: the caller runs the
// two operations one after the other).
extern "C" int emo_upsample_trilinear_gn_sums_f32(const float* x, float* out, int N, int C, int G, int D, int H, int W, int fd,
                                                  int fh, int fw, void* partial, int64_t partial_bytes, int* split_out,
                                                  void* stream) {
  if (!x || !out || !partial || !split_out || N <= 0 || C <= 0 || G <= 0 || C % G || D <= 0 || H <= 0 || W <= 0) return EMO_ERR_BAD_ARG;
  if ((fd != 1 && fd != 2) || (fh != 1 && fh != 2) || fw != 2 || (W & 1)) return EMO_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return EMO_ERR_UNSUPPORTED;
  if (partial_bytes < (int64_t)N * G * UPS_MAX_SPLIT * 2 * (int64_t)sizeof(double)) return EMO_ERR_BAD_ARG;
  const unsigned cpg = (unsigned)(C / G);
  const long qrun = (long)cpg * (fd == 2 ? D + 1 : D) * (fh == 2 ? H + 1 : H) * (W / 2);
  unsigned split, per;
  if (!ups_w2_plan((long)N * G, qrun, split, per)) return EMO_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(upsample_trilinear_w2_kernel<true>, dim3((unsigned)(N * G), split), dim3(256), 0, (hipStream_t)stream, x, out,
                     (unsigned)qrun, per, cpg, D, H, W, fd, fh, reinterpret_cast<double*>(partial));
  *split_out = (int)split;
  return emo_launch_status();
}

extern "C" int emo_avgpool_f32(const float* x, float* out, int64_t NC, int D, int H, int W, int kd, int kh, int kw,
                               void* stream) {
  if (!x || !out || NC <= 0 || D <= 0 || H <= 0 || W <= 0) return EMO_ERR_BAD_ARG;
  if (kd < 1 || kh < 1 || kw < 1 || kd > 16 || kh > 16 || kw > 16) return EMO_ERR_UNSUPPORTED;
  if (D % kd || H % kh || W % kw) return EMO_ERR_UNSUPPORTED;
  const long total = NC * (D / kd) * (H / kh) * (W / kw);
  if ((kw == 1 || kw == 2) && W % (4 * kw) == 0 && total / 4 < (1l << 32) &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const long quads = total / 4;
    if (kw == 1)
      hipLaunchKernelGGL(avgpool_x4_kernel<1>, dim3(grid_for(quads)), dim3(256), 0, (hipStream_t)stream, x, out, (unsigned)quads, D, H, W, kd, kh);
    else
      hipLaunchKernelGGL(avgpool_x4_kernel<2>, dim3(grid_for(quads)), dim3(256), 0, (hipStream_t)stream, x, out, (unsigned)quads, D, H, W, kd, kh);
    return emo_launch_status();
  }
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, (long)NC, D, H,
                     W, kd, kh, kw);
  return emo_launch_status();
}

extern "C" int emo_add_f32(const float* a, const float* b, float* out, int64_t n, int64_t period, float alpha,
                           void* stream) {
  if (!a || !b || !out || n <= 0 || period <= 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n,
                     (long)period, alpha);
  return emo_launch_status();
}

extern "C" int emo_mul_mask_f32(const float* img, const float* mask, float* out, int N, int C, int64_t HW, void* stream) {
  if (!img || !mask || !out || N <= 0 || C <= 0 || HW <= 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(mul_mask_kernel, dim3(grid_for((long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, img, mask, out,
                     (long)N, C, (long)HW);
  return emo_launch_status();
}

extern "C" int emo_stage2_compose_f32(const float* img, const float* add, const float* mask, const float* face_mask,
                                      float* out, int N, int C, int64_t HW, void* stream) {
  if (!img || !add || !mask || !face_mask || !out || N <= 0 || C <= 0 || HW <= 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(stage2_compose_kernel, dim3(grid_for((long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, img,
                     add, mask, face_mask, out, (long)N, C, (long)HW);
  return emo_launch_status();
}

extern "C" int emo_resize2d_f32(const float* x, int64_t plane_stride, int64_t row_stride, float* out, int64_t NC, int H,
                                int W, int Ho, int Wo, int bicubic, int clamp01, void* stream) {
  if (!x || !out || NC <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || row_stride < W || plane_stride < 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(resize2d_kernel, dim3(grid_for(NC * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)plane_stride, (long)row_stride, out, (long)NC, H, W, Ho, Wo, bicubic, clamp01);
  return emo_launch_status();
}

extern "C" int emo_resize2d_windows_f32(const float* x, int64_t plane_stride, int64_t row_stride, const int* windows, float* out,
                                        int N, int C, int Ho, int Wo, int bicubic, int clamp01, void* stream) {
  if (!x || !out || !windows || N <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || row_stride <= 0 || plane_stride < 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(resize2d_windows_kernel, dim3(grid_for((long)N * C * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)plane_stride, (long)row_stride, windows, out, (long)N, C, Ho, Wo, bicubic, clamp01);
  return emo_launch_status();
}
