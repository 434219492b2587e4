// Elementwise / gather kernels of the embedder ResNets and of ExpressionEmbed's face alignment for gfx950
// (SURVEY.md section 8f-1).  All HBM-bound, one output element per thread, coalesced along the innermost image axis.
#include "common.h"

namespace {

__host__ __device__ inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 65536 * 16 ? 65536 * 16 : b));
}

// nn.MaxPool2d(k, stride, pad) (torchvision ResNet stem: 3, 2, 1; padding is -inf) on relu(x * scale + shift): the stem's
// norm-apply + ReLU (identity_embedder.py:60-63 / expression_embedder.py:426-429) is folded into the pooling read.
__global__ __launch_bounds__(256) void maxpool2d_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ out,
                                                        long NC, int H, int W, int Ho, int Wo, int k, int stride, int pad,
                                                        int relu) {
  const long total = NC * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    const long r = i / Wo;
    const int yo = (int)(r % Ho);
    const long nc = r / Ho;
    const float* p = x + nc * (long)H * W;
    const float sc = scale ? scale[nc] : 1.0f, sh = scale ? shift[nc] : 0.0f;
    float m = -INFINITY;
    for (int a = 0; a < k; ++a) {
      const int yy = yo * stride - pad + a;
      if (yy < 0 || yy >= H) continue;
      for (int b = 0; b < k; ++b) {
        const int xx = xo * stride - pad + b;
        if (xx < 0 || xx >= W) continue;
        float v = __fmaf_rn(p[(long)yy * W + xx], sc, sh);
        if (relu) v = fmaxf(v, 0.0f);
        m = fmaxf(m, v);
      }
    }
    out[i] = m;
  }
}

// tail of a post-activation residual block (torchvision BasicBlock / Bottleneck.forward):
//   out = relu( (a * sa + ta) + (b * sb + tb) )      a = last conv of the block, (sa, ta) its norm as per-(n,c) affine;
//   b = block input (sb null) or the downsample conv output with its norm (sb, tb)
__global__ __launch_bounds__(256) void affine_add_relu_kernel(const float* __restrict__ a, const float* __restrict__ sa,
                                                              const float* __restrict__ ta, const float* __restrict__ b,
                                                              const float* __restrict__ sb, const float* __restrict__ tb,
                                                              float* __restrict__ out, long NC, long S, int relu) {
  const long total = NC * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long nc = i / S;
    float v = sa ? __fmaf_rn(a[i], sa[nc], ta[nc]) : a[i];
    if (b) v += sb ? __fmaf_rn(b[i], sb[nc], tb[nc]) : b[i];
    out[i] = relu ? fmaxf(v, 0.0f) : v;
  }
}

// F.grid_sample on 4-D input, bilinear, zeros padding, align_corners=False (ATen GridSamplerKernel.cpp ApplyGridSample:
// ix = (gx + 1) * (W / 2) - 0.5; weights from floor(); out-of-range corners contribute 0), with the affine grid of
// ExpressionEmbed computed in the kernel: grid(yo, xo) = A[n] @ (lin[xo], lin[yo], 1)  -- expression_embedder.py:221-231
// (`identity_grid.bmm(inv_theta_2d^T)` then F.grid_sample), so the [N,128,128,2] warp is never written unless asked for.
__global__ __launch_bounds__(256) void grid_sample2d_kernel(const float* __restrict__ img, const float* __restrict__ grid,
                                                            const float* __restrict__ theta, const float* __restrict__ lin,
                                                            float* __restrict__ out, float* __restrict__ grid_out, int N,
                                                            int C, int H, int W, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xo = (int)(i % Wo);
    const long r = i / Wo;
    const int yo = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float gx, gy;
    if (grid) {
      gx = grid[2 * i];
      gy = grid[2 * i + 1];
    } else {
      const float* t = theta + n * 6;
      const float u = lin[xo], v = lin[yo];
      gx = __fmaf_rn(1.0f, t[2], __fmaf_rn(v, t[1], __fmul_rn(u, t[0])));
      gy = __fmaf_rn(1.0f, t[5], __fmaf_rn(v, t[4], __fmul_rn(u, t[3])));
    }
    if (grid_out) {
      grid_out[2 * i] = gx;
      grid_out[2 * i + 1] = gy;
    }
    const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), 0.5f * (float)W), 0.5f);
    const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), 0.5f * (float)H), 0.5f);
    const float fx = floorf(ix), fy = floorf(iy);
    const float w = __fsub_rn(ix, fx), e = __fsub_rn(1.0f, w);
    const float s_ = __fsub_rn(iy, fy), n_ = __fsub_rn(1.0f, s_);   // s_: weight of the south row, n_: north row
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool okx0 = x0 >= 0 && x0 < W, okx1 = x1 >= 0 && x1 < W, oky0 = y0 >= 0 && y0 < H, oky1 = y1 >= 0 && y1 < H;
    const float wnw = __fmul_rn(n_, e), wne = __fmul_rn(n_, w), wsw = __fmul_rn(s_, e), wse = __fmul_rn(s_, w);
    for (int c = 0; c < C; ++c) {
      const float* p = img + ((long)n * C + c) * H * W;
      const float vnw = (oky0 && okx0) ? p[(long)y0 * W + x0] : 0.0f;
      const float vne = (oky0 && okx1) ? p[(long)y0 * W + x1] : 0.0f;
      const float vsw = (oky1 && okx0) ? p[(long)y1 * W + x0] : 0.0f;
      const float vse = (oky1 && okx1) ? p[(long)y1 * W + x1] : 0.0f;
      out[((long)n * C + c) * Ho * Wo + (long)yo * Wo + xo] =
          __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(vnw, wnw), __fmul_rn(vne, wne)), __fmul_rn(vsw, wsw)), __fmul_rn(vse, wse));
    }
  }
}

}  // namespace

extern "C" int emo_maxpool2d_f32(const float* x, const float* scale, const float* shift, float* out, int64_t NC, int H,
                                 int W, int k, int stride, int pad, int relu, void* stream) {
  if (!x || !out || NC <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k) return EMO_ERR_BAD_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMO_ERR_BAD_ARG;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(maxpool2d_kernel, dim3(grid_for(NC * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                     out, (long)NC, H, W, Ho, Wo, k, stride, pad, relu);
  return emo_launch_status();
}

extern "C" int emo_affine_add_relu_f32(const float* a, const float* sa, const float* ta, const float* b, const float* sb,
                                       const float* tb, float* out, int64_t NC, int64_t S, int relu, void* stream) {
  if (!a || !out || NC <= 0 || S <= 0) return EMO_ERR_BAD_ARG;
  if ((sa == nullptr) != (ta == nullptr) || (sb == nullptr) != (tb == nullptr) || (sb && !b)) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(affine_add_relu_kernel, dim3(grid_for(NC * S)), dim3(256), 0, (hipStream_t)stream, a, sa, ta, b, sb,
                     tb, out, (long)NC, (long)S, relu);
  return emo_launch_status();
}

extern "C" int emo_grid_sample2d_f32(const float* img, const float* grid, const float* theta, const float* lin, float* out,
                                     float* grid_out, int N, int C, int H, int W, int Ho, int Wo, void* stream) {
  if (!img || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return EMO_ERR_BAD_ARG;
  if ((grid == nullptr) == (theta == nullptr)) return EMO_ERR_BAD_ARG;   // exactly one grid source
  if (theta && (!lin || Ho != Wo)) return EMO_ERR_BAD_ARG;               // the reference lattice is square (one linspace)
  hipLaunchKernelGGL(grid_sample2d_kernel, dim3(grid_for((long)N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, img,
                     grid, theta, lin, out, grid_out, N, C, H, W, Ho, Wo);
  return emo_launch_status();
}
