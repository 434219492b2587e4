// GroupNorm statistics -> per-(sample, channel) affine, for gfx950.  SURVEY.md section 8 row a10.
//
// The reference normalises with nn.GroupNorm(32, C) (norm_layers['gn'|'gn_3d'],
// networks/volumetric_avatar/utils.py:953-957) or AdaptiveGroupNorm (utils.py:302-325) in front of every conv of a
// ResBlock.  Here the normalisation is never materialised: this file only reduces the statistics and emits
//     scale[n,c], shift[n,c]   such that   GN(x)[n,c,:] == x[n,c,:] * scale[n,c] + shift[n,c]
// and the consumer (conv_igemm.h input staging) applies them on the fly together with the ReLU.
//
// A group's (C/G) channels are adjacent in NC(D)HW memory, so the reduction domain of (n, g) is ONE contiguous
// run of L = (C/G)*S floats: kernel 1 streams it with 16-byte loads (HBM-bound), accumulating sum and
// sum-of-squares in fp64 (no cancellation issue for var = E[x^2] - mean^2), split over up to 64 blocks per run
// so that small batches still fill 256 CUs; kernel 2 combines the partials and folds gamma/beta.
//
// Adaptive form (reference quirk kept on purpose: AdaptiveGroupNorm applies its static affine twice, see
// oracle/restate.py:ada_group_norm):   y = (xhat*gamma + beta) * ag + ab,   ag = gamma + d_gamma, ab = beta + d_beta
#include "common.h"

namespace {

constexpr int GN_MAX_SPLIT = 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, long L, int split,
                                                         double* __restrict__ partial) {
  const long run = blockIdx.x;
  const int sp = blockIdx.y;
  const float* base = x + run * L;
  // slice boundaries, multiples of 4 elements
  long per = ((L + split - 1) / split + 3) & ~3L;
  long lo = (long)sp * per;
  long hi = lo + per < L ? lo + per : L;
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
  if (((L & 3) == 0) && ((((uintptr_t)base) & 15) == 0)) {
    const float4* b4 = reinterpret_cast<const float4*>(base);
    const long lo4 = lo >> 2, hi4 = hi >> 2;   // hi is a multiple of 4 here because L is
    for (long i = lo4 + threadIdx.x; i < hi4; i += 256) {
      const float4 v = b4[i];
      s0 += (double)v.x + (double)v.y;
      s1 += (double)v.z + (double)v.w;
      q0 += (double)v.x * v.x + (double)v.y * v.y;
      q1 += (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const double v = base[i];
      s0 += v;
      q0 += v * v;
    }
  }
  double s = wave_sum(s0 + s1), q = wave_sum(q0 + q1);
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* p = partial + (run * GN_MAX_SPLIT + sp) * 2;
    p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    p[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ partial, int split, int N, int C,
                                                          int G, long L, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ ada_gamma,
                                                          const float* __restrict__ ada_beta, long ada_stride,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  const int cpg = C / G;
  const int g = c / cpg;
  const double* p = partial + ((long)(n * G + g) * GN_MAX_SPLIT) * 2;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < split; ++k) { s += p[2 * k]; q += p[2 * k + 1]; }
  const double mean = s / (double)L;
  double var = q / (double)L - mean * mean;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double gm = gamma ? (double)gamma[c] : 1.0;
  const double bt = beta ? (double)beta[c] : 0.0;
  double sc = rstd * gm;
  double sh = bt - mean * sc;
  if (ada_gamma) {
    const double ag = ada_gamma[(long)n * ada_stride + c];
    const double ab = ada_beta[(long)n * ada_stride + c];
    sc = sc * ag;
    sh = sh * ag + ab;
  }
  scale[i] = (float)sc;
  shift[i] = (float)sh;
  if (mean_out && c == g * cpg) {
    mean_out[n * G + g] = (float)mean;
    rstd_out[n * G + g] = (float)rstd;
  }
}

// GroupNorm affine from the per-tile statistics the convolution epilogue leaves behind (conv_igemm.h, gn_stats):
// stats [N][T][C][2] = (mean, M2) of `cnt` values each.  One block per (sample, group); the cpg*T entries of a group are
// combined in fp64 with the equal-count form of Chan's update
//     mean = avg(mean_i),   M2 = sum(M2_i) + cnt * (sum(mean_i^2) - K * mean^2)
// (the fp64 subtraction loses nothing that matters: mean_i are fp32 values).  Then gamma / beta / adaptive weights are
// folded exactly as in gn_finalize_kernel.
__global__ __launch_bounds__(256) void gn_from_tiles_kernel(const float2* __restrict__ stats, int T, int C, int G, int cnt,
                                                            float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ ada_gamma,
                                                            const float* __restrict__ ada_beta, long ada_stride,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int n = blockIdx.x / G, g = blockIdx.x - n * G;
  const int cpg = C / G;
  const float2* base = stats + (long)n * T * C + (long)g * cpg;
  double s1 = 0.0, s2 = 0.0, sm = 0.0;
  const long K = (long)T * cpg;
  for (long i = threadIdx.x; i < K; i += 256) {
    const long t = i / cpg;
    const int c = (int)(i - t * cpg);
    const float2 e = base[t * C + c];
    s1 += (double)e.x;
    s2 += (double)e.x * (double)e.x;
    sm += (double)e.y;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); sm = wave_sum(sm);
  __shared__ double red[3][4];
  __shared__ double fin[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; red[2][wave] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double a2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double am = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const double mean = a1 / (double)K;
    double m2 = am + (double)cnt * (a2 - (double)K * mean * mean);
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / ((double)K * (double)cnt);
    fin[0] = mean;
    fin[1] = 1.0 / sqrt(var + (double)eps);
    if (mean_out) { mean_out[n * G + g] = (float)mean; rstd_out[n * G + g] = (float)fin[1]; }
  }
  __syncthreads();
  const double mean = fin[0], rstd = fin[1];
  for (int k = threadIdx.x; k < cpg; k += 256) {
    const int c = g * cpg + k;
    const double gm = gamma ? (double)gamma[c] : 1.0;
    const double bt = beta ? (double)beta[c] : 0.0;
    double sc = rstd * gm;
    double sh = bt - mean * sc;
    if (ada_gamma) {
      const double ag = ada_gamma[(long)n * ada_stride + c];
      const double ab = ada_beta[(long)n * ada_stride + c];
      sc = sc * ag;
      sh = sh * ag + ab;
    }
    scale[(long)n * C + c] = (float)sc;
    shift[(long)n * C + c] = (float)sh;
  }
}

}  // namespace

extern "C" int64_t emo_groupnorm_workspace_bytes(int N, int G) {
  if (N <= 0 || G <= 0) return EMO_ERR_BAD_ARG;
  return (int64_t)N * G * GN_MAX_SPLIT * 2 * (int64_t)sizeof(double);
}

extern "C" int emo_groupnorm_affine_f32(const float* x, int N, int C, int64_t S, int G, float eps, const float* gamma,
                                        const float* beta, const float* ada_gamma, const float* ada_beta,
                                        int64_t ada_stride, float* scale, float* shift, float* mean_out,
                                        float* rstd_out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !scale || !shift || !workspace) return EMO_ERR_BAD_ARG;
  if (N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) return EMO_ERR_BAD_ARG;
  if ((ada_gamma == nullptr) != (ada_beta == nullptr)) return EMO_ERR_BAD_ARG;
  if ((mean_out == nullptr) != (rstd_out == nullptr)) return EMO_ERR_BAD_ARG;
  if (workspace_bytes < emo_groupnorm_workspace_bytes(N, G)) return EMO_ERR_BAD_ARG;
  if ((long)N * G > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  const long L = (long)(C / G) * S;
  int split = (int)((L + 16383) / 16384);
  if (split < 1) split = 1;
  if (split > GN_MAX_SPLIT) split = GN_MAX_SPLIT;
  hipStream_t s = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(N * G, split), dim3(256), 0, s, x, L, split, part);
  int rc = emo_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(emo_cdiv((long)N * C, 256)), dim3(256), 0, s, part, split, N, C, G, L,
                     eps, gamma, beta, ada_gamma, ada_beta, (long)ada_stride, scale, shift, mean_out, rstd_out);
  return emo_launch_status();
}

// The second half of emo_groupnorm_affine_f32 alone: `partial` holds `split` (sum, sum of squares) slices per (sample, group)
// in the workspace layout ([N * G][64][2] doubles), left there by a producer that had the tensor in its registers anyway
// (emo_upsample_trilinear_gn_sums_f32, resample.hip).  S = elements per (sample, channel) of the tensor the sums are of.
extern "C" int emo_groupnorm_affine_from_sums_f32(const void* partial, int split, int N, int C, int64_t S, int G, float eps,
                                                  const float* gamma, const float* beta, const float* ada_gamma,
                                                  const float* ada_beta, int64_t ada_stride, float* scale, float* shift,
                                                  float* mean_out, float* rstd_out, void* stream) {
  if (!partial || !scale || !shift) return EMO_ERR_BAD_ARG;
  if (N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G || split < 1 || split > GN_MAX_SPLIT) return EMO_ERR_BAD_ARG;
  if ((ada_gamma == nullptr) != (ada_beta == nullptr)) return EMO_ERR_BAD_ARG;
  if ((mean_out == nullptr) != (rstd_out == nullptr)) return EMO_ERR_BAD_ARG;
  if ((long)N * G > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  const long L = (long)(C / G) * S;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(emo_cdiv((long)N * C, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const double*>(partial), split, N, C, G, L, eps, gamma, beta, ada_gamma, ada_beta,
                     (long)ada_stride, scale, shift, mean_out, rstd_out);
  return emo_launch_status();
}

extern "C" int emo_groupnorm_affine_from_tiles_f32(const float* stats, int N, int C, int64_t T, int cnt, int G, float eps,
                                                   const float* gamma, const float* beta, const float* ada_gamma,
                                                   const float* ada_beta, int64_t ada_stride, float* scale, float* shift,
                                                   float* mean_out, float* rstd_out, void* stream) {
  if (!stats || !scale || !shift) return EMO_ERR_BAD_ARG;
  if (N <= 0 || C <= 0 || T <= 0 || cnt <= 0 || G <= 0 || C % G) return EMO_ERR_BAD_ARG;
  if ((ada_gamma == nullptr) != (ada_beta == nullptr)) return EMO_ERR_BAD_ARG;
  if ((mean_out == nullptr) != (rstd_out == nullptr)) return EMO_ERR_BAD_ARG;
  if ((long)N * G > 0x7fffffffL || T > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  if ((((uintptr_t)stats) & 7u) != 0) return EMO_ERR_ALIGN;
  hipLaunchKernelGGL(gn_from_tiles_kernel, dim3(N * G), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(stats), (int)T, C, G, cnt, eps, gamma, beta, ada_gamma, ada_beta,
                     (long)ada_stride, scale, shift, mean_out, rstd_out);
  return emo_launch_status();
}
