// Generic 2-D convolution for gfx950: any kernel size / stride / zero padding, fp32 on the MFMA pipe.
//
// Used by the embedders: the torchvision-style ResNets behind IdtEmbed / ExpressionEmbed /
// HeadPoseRegressor (networks/volumetric_avatar/identity_embedder.py:59-69, expression_embedder.py:424-439,
// head_pose_regressor.py:21-32) are strided (7x7 s2, 3x3 s2, 1x1 s2) and end at 8x8 / 4x4 feature maps -- shapes the
// tiled implicit-GEMM kernel of the hot path (conv_igemm.h) does not cover.  Work is ~3 GMAC per frame against ~510 GMAC
// in the decoder, so this kernel is built for generality, not for the last 20 % of the MFMA roofline:
//
//   GEMM view   D[co][col] = sum_k  Wt[k][co] * X[k][col],   k = (ci, ky, kx),  col = (n, yo, xo)  (batch folded
//               into the columns, so a 4x4 map with 16 frames still fills 64-wide tiles)
//   block       256 threads = 4 waves (2x2), tile 64 co x 64 col, one 32x32 accumulator per wave
//   split-K     the deep layers are tiny GEMMs with a long K (4x4 map, K = 4608: 8..32 tiles, 144 serial chunks), so K is
//               split over gridDim.z into a workspace and summed in fixed order by a second kernel (deterministic; no
//               atomics); emo_conv2d_generic_splits() is the launch heuristic the host sizes the workspace with
//   K loop      chunks of 32: im2col gather of X (with the producer's norm-apply + ReLU folded in, as in the hot-path
//               kernel) and a coalesced copy of Wt into LDS, next chunk's global loads in flight during the MFMAs
//   MFMA        v_mfma_f32_32x32x2_f32: A lane (l&31, k=l>>5), B lane (k=l>>5, l&31); D col = l&31,
//               row = (r&3) + 8*(r>>2) + 4*(l>>5)
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 64, GBN = 64, GKC = 32, GLD = 96;   // LDS row stride 96 floats: rows k, k+1 land on disjoint banks

struct GenericConvArgs {
  const float* x;
  const float* wt;      // [K][CoutP], CoutP multiple of 64, zero padded
  const float* bias;    // [Cout] or null
  const float* scale;   // [N*Cin] or null
  const float* shift;
  float* out;
  int N, Cin, H, W, Cout, CoutP, Ho, Wo, KH, KW, stride, pad, relu_in, K;
  int chunks_per_split;   // K chunks handled by one blockIdx.z
  float* partial;         // [splits][N*Cout*Ho*Wo] when splits > 1 (bias is added by the reduction), else null
};

template <bool AFFINE>
__global__ __launch_bounds__(256) void conv2d_generic_kernel(const GenericConvArgs a) {
  __shared__ float As[GKC][GLD];
  __shared__ float Bs[GKC][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int co0 = blockIdx.y * GBM;
  const long ncols = (long)a.N * a.Ho * a.Wo;
  // this thread's im2col column (fixed for the whole K loop)
  const int col = tid & 63, krow0 = tid >> 6;
  const long gcol = (long)blockIdx.x * GBN + col;
  const bool col_ok = gcol < ncols;
  const int hw = a.Ho * a.Wo;
  const int n = col_ok ? (int)(gcol / hw) : 0;
  const int p = col_ok ? (int)(gcol - (long)n * hw) : 0;
  const int yo = p / a.Wo, xo = p - yo * a.Wo;
  const int yb = yo * a.stride - a.pad, xb = xo * a.stride - a.pad;
  const float* xn = a.x + (long)n * a.Cin * a.H * a.W;
  const int khw = a.KH * a.KW;

  float ra[8], rb[8];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + krow0 + 4 * i;
      const bool kok = k < a.K;
      ra[i] = kok ? a.wt[(long)k * a.CoutP + co0 + col] : 0.0f;
      const int kc = kok ? k : 0;
      const int ci = kc / khw;
      const int r = kc - ci * khw;
      const int ky = r / a.KW, kx = r - ky * a.KW;
      const int yi = yb + ky, xi = xb + kx;
      const bool ok = kok && col_ok && yi >= 0 && yi < a.H && xi >= 0 && xi < a.W;
      float v = 0.0f;
      if (ok) {
        v = xn[((long)ci * a.H + yi) * a.W + xi];
        if (AFFINE) {
          v = __fmaf_rn(v, a.scale[n * a.Cin + ci], a.shift[n * a.Cin + ci]);
          if (a.relu_in) v = fmaxf(v, 0.0f);
        }
      }
      rb[i] = v;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

  const int k_begin = blockIdx.z * a.chunks_per_split * GKC;
  const int k_end = min(a.K, k_begin + a.chunks_per_split * GKC);
  issue(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += GKC) {
    __syncthreads();   // previous chunk's MFMAs are done with the LDS tiles
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      As[krow0 + 4 * i][col] = ra[i];
      Bs[krow0 + 4 * i][col] = rb[i];
    }
    __syncthreads();
    if (k0 + GKC < k_end) issue(k0 + GKC);
    const int kl = lane >> 5, j = lane & 31;
#pragma unroll
    for (int kk = 0; kk < GKC; kk += 2) {
      const float av = As[kk + kl][wm * 32 + j];
      const float bv = Bs[kk + kl][wn * 32 + j];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }

  const long oc = (long)blockIdx.x * GBN + wn * 32 + (lane & 31);
  if (oc < ncols) {
    const int on = (int)(oc / hw);
    const int op = (int)(oc - (long)on * hw);
    float* base = a.partial ? a.partial + (long)blockIdx.z * ncols * a.Cout : a.out;
    float* o = base + (long)on * a.Cout * hw + op;
    const bool add_bias = a.bias && !a.partial;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (co < a.Cout) o[(long)co * hw] = acc[r] + (add_bias ? a.bias[co] : 0.0f);
    }
  }
}

// out[i] = sum_z partial[z][i] + bias[channel(i)], z ascending
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                            float* __restrict__ out, long total, int splits, int Cout,
                                                            int hw) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float s = partial[i];
    for (int z = 1; z < splits; ++z) s += partial[(long)z * total + i];
    out[i] = s + (bias ? bias[(i / hw) % Cout] : 0.0f);
  }
}

// aim for >= 2 blocks per CU (256 CUs), keep >= 4 K-chunks per split
int splits_for(long ncols, int Cout, int K) {
  const long tiles = ((ncols + GBN - 1) / GBN) * ((Cout + GBM - 1) / GBM);
  const int chunks = (K + GKC - 1) / GKC;
  long want = (512 + tiles - 1) / tiles;
  long cap = chunks / 4;
  if (want > cap) want = cap;
  if (want > 32) want = 32;
  return want < 1 ? 1 : (int)want;
}

}  // namespace

extern "C" int emo_conv2d_generic_splits(int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
  if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return EMO_ERR_BAD_ARG;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return EMO_ERR_BAD_ARG;
  return splits_for((long)N * Ho * Wo, Cout, Cin * KH * KW);
}

extern "C" int emo_conv2d_generic_f32(const float* x, const float* wt, const float* bias, const float* scale,
                                      const float* shift, float* out, int N, int Cin, int H, int W, int Cout, int KH,
                                      int KW, int stride, int pad, int relu_in, int splits, float* workspace,
                                      void* stream) {
  if (!x || !wt || !out || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0)
    return EMO_ERR_BAD_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMO_ERR_BAD_ARG;
  if (splits < 1 || (splits > 1 && !workspace)) return EMO_ERR_BAD_ARG;
  GenericConvArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.scale = scale; a.shift = shift; a.out = out;
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.CoutP = (Cout + GBM - 1) / GBM * GBM;
  a.Ho = (H + 2 * pad - KH) / stride + 1;
  a.Wo = (W + 2 * pad - KW) / stride + 1;
  a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.relu_in = relu_in; a.K = Cin * KH * KW;
  if (a.Ho <= 0 || a.Wo <= 0) return EMO_ERR_BAD_ARG;
  const long ncols = (long)N * a.Ho * a.Wo;
  const int chunks = (a.K + GKC - 1) / GKC;
  if (splits > chunks) splits = chunks;
  a.chunks_per_split = (chunks + splits - 1) / splits;
  splits = (chunks + a.chunks_per_split - 1) / a.chunks_per_split;   // no empty split
  a.partial = splits > 1 ? workspace : nullptr;
  dim3 grid((unsigned)((ncols + GBN - 1) / GBN), (unsigned)(a.CoutP / GBM), (unsigned)splits);
  if (scale)
    hipLaunchKernelGGL(conv2d_generic_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv2d_generic_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  if (splits > 1) {
    const long total = ncols * Cout;
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, workspace, bias,
                       out, total, splits, Cout, a.Ho * a.Wo);
  }
  return emo_launch_status();
}
