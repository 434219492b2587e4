// Small wave-reduced kernels of the hot path for gfx950: the embedding arithmetic in front of the WarpGenerator
// (part of a5) and the output packing (a11).  None of these is MFMA-shaped:
// M is a few thousand rows, the per-frame "N" is 1..16, so each output row is one 64-lane dot product.
#include "common.h"

namespace {

// C[b][m][0..NN) = sum_k A[m][k] * B[b][k][0..NN)   (A shared by the batch: Linear / 1x1 conv on 4x4 / projector u)
// one wave per (m, b); lanes stride over k; NN <= 16 accumulators per lane; butterfly reduction.
template <int NN>
__global__ __launch_bounds__(256) void small_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int K, long b_stride,
                                                         long c_stride) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  const int b = blockIdx.y;
  if (m >= M) return;
  const float* a = A + (long)m * K;
  const float* bb = B + (long)b * b_stride;
  float acc[NN];
#pragma unroll
  for (int j = 0; j < NN; ++j) acc[j] = 0.0f;
  for (int k = lane; k < K; k += 64) {
    const float av = a[k];
    const float* br = bb + (long)k * NN;
#pragma unroll
    for (int j = 0; j < NN; ++j) acc[j] = __fmaf_rn(av, br[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < NN; ++j) {
    float v = acc[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    acc[j] = v;
  }
  if (lane == 0) {
    float* c = C + (long)b * c_stride + (long)m * NN;
#pragma unroll
    for (int j = 0; j < NN; ++j) c[j] = acc[j];
  }
}

// ProjectorNorm second half + assign_adaptive_norm_params (networks/volumetric_avatar/utils.py:1137-1151, :983-995):
//   T[b][c][0..E) (= u_i @ embed) times v_i [E][2] -> (d_gamma, d_beta); ada_gamma = gamma_c + d_gamma, ada_beta = beta_c + d_beta
// rows c of ALL adaptive norms of a net are concatenated; norm_of_row[c] selects v_i.
__global__ __launch_bounds__(256) void projector_finalize_kernel(const float* __restrict__ T, const float* __restrict__ V,
                                                                 const int* __restrict__ norm_of_row,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* __restrict__ ag,
                                                                 float* __restrict__ ab, int B, int R, int E) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * R) return;
  const int c = i % R;
  const float* t = T + (long)i * E;
  const float* v = V + (long)norm_of_row[c] * E * 2;
  float dg = 0.0f, db = 0.0f;
  for (int k = 0; k < E; ++k) {
    dg = __fmaf_rn(t[k], v[2 * k], dg);
    db = __fmaf_rn(t[k], v[2 * k + 1], db);
  }
  ag[i] = gamma[c] + dg;
  ab[i] = beta[c] + db;
}

// utils/point_transforms.py:188-242 get_transform_matrix: theta = S @ R @ T  (one thread per sample)
__global__ void pose_theta_kernel(const float* __restrict__ scale, int scale_cols, const float* __restrict__ rotation,
                                  const float* __restrict__ translation, float* __restrict__ theta, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float sx = scale[b * scale_cols], sy = scale_cols == 3 ? scale[b * 3 + 1] : sx,
              sz = scale_cols == 3 ? scale[b * 3 + 2] : sx;
  const float kPi = 3.14159265358979323846f;
  auto clampf = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
  const float yaw = clampf(rotation[b * 3 + 0], -kPi / 2, kPi), pitch = clampf(rotation[b * 3 + 1], -kPi / 2, kPi),
              roll = clampf(rotation[b * 3 + 2], -kPi / 2, kPi);
  const float yc = cosf(yaw), ys = sinf(yaw), pc = cosf(pitch), ps = sinf(pitch), rc = cosf(roll), rs = sinf(roll);
  float R[3][3];
  R[0][0] = yc * pc;  R[0][1] = yc * ps * rs - ys * rc;  R[0][2] = yc * ps * rc + ys * rs;
  R[1][0] = ys * pc;  R[1][1] = ys * ps * rs + yc * rc;  R[1][2] = ys * ps * rc - yc * rs;
  R[2][0] = -ps;      R[2][1] = pc * rs;                 R[2][2] = pc * rc;
  const float S[3] = {sx, sy, sz};
  const float t[3] = {translation[b * 3], translation[b * 3 + 1], translation[b * 3 + 2]};
  float* o = theta + (long)b * 16;
  for (int i = 0; i < 3; ++i) {
    float row[3];
    for (int j = 0; j < 3; ++j) { row[j] = S[i] * R[i][j]; o[i * 4 + j] = row[j]; }
    // (S R T)[i][3] = sum_j (S R)[i][j] * t[j]   (the 4th column of S R is zero, T's diagonal is one)
    o[i * 4 + 3] = row[0] * t[0] + row[1] * t[1] + row[2] * t[2];
  }
  o[12] = 0.0f; o[13] = 0.0f; o[14] = 0.0f; o[15] = 1.0f;
}

// notebooks/infer.py:641-644: img.clamp(0,1) -> ToPILImage (mul(255).byte(), i.e. truncation) -> HWC uint8
__global__ __launch_bounds__(256) void pack_rgb8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                        long N, long HW) {
  const long total = N * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    const float* src = img + n * 3 * HW + p;
    uint8_t* dst = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = src[c * HW];
      v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
      dst[c] = (uint8_t)(v * 255.0f);
    }
  }
}

// the inverse direction, notebooks/infer.py:211-223 convert_to_tensor (ToTensor): [N,H,W,3] uint8 -> [N,3,H,W] fp32 = byte / 255
// (an fp32 division like torch's, not a multiplication by the rounded reciprocal)
__global__ __launch_bounds__(256) void unpack_rgb8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, long N,
                                                          long HW) {
  const long total = N * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / HW, p = i - n * HW;
    const uint8_t* src = in + i * 3;
    float* dst = out + n * 3 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c * HW] = __fdiv_rn((float)src[c], 255.0f);
  }
}

// inverse of B 4x4 matrices (head-pose affines: notebooks/infer.py:443, expression_embedder.py:185-188 call
// `theta.float().inverse()`): Gauss-Jordan with partial pivoting in double, one thread per matrix.
__global__ __launch_bounds__(64) void mat4_inverse_kernel(const float* __restrict__ in, float* __restrict__ out, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double m[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[i][j] = (double)in[b * 16 + i * 4 + j];
      m[i][4 + j] = i == j ? 1.0 : 0.0;
    }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    double best = fabs(m[c][c]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r > c && fabs(m[r][c]) > best) {
        best = fabs(m[r][c]);
        piv = r;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r == piv && piv != c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double t = m[c][j];
          m[c][j] = m[r][j];
          m[r][j] = t;
        }
      }
    const double inv = 1.0 / m[c][c];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[c][j] *= inv;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        const double f = m[r][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[b * 16 + i * 4 + j] = (float)m[i][4 + j];
}

}  // namespace

extern "C" int emo_small_gemm_f32(const float* A, const float* B, float* C, int M, int K, int NN, int batch,
                                  int64_t b_stride, int64_t c_stride, void* stream) {
  if (!A || !B || !C || M <= 0 || K <= 0 || batch <= 0) return EMO_ERR_BAD_ARG;
  if (batch > 65535) return EMO_ERR_UNSUPPORTED;
  dim3 g(emo_cdiv(M, 4), batch);
  hipStream_t s = (hipStream_t)stream;
  switch (NN) {
    case 1: hipLaunchKernelGGL(small_gemm_kernel<1>, g, dim3(256), 0, s, A, B, C, M, K, (long)b_stride, (long)c_stride); break;
    case 2: hipLaunchKernelGGL(small_gemm_kernel<2>, g, dim3(256), 0, s, A, B, C, M, K, (long)b_stride, (long)c_stride); break;
    case 4: hipLaunchKernelGGL(small_gemm_kernel<4>, g, dim3(256), 0, s, A, B, C, M, K, (long)b_stride, (long)c_stride); break;
    case 16: hipLaunchKernelGGL(small_gemm_kernel<16>, g, dim3(256), 0, s, A, B, C, M, K, (long)b_stride, (long)c_stride); break;
    default: return EMO_ERR_UNSUPPORTED;
  }
  return emo_launch_status();
}

extern "C" int emo_projector_finalize_f32(const float* T, const float* V, const int* norm_of_row, const float* gamma,
                                          const float* beta, float* ada_gamma, float* ada_beta, int B, int R, int E,
                                          void* stream) {
  if (!T || !V || !norm_of_row || !gamma || !beta || !ada_gamma || !ada_beta || B <= 0 || R <= 0 || E <= 0)
    return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(projector_finalize_kernel, dim3(emo_cdiv((long)B * R, 256)), dim3(256), 0, (hipStream_t)stream, T,
                     V, norm_of_row, gamma, beta, ada_gamma, ada_beta, B, R, E);
  return emo_launch_status();
}

extern "C" int emo_pose_theta_f32(const float* scale, int scale_cols, const float* rotation, const float* translation,
                                  float* theta, int B, void* stream) {
  if (!scale || !rotation || !translation || !theta || B <= 0) return EMO_ERR_BAD_ARG;
  if (scale_cols != 1 && scale_cols != 3) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(pose_theta_kernel, dim3(emo_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, scale, scale_cols,
                     rotation, translation, theta, B);
  return emo_launch_status();
}

extern "C" int emo_pack_rgb8(const float* img, uint8_t* out, int N, int H, int W, void* stream) {
  if (!img || !out || N <= 0 || H <= 0 || W <= 0) return EMO_ERR_BAD_ARG;
  const long total = (long)N * H * W;
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_rgb8_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, img, out, (long)N,
                     (long)H * W);
  return emo_launch_status();
}

extern "C" int emo_unpack_rgb8(const uint8_t* in, float* out, int N, int H, int W, void* stream) {
  if (!in || !out || N <= 0 || H <= 0 || W <= 0) return EMO_ERR_BAD_ARG;
  const long total = (long)N * H * W;
  long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(unpack_rgb8_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, in, out, (long)N,
                     (long)H * W);
  return emo_launch_status();
}

extern "C" int emo_mat4_inverse_f32(const float* in, float* out, int B, void* stream) {
  if (!in || !out || B <= 0) return EMO_ERR_BAD_ARG;
  hipLaunchKernelGGL(mat4_inverse_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, B);
  return emo_launch_status();
}
