// Entry of the LDS-staged 3-D grid_sample behind emo_grid_sample3d_f32 (grid_sample3d.hip checks the arguments), and the
// packed-4 layout repack -- SURVEY.md section 8 rows a1 + a2.
#include "common.h"

#define EMO_GS3D_TILE_PAD_SIGNATURE(name)                                                                              \
  int name(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,            \
           const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo, long vol_bstride, \
           int in_layout, int out_layout, int variant, int grid_kind, hipStream_t s)
EMO_GS3D_TILE_PAD_SIGNATURE(emo_gs3d_tile_zeros);
EMO_GS3D_TILE_PAD_SIGNATURE(emo_gs3d_tile_border);
EMO_GS3D_TILE_PAD_SIGNATURE(emo_gs3d_tile_reflection);

int emo_gs3d_tile_dispatch(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                           const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                           int64_t vol_batch_stride, int padding_mode, int in_layout, int out_layout, int variant,
                           int grid_kind, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define EMO_GS3D_ARGS vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, (long)vol_batch_stride, in_layout, out_layout, variant, grid_kind, s
  switch (padding_mode) {
    case EMO_PAD_ZEROS: return emo_gs3d_tile_zeros(EMO_GS3D_ARGS);
    case EMO_PAD_BORDER: return emo_gs3d_tile_border(EMO_GS3D_ARGS);
    case EMO_PAD_REFLECTION: return emo_gs3d_tile_reflection(EMO_GS3D_ARGS);
    default: return EMO_ERR_BAD_ARG;
  }
#undef EMO_GS3D_ARGS
}

namespace {
// NCDHW [n][C][S] <-> packed-4 [n][C/4][S][4]: one thread per (quad, position); reads / writes of 4 planes coalesced along S
__global__ __launch_bounds__(256) void repack_p4_kernel(const float* __restrict__ in, float* __restrict__ out, int Q, int S,
                                                        int to_p4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // over Q * S
  if (i >= (long)Q * S) return;
  const int n = blockIdx.y;
  const int q = (int)(i / S);
  const int sp = (int)(i - (long)q * S);
  const long base = (long)n * Q * S * 4;
  if (to_p4) {
    const float* ip = in + base + (long)q * 4 * S + sp;
    float4 v = make_float4(ip[0], ip[S], ip[2L * S], ip[3L * S]);
    *reinterpret_cast<float4*>(out + base + i * 4) = v;
  } else {
    const float4 v = *reinterpret_cast<const float4*>(in + base + i * 4);
    float* op = out + base + (long)q * 4 * S + sp;
    op[0] = v.x; op[S] = v.y; op[2L * S] = v.z; op[3L * S] = v.w;
  }
}
}  // namespace

int emo_repack_p4_dispatch(const float* in, float* out, int N, int C, int DHW, int to_p4, void* stream) {
  if (C % 4) return EMO_ERR_UNSUPPORTED;
  const long items = (long)(C / 4) * DHW;
  if ((items + 255) / 256 > 0x7fffffffL || N > 65535) return EMO_ERR_UNSUPPORTED;
  dim3 g((unsigned)((items + 255) / 256), N);
  hipLaunchKernelGGL(repack_p4_kernel, g, dim3(256), 0, (hipStream_t)stream, in, out, C / 4, DHW, to_p4);
  return emo_launch_status();
}
