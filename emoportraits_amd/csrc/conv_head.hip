// Pointwise convolution with a HANDFUL of output channels for gfx950 -- the decoder's image head: GroupNorm -> ReLU -> 1x1 conv
// 128 -> 3 -> sigmoid at the full output resolution (networks/volumetric_avatar/decoder.py:381-392, utils.py:661-788 toolkit).
//
// With 3 output channels the layer is a stream, not a GEMM: 2 x Cin x 3 FLOPs per position against 4 x Cin bytes read -- 1.5
// FLOP per byte, 2.1 GB per 16 frames at 512 x 512.  On the fp32 MFMA kernel a 32-row tile is 29/32 idle and the layer ran at
// 3.3 TB/s of its bytes behind LDS staging it does not need (0.65 ms: 1.5 % of the driver pass, a third of what was left on that
// kernel).  Here every thread owns four consecutive positions and walks the input channels: one 16-byte load per channel (a
// wave reads 1 KB contiguous of one channel plane), the per-(sample, channel) affine of the folded GroupNorm + ReLU applied in
// registers, COUT fused multiply-adds per value against weights the compiler keeps in scalar registers (uniform addresses:
// s_load), bias + activation, one 16-byte store per output channel.  No LDS, no barrier; 8 loads in flight per thread.
//
// Arithmetic per output: acc = fma(w[o][c], relu(fma(x, scale, shift)), acc) for c = 0 .. Cin-1 in order, + bias, activation --
// the operations of the implicit-GEMM kernel's staging and epilogue with a sequential fp32 sum over the channels in place of
// the MFMA's blocked one (held to the same bounds against the oracle: tests/test_kernels_gpu.py, test_bench_config_parity_gpu.py).
#include "common.h"

namespace {

__device__ __forceinline__ float head_act(float v, int act) {
  if (act == EMO_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == EMO_ACT_TANH) return tanhf(v);
  if (act == EMO_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ out, int Cin,
                                                        long S, int relu_in, int act) {
  const int n = blockIdx.y;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;          // quad of positions
  if (4 * q >= S) return;
  const float4* xp = reinterpret_cast<const float4*>(x + (long)n * Cin * S) + q;
  const long S4 = S >> 2;
  const float* sc = scale ? scale + (long)n * Cin : nullptr;
  const float* sh = scale ? shift + (long)n * Cin : nullptr;
  const float floor_ = relu_in ? 0.0f : -__builtin_huge_valf();
  float acc[COUT][4];
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[o][j] = 0.0f;
  constexpr int U = 8;
  int c = 0;
  for (; c + U <= Cin; c += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = xp[(long)(c + u) * S4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      if (sc) {
        const float a = sc[c + u], b = sh[c + u];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __fmaf_rn(e[j], a, b);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = fmaxf(e[j], floor_);
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float wv = w[o * Cin + c + u];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = __fmaf_rn(wv, e[j], acc[o][j]);
      }
    }
  }
  for (; c < Cin; ++c) {
    const float4 v = xp[(long)c * S4];
    float e[4] = {v.x, v.y, v.z, v.w};
    if (sc) {
      const float a = sc[c], b = sh[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = __fmaf_rn(e[j], a, b);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = fmaxf(e[j], floor_);
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const float wv = w[o * Cin + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[o][j] = __fmaf_rn(wv, e[j], acc[o][j]);
    }
  }
  float4* op = reinterpret_cast<float4*>(out + (long)n * COUT * S) + q;
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float b = bias ? bias[o] : 0.0f;
    op[(long)o * S4] = make_float4(head_act(acc[o][0] + b, act), head_act(acc[o][1] + b, act), head_act(acc[o][2] + b, act),
                                   head_act(acc[o][3] + b, act));
  }
}

}  // namespace

// out[n][o][p] = act(bias[o] + sum_c w[o][c] * in(x[n][c][p])),  in(v) = relu_in ? max(v * scale[n][c] + shift[n][c], 0) : affine
// x [N, Cin, S] (S positions per channel, any number of spatial dimensions), w [Cout, Cin] plain row-major fp32, Cout in 1..4,
// S a multiple of 4, x and out 16-byte aligned (EMO_ERR_UNSUPPORTED / EMO_ERR_ALIGN otherwise: the caller runs the
// implicit-GEMM kernel).  scale / shift [N, Cin] both or neither; bias [Cout] or NULL.
extern "C" int emo_conv_head_f32(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                                 float* out, int N, int Cin, int Cout, int64_t S, int relu_in, int act, void* stream) {
  if (!x || !w || !out || N <= 0 || Cin <= 0 || Cout <= 0 || S <= 0) return EMO_ERR_BAD_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMO_ERR_BAD_ARG;
  if (act < EMO_ACT_NONE || act > EMO_ACT_SIGMOID) return EMO_ERR_BAD_ARG;
  if (Cout > 4 || (S & 3) || N > 65535) return EMO_ERR_UNSUPPORTED;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) != 0) return EMO_ERR_ALIGN;
  const long blocks = (S / 4 + 255) / 256;
  if (blocks > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks, (unsigned)N);
  hipStream_t s = (hipStream_t)stream;
  switch (Cout) {
    case 1: hipLaunchKernelGGL(conv_head_kernel<1>, grid, dim3(256), 0, s, x, w, bias, scale, shift, out, Cin, (long)S, relu_in, act); break;
    case 2: hipLaunchKernelGGL(conv_head_kernel<2>, grid, dim3(256), 0, s, x, w, bias, scale, shift, out, Cin, (long)S, relu_in, act); break;
    case 3: hipLaunchKernelGGL(conv_head_kernel<3>, grid, dim3(256), 0, s, x, w, bias, scale, shift, out, Cin, (long)S, relu_in, act); break;
    default: hipLaunchKernelGGL(conv_head_kernel<4>, grid, dim3(256), 0, s, x, w, bias, scale, shift, out, Cin, (long)S, relu_in, act); break;
  }
  return emo_launch_status();
}
