// ABI identity of libemoportraits_hip.so (see include/emo_hip.h).
#include "common.h"

extern "C" int emo_abi_version(void) { return EMO_ABI_VERSION; }

extern "C" const char* emo_build_info(void) {
  return "libemoportraits_hip gfx950 (CDNA4, wave64) fp32; built " __DATE__ " " __TIME__
         " with hipcc " __clang_version__;
}

// ABI 9.  What the planners size their launches by (pack.py read 256 from a constant before: on a partitioned device -- CPX, 32 .. 128
// CUs -- Python and the C launchers disagreed on when the pointwise / two-tile kernels pay off)
extern "C" int emo_device_cu_count(void) { return emo_cu_count(); }

// ABI 9.  Diagnostic, not on the hot path: a bare stream of v_mfma_f32_32x32x16_f16 -- one wave per SIMD, eight independent
// accumulator tiles, pseudo-random fp16 operands (operand toggling is part of the power the matrix pipes draw) -- optionally with
// the fragment reads of the split convolution's K loop (8 ds_read_b128 per 12 MFMAs).  bench.py times it for about a second behind
// its timed region: the rate it sustains is what the POWER-MANAGED chip gives a kernel that does nothing but MFMAs
// (`roofline.sustained_peak`), beside the 2.5 PF of the data sheet.
typedef _Float16 emo_halfx8 __attribute__((ext_vector_type(8)));
typedef float emo_floatx16 __attribute__((ext_vector_type(16)));

template <int LDS_READS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void emo_mfma_stream_kernel(float* __restrict__ sink, int iters) {
  __shared__ __attribute__((aligned(16))) emo_halfx8 lds8[4096];      // 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) {
    emo_halfx8 v;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (_Float16)((float)((unsigned)(i * 8 + u) * 2654435761u >> 20 & 1023) * (1.0f / 256.0f) - 2.0f);
    lds8[i] = v;
  }
  __syncthreads();
  emo_floatx16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  emo_halfx8 fa[2][4], fb[2][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { fa[0][k] = lds8[k * 64 + lane]; fb[0][k] = lds8[(4 + k) * 64 + lane]; fa[1][k] = fa[0][k]; fb[1][k] = fb[0][k]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int gs = 0; gs < 8; ++gs) {
      const int cur = LDS_READS ? (gs & 1) : 0, nxt = cur ^ 1;
      __builtin_amdgcn_sched_barrier(0);
      if (LDS_READS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          fa[nxt][k] = lds8[((gs * 8 + k) * 64 + lane + (it & 7) * 512) & 4095];
          fb[nxt][k] = lds8[((gs * 8 + 4 + k) * 64 + lane + (it & 7) * 512) & 4095];
        }
      }
      // 12 MFMAs per step: 4 accumulator tiles x 3 products, as a step of the fp16-split kernels
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[(p == 2 ? 0 : 4) + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[cur][(p + t) & 3], fa[cur][(p * 2 + t) & 3], acc[(p == 2 ? 0 : 4) + t], 0, 0, 0);
      if (LDS_READS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;      // (keeps the accumulators alive; practically never true)
}

extern "C" int emo_mfma_stream_f16(float* sink, int iters, int lds_reads, int64_t* mfma_per_launch, void* stream) {
  if (!sink || iters <= 0) return EMO_ERR_BAD_ARG;
  const int ncu = emo_cu_count();
  if (mfma_per_launch) *mfma_per_launch = (int64_t)ncu * 4 * 96 * iters;     // blocks x waves x MFMAs per iteration
  if (lds_reads) hipLaunchKernelGGL(emo_mfma_stream_kernel<1>, dim3(ncu), dim3(256), 0, (hipStream_t)stream, sink, iters);
  else hipLaunchKernelGGL(emo_mfma_stream_kernel<0>, dim3(ncu), dim3(256), 0, (hipStream_t)stream, sink, iters);
  return emo_launch_status();
}
