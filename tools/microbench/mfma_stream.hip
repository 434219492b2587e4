// Microbenchmark behind csrc/conv_igemm_bf16x3.h: what a one-wave-per-SIMD stream of v_mfma_f32_32x32x16_bf16 reaches on
// MI355X when the pieces of that kernel's K loop are added one at a time.  One block of 256 threads per CU (160 KB of dynamic
// LDS), a "step" = 24 MFMAs (4 accumulator tiles x 6 products, two accumulator sets) as in the kernel; per 9 steps:
//   bit 0   12 ds_read_b128 per step, one step ahead, interleaved with the MFMAs (sched_group_barrier, as the kernel)
//   bit 1   a barrier (s_waitcnt lgkmcnt(0); s_barrier) every 3 steps
//   bit 2   5 LDS-DMA pieces (1 KiB per wave-instruction, L2-resident source) behind every barrier, waited with vmcnt(5)
//   bit 3   the patch conversion: 16 elements x (fma, med3, 3-way bf16 split) + 6 ds_write_b128 in two of the 9 steps
// Prints cycles per MFMA per SIMD (s_memtime, wave 0 of every block) and the rate in fp32-equivalent TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/microbench/mfma_stream.hip -o tools/microbench/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void stream(const char* __restrict__ wsrc, float* __restrict__ out, long long* __restrict__ cycles, int ncg, float sc, float sh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bf16x8* const lds8 = reinterpret_cast<bf16x8*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem);
  // fill 144 KB of LDS with small pseudo-random bf16 values (operand toggling is part of the power the matrix pipes draw)
  for (int i = tid; i < 9216; i += 256) {
    bf16x8 v;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (__bf16)(((i * 8 + u) * 2654435761u >> 20 & 255) * (1.0f / 256.0f) - 0.5f);
    lds8[i] = v;
  }
  __syncthreads();
  floatx16 acc[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][t][r] = 0.0f;
  bf16x8 fa[2][3][2], fb[2][3][2];   // [set][plane][tile]
#define LOAD(set_, step_)                                                                            \
  _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                    \
    fa[set_][pl][i] = lds8[((step_) * 12 + pl * 4 + i) * 64 + lane];                                 \
    fb[set_][pl][i] = lds8[((step_) * 12 + pl * 4 + 2 + i) * 64 + lane + wave * 16];                 \
  }
  LOAD(0, 0)
  floatx4 qv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) qv[u] = floatx4{sc * (lane + u), sh, sc, sh * u};
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
  constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int cg = 0; cg < ncg; ++cg) {
#pragma unroll
    for (int gs = 0; gs < 9; ++gs) {
      if (gs % 3 == 2) {
        if (MODE & 2) {
          if (MODE & 4) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (MODE & 4) {
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const int j = i < 4 ? wave + 4 * i : 16 + (wave & 1);
            dma16(wsrc + (gs / 3) * 18432 + j * 1024 + lane * 16, smem_lds + (unsigned)(8704 * 16 + j * 1024));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) { LOAD((gs + 1) & 1, (gs + 1) % 9) }
      if ((MODE & 8) && (gs == 2 || gs == 3)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf16x8 h, m, l;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            float v = __builtin_amdgcn_fmed3f(__fmaf_rn(qv[u][i + 2 * (gs - 2)], sc, sh), 0.0f, 1e30f);
            h[u] = (__bf16)v;
            const float r1 = v - (float)h[u];
            m[u] = (__bf16)r1;
            l[u] = (__bf16)(r1 - (float)m[u]);
          }
          bf16x8* d = lds8 + 7168 + tid + i * 256;      // (above the fragment area, below the DMA target)
          d[0] = h; d[512] = m; d[1024] = l;
        }
      }
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = (PA[p] == 0 && PB[p] == 0) ? 0 : 1;
          acc[k][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[(MODE & 1) ? gs & 1 : 0][PB[p]][t & 1], fa[(MODE & 1) ? gs & 1 : 0][PA[p]][t >> 1],
                                                              acc[k][t], 0, 0, 0);
        }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
#pragma unroll
      for (int k = 12; k < 24; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[k][t][r];
  if (s == 12345.678f) out[tid] = s;
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* wsrc, float* out, long long* cyc, int blocks, int ncg) {
  auto kern = stream<MODE>;
  const size_t lds = 160 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  kern<<<blocks, 256, lds>>>(wsrc, out, cyc, ncg, 0.001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kern<<<blocks, 256, lds>>>(wsrc, out, cyc, ncg, 0.001f, 0.5f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0; long long mx = 0;
  for (long long v : h) { sum += (double)v; if (v > mx) mx = v; }
  const double mfma = (double)ncg * 9 * 24;
  const double flops_eq = (double)blocks * 4 * mfma * 32768.0 / 6.0;      // fp32-equivalent: six products per fp32 product
  printf("{\"mode\": %d, \"ms\": %.3f, \"counter_ticks_per_mfma_avg\": %.2f, \"counter_ticks_per_mfma_max\": %.2f, \"fp32_equiv_tflops\": %.1f, "
         "\"bf16_tflops\": %.0f}\n", MODE, ms, sum / blocks / mfma, mx / mfma, flops_eq / ms / 1e9, 6.0 * flops_eq / ms / 1e9);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount;
  char* wsrc; float* out; long long* cyc;
  hipMalloc(&wsrc, 3 * 18432 + 4096);
  hipMemset(wsrc, 0x3c, 3 * 18432 + 4096);
  hipMalloc(&out, 256 * 4);
  hipMalloc(&cyc, blocks * sizeof(long long));
  printf("{\"cus\": %d, \"clock_mhz\": %.0f, \"what\": \"modes: bit0 fragment reads, bit1 barrier per 3 steps, bit2 LDS-DMA behind the barrier, bit3 patch conversion\"}\n",
         blocks, p.clockRate / 1000.0);
  const int ncg = 400;
  run<0>(wsrc, out, cyc, blocks, ncg);
  run<1>(wsrc, out, cyc, blocks, ncg);
  run<3>(wsrc, out, cyc, blocks, ncg);
  run<7>(wsrc, out, cyc, blocks, ncg);
  run<9>(wsrc, out, cyc, blocks, ncg);
  run<15>(wsrc, out, cyc, blocks, ncg);
  return 0;
}
