// What a bare MFMA stream sustains on this part, by instruction shape and by operand DATA (gfx950; standalone:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_power.hip -o tools/microbench/mfma_power && tools/microbench/mfma_power
// One block of 4 waves per CU, one wave per SIMD, 8 independent accumulator tiles per wave, operands in registers (no LDS, no
// memory traffic inside the timed loop).  Every variant issues the same number of matrix-pipe cycles per iteration, so
// "TFLOP/s" and "effective clock" (issued matrix cycles / time: the rate is issue-bound, one MFMA behind the other) read
// the power management directly.  Operand data: "random" (uniform in [-2, 2)), "small" (the low-order term of a two-term
// split: random mantissas at 2^-11 of the scale), "zeros".  JSON lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16; 3 / 4: 32x32x16 f16 with one / both operands held
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void stream_kernel(const unsigned* __restrict__ operands, float* __restrict__ sink, int iters) {
  const int tid = threadIdx.x;
  // 8 operand registers sets of 16 bytes per lane, from memory once
  unsigned raw[8][4];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[k][u] = operands[((k * 256 + tid) * 4 + u) & 8191];
  float s = 0.0f;
  if (SHAPE == 1) {
    floatx4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = floatx4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)      // 4 x 8 x 2 = 64 MFMAs of 16 cycles = 1024 matrix cycles per iteration
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const halfx8 a = __builtin_bit_cast(halfx8, *reinterpret_cast<const uint4*>(raw[(t + r) & 7]));
          const halfx8 b = __builtin_bit_cast(halfx8, *reinterpret_cast<const uint4*>(raw[(t + 2 * r + 1) & 7]));
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
  } else {
    floatx16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)      // 4 x 8 = 32 MFMAs of 32 cycles = 1024 matrix cycles per iteration
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (SHAPE == 0 || SHAPE == 3 || SHAPE == 4) {
            // SHAPE 3: operand A stays for the 8 consecutive MFMAs of the t loop (only B changes from one instruction to the next);
            // SHAPE 4: both operands stay (8 x the same product into 8 accumulators): how much of the energy is operand TOGGLING
            const int ia = SHAPE == 0 ? (t + r) & 7 : r, ib = SHAPE == 4 ? (r + 4) & 7 : (t + 2 * r + 1) & 7;
            const halfx8 a = __builtin_bit_cast(halfx8, *reinterpret_cast<const uint4*>(raw[ia]));
            const halfx8 b = __builtin_bit_cast(halfx8, *reinterpret_cast<const uint4*>(raw[ib]));
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
          } else {
            const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(raw[(t + r) & 7]));
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(raw[(t + 2 * r + 1) & 7]));
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][15];
  }
  if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  unsigned* d_op;
  float* d_sink;
  (void)hipMalloc(&d_op, 8192 * 4);
  (void)hipMalloc(&d_sink, (size_t)ncu * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const char* shapes[5] = {"v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_bf16",
                           "v_mfma_f32_32x32x16_f16, operand A held over 8 consecutive instructions",
                           "v_mfma_f32_32x32x16_f16, both operands held over 8 consecutive instructions"};
  const char* datas[3] = {"random", "small", "zeros"};
  for (int shape = 0; shape < 5; ++shape)
    for (int data = 0; data < 3; ++data) {
      std::vector<unsigned> h(8192);
      unsigned st = 12345u + 77u * data;
      for (int i = 0; i < 8192; ++i) {
        unsigned short v[2];
        for (int j = 0; j < 2; ++j) {
          st = st * 1664525u + 1013904223u;
          float f = ((st >> 8) & 0xffff) * (4.0f / 65536.0f) - 2.0f;
          if (data == 1) f *= 1.0f / 2048.0f;
          if (data == 2) f = 0.0f;
          v[j] = shape == 2 ? f2b(f) : f2h(f);
        }
        h[i] = (unsigned)v[0] | ((unsigned)v[1] << 16);
      }
      (void)hipMemcpy(d_op, h.data(), 8192 * 4, hipMemcpyHostToDevice);
      auto launch = [&](int iters) {
        if (shape == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(ncu), dim3(256), 0, 0, d_op, d_sink, iters);
        if (shape == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(ncu), dim3(256), 0, 0, d_op, d_sink, iters);
        if (shape == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(ncu), dim3(256), 0, 0, d_op, d_sink, iters);
        if (shape == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(ncu), dim3(256), 0, 0, d_op, d_sink, iters);
        if (shape == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(ncu), dim3(256), 0, 0, d_op, d_sink, iters);
      };
      const int iters = 400000;     // x 1024 matrix cycles: ~0.2-0.3 s per launch
      launch(iters);                // warm: the clock settles under load
      launch(iters);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, 0);
      for (int r = 0; r < 3; ++r) launch(iters);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double cycles = 3.0 * iters * 1024.0;                 // matrix-pipe cycles per SIMD
      const double flops = cycles / 32.0 * 32768.0 * 4.0 * ncu;   // 32768 flop per 32 cycles per SIMD, all shapes
      printf("{\"shape\": \"%s\", \"operands\": \"%s\", \"tflops\": %.1f, \"effective_clock_ghz\": %.3f, \"seconds\": %.3f, \"cus\": %d}\n",
             shapes[shape], datas[data], flops / (ms * 1e-3) / 1e12, cycles / (ms * 1e-3) / 1e9, ms * 1e-3, ncu);
      fflush(stdout);
    }
  return 0;
}
