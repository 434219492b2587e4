// Microbenchmark: rate of L2-resident global loads per CU for 4 / 8 / 16 bytes per lane (wave-contiguous addresses), at the
// occupancy of the fp16-operand conv kernel (2 blocks of 256 threads per CU).  Answers whether the patch staging of
// conv_igemm_f16.h (dword loads, 256 B per wave-instruction) is bound by instructions or by bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/vmem_rate.hip -o tools/microbench/vmem_rate && tools/microbench/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W>   // dwords per lane
__global__ __launch_bounds__(256) void loads(const float* __restrict__ src, float* __restrict__ out, int iters, int window_dw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = src + (size_t)blockIdx.x * window_dw;
  float acc = 0.f;
  unsigned off = (wave * 64 + lane) * W;            // wave-contiguous, W dwords per lane
  const unsigned stride = 256 * W;                  // the 4 waves of the block advance together
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned o = (off + u * stride) % (unsigned)window_dw;
      if (W == 1) acc += base[o];
      if (W == 2) { float2 v = *reinterpret_cast<const float2*>(base + o); acc += v.x + v.y; }
      if (W == 4) { float4 v = *reinterpret_cast<const float4*>(base + o); acc += v.x + v.y + v.z + v.w; }
    }
    off += 8 * stride;
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int W>
static void run(const float* src, float* out, int blocks, int iters, int window_dw, int ncu, double mhz) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  loads<W><<<blocks, 256>>>(src, out, iters, window_dw);
  hipDeviceSynchronize();
  hipEventRecord(a);
  loads<W><<<blocks, 256>>>(src, out, iters, window_dw);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double)blocks * 4 * iters * 8;          // wave-instructions
  const double bytes = instr * 64 * 4 * W;
  const double clk = ms * 1e-3 * mhz * 1e6;
  printf("{\"dwords_per_lane\": %d, \"ms\": %.3f, \"wave_instr_per_clk_per_cu\": %.4f, \"bytes_per_clk_per_cu\": %.1f, \"TBps\": %.2f}\n",
         W, ms, instr / clk / ncu, bytes / clk / ncu, bytes / ms / 1e9);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  const int blocks = ncu * 2, window_dw = 8192;     // 32 KB per block: 16 MB in all, L2-resident, larger than a CU's L1 share
  float *src, *out;
  hipMalloc(&src, (size_t)blocks * window_dw * 4);
  hipMalloc(&out, 4);
  hipMemset(src, 0, (size_t)blocks * window_dw * 4);
  printf("{\"cus\": %d, \"clock_mhz\": %.0f, \"blocks\": %d, \"window_bytes\": %d}\n", ncu, mhz, blocks, window_dw * 4);
  for (int w : {2048, 8192}) {   // 8 KB per block: L1-resident; 32 KB per block: L2
    printf("{\"window_bytes\": %d}\n", w * 4);
    run<1>(src, out, blocks, 2048, w, ncu, mhz);
    run<2>(src, out, blocks, 2048, w, ncu, mhz);
    run<4>(src, out, blocks, 2048, w, ncu, mhz);
  }
  return 0;
}
