// TEST INFRASTRUCTURE ONLY -- a stand-in for <hip/hip_runtime.h> that lets g++ compile the barrier-free HIP kernels of
// emoportraits_amd/csrc (resample.hip, conv_head.hip: the SAME sources the product is built from) as host C++ and run them
// thread by thread (tests/emul/stream_kernels_emul.cpp, tests/test_stream_kernels_emul.py).  A launch is a loop over blocks and
// threads; every thread runs to completion before the next one starts, so kernels that exchange data between threads
// (__shfl_*, __syncthreads + shared memory) do NOT compute what they compute on the GPU here: those intrinsics are stubs that
// keep such kernels compiling, and the tests only look at results that do not pass through them.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }

static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
template <typename T> static inline T __shfl_down(T v, int, int = 64) { return v; }   // stub (see the header comment)
template <typename T> static inline T __shfl_xor(T v, int, int = 64) { return v; }    // stub
static inline void __syncthreads() {}                                                  // stub

template <typename K, typename... A>
static inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx) {
              threadIdx = dim3(tx, ty, tz);
              kernel(args...);
            }
      }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
