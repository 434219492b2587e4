// TEST INFRASTRUCTURE ONLY -- a stand-in for <hip/hip_runtime.h> that lets g++ compile the barrier-free HIP kernels of
// emoportraits_amd/csrc (resample.hip, conv_head.hip: the SAME sources the product is built from) as host C++ and run them
// thread by thread (tests/emul/stream_kernels_emul.cpp, tests/test_stream_kernels_emul.py).  A launch is a loop over blocks and
// threads; every thread runs to completion before the next one starts, so kernels that exchange data between threads
// (__shfl_*, __syncthreads + shared memory) do NOT compute what they compute on the GPU in this (default, fast) mode: those
// intrinsics are stubs that keep such kernels compiling, and the tests only look at results that do not pass through them.
// With -DHIPSHIM_THREADS the threads of a block are OS threads, __syncthreads is a barrier and __shfl_* exchange between the
// lanes of a 64-thread wave: block reductions compute what they compute on the GPU (tiny grids: a thread per GPU thread).
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

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
// Wave votes are evaluated PER LANE (as if every lane were a wave of its own).  Valid for the product's uses only because the
// branches they select are result-equivalent per lane: the samplers' `__all(all eight corners in range)` picks between an
// unmasked and a masked accumulation that perform the same operations on the same operands for a lane whose own corners are
// all in range (csrc/grid_sample3d.hip: gather_quad).
static inline bool __all(bool pred) { return pred; }
// dynamic shared memory: the launch's `shmem` bytes, one buffer per block (a kernel's `extern __shared__ T name[];` is rewritten
// by the test into `T* const name = (T*)hipshim_dynamic_smem();`)
inline unsigned char* hipshim_dyn_smem_ptr = nullptr;
static inline void* hipshim_dynamic_smem() { return hipshim_dyn_smem_ptr; }
struct hipshim_smem_buffer {
  unsigned char* raw;
  explicit hipshim_smem_buffer(size_t n) : raw(new unsigned char[n + 64]) {
    hipshim_dyn_smem_ptr = raw + (64 - reinterpret_cast<uintptr_t>(raw) % 64) % 64;
    memset(hipshim_dyn_smem_ptr, 0xff, n);
  }
  ~hipshim_smem_buffer() { delete[] raw; hipshim_dyn_smem_ptr = nullptr; }
};

enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// scalar-unit / scheduler intrinsics of the kernels that mean nothing on the host
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }      // (the product only applies it to wave-uniform values)
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {   // v_med3_f32: the median of three
  return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}

static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }

#ifndef HIPSHIM_THREADS
// ---- sequential mode: one thread after the other; kernels that exchange data between threads are NOT modelled ----
// (the built-in variables live in a namespace of the mode: inline variables are process-wide unique symbols, and a test process
// loads a library of each mode)
inline namespace hipshim_sequential {
inline dim3 threadIdx, blockIdx, blockDim, gridDim;
}
template <typename T> static inline T __shfl_down(T v, int, int = 64) { return v; }   // stub (see the header comment)
template <typename T> static inline T __shfl_xor(T v, int, int = 64) { return v; }    // stub
static inline void __syncthreads() {}                                                  // stub

template <typename K, typename... A>
static inline void emu_launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
  gridDim = grid;
  blockDim = block;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        hipshim_smem_buffer smem_(shmem);
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx) {
              threadIdx = dim3(tx, ty, tz);
              kernel(args...);
            }
      }
}
#else
// ---- threaded mode (-DHIPSHIM_THREADS): the threads of a block are OS threads, one block at a time.  __syncthreads is a
//      barrier of the block; __shfl_* exchange through a slot per lane behind barriers of the 64-thread wave (every lane of a
//      wave that has not returned must reach the shuffle, as on the GPU).  Slow (a thread per GPU thread): tiny grids only ----
#include <pthread.h>
#include <thread>
#include <vector>
inline namespace hipshim_threaded {
inline dim3 blockIdx, blockDim, gridDim;
inline thread_local dim3 threadIdx;
}
namespace hipshim {
struct Block {
  pthread_barrier_t all;
  std::vector<pthread_barrier_t> wave;
  std::vector<unsigned long long> slot;      // one 8-byte exchange slot per thread
};
inline Block* cur = nullptr;
inline unsigned linear_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }
template <typename T> static inline T shfl(T v, int src_lane_delta, bool xor_mode, int width) {
  static_assert(sizeof(T) <= 8, "4- or 8-byte shuffles");
  const unsigned t = linear_tid(), w = t / 64, lane = t % 64;
  unsigned long long bits = 0;
  memcpy(&bits, &v, sizeof(T));
  cur->slot[t] = bits;
  pthread_barrier_wait(&cur->wave[w]);
  int src = xor_mode ? (int)(lane ^ (unsigned)src_lane_delta) : (int)lane + src_lane_delta;
  const int seg = (int)lane / width * width;
  if (src < seg || src >= seg + width) src = (int)lane;                      // out of the segment: the lane's own value
  unsigned long long got = cur->slot[w * 64 + (unsigned)src];
  pthread_barrier_wait(&cur->wave[w]);
  T out;
  memcpy(&out, &got, sizeof(T));
  return out;
}
}  // namespace hipshim
// v_mfma_f32_32x32x2_f32 as an exchange between the 64 lanes of a wave: D[i][j] += A[i][0] B[0][j], then += A[i][1] B[1][j] (a
// k-ordered chain of fused multiply-adds); lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31] and holds, in register r,
// D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].  Every lane of the wave must execute it (as on the GPU).
template <typename V16>
static inline V16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, V16 c, int, int, int) {
  const unsigned t = hipshim::linear_tid(), w = t / 64, lane = t % 64;
  float ab[2] = {a, b};
  memcpy(&hipshim::cur->slot[t], ab, 8);
  pthread_barrier_wait(&hipshim::cur->wave[w]);
  auto A = [&](unsigned i, unsigned k) { float v[2]; memcpy(v, &hipshim::cur->slot[w * 64 + 32 * k + i], 8); return v[0]; };
  auto B = [&](unsigned k, unsigned j) { float v[2]; memcpy(v, &hipshim::cur->slot[w * 64 + 32 * k + j], 8); return v[1]; };
  const unsigned j = lane & 31;
  for (unsigned r = 0; r < 16; ++r) {
    const unsigned i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    c[r] = fmaf(A(i, 1), B(1, j), fmaf(A(i, 0), B(0, j), c[r]));
  }
  pthread_barrier_wait(&hipshim::cur->wave[w]);
  return c;
}
// global_load_lds (LDS-DMA): every lane copies `size` bytes from its own global address to LDS at the wave-uniform base the
// instruction names + offset + lane * size
template <typename G, typename L>
static inline void __builtin_amdgcn_global_load_lds(G gptr, L ldsptr, unsigned size, unsigned offset, unsigned) {
  const unsigned lane = hipshim::linear_tid() % 64;
  memcpy(reinterpret_cast<char*>((uintptr_t)ldsptr) + offset + lane * size, reinterpret_cast<const char*>((uintptr_t)gptr), size);
}
template <typename T> static inline T __shfl_down(T v, int delta, int width = 64) { return hipshim::shfl(v, delta, false, width); }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { return hipshim::shfl(v, mask, true, width); }
static inline void __syncthreads() { pthread_barrier_wait(&hipshim::cur->all); }

template <typename K, typename... A>
static inline void emu_launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
  gridDim = grid;
  blockDim = block;
  const unsigned nthreads = block.x * block.y * block.z, nwaves = (nthreads + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        hipshim_smem_buffer smem_(shmem);
        hipshim::Block blk;
        pthread_barrier_init(&blk.all, nullptr, nthreads);
        blk.wave.resize(nwaves);
        for (unsigned w = 0; w < nwaves; ++w) {
          const unsigned n = w + 1 < nwaves ? 64 : nthreads - 64 * w;
          pthread_barrier_init(&blk.wave[w], nullptr, n);
        }
        blk.slot.assign(nthreads, 0ull);
        hipshim::cur = &blk;
        std::vector<std::thread> ts;
        ts.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; ++t)
          ts.emplace_back([=]() {
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            kernel(args...);
          });
        for (auto& th : ts) th.join();
        for (unsigned w = 0; w < nwaves; ++w) pthread_barrier_destroy(&blk.wave[w]);
        pthread_barrier_destroy(&blk.all);
        hipshim::cur = nullptr;
      }
}
#endif
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, (size_t)(shmem), __VA_ARGS__)
