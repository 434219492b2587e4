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
#include <stdio.h>
#include <stdlib.h>
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
#ifdef HIPSHIM_THREADS
#include <stdlib.h>
// "compute units": persistent kernels launch one block each (HIPSHIM_CUS, default 8: few blocks, several chained items per block)
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
  const char* e = getenv("HIPSHIM_CUS");
  const int n = e ? atoi(e) : 8;
  *v = n > 0 ? n : 8;
  return hipSuccess;
}
#else
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
#endif

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
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0ull; }
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }
// raw buffer resource of the kernels (base in words 0 / 1, no range limit) and LDS byte addresses (offsets into the block's buffer)
static inline const char* hipshim_buffer_base(int w0, int w1) {
  return reinterpret_cast<const char*>(((unsigned long long)(unsigned)w0) | (((unsigned long long)(unsigned)w1 & 0xffffull) << 32));
}
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
#include <atomic>
#include <memory>
#include <sched.h>
#include <thread>
#include <vector>
// A block is 256 OS threads on however few cores the test machine has, and an emulated MFMA is an exchange between the 64
// threads of a wave: a futex barrier (sleep / wake) costs a millisecond there.  This one spins on its generation word and
// yields the core while it waits.
struct hipshim_barrier {
  std::atomic<unsigned> count{0}, generation{0};
  unsigned n = 1;
};
static inline void pthread_barrier_init(hipshim_barrier* b, void*, unsigned n) { b->n = n; b->count = 0; b->generation = 0; }
static inline void pthread_barrier_destroy(hipshim_barrier*) {}
static inline void pthread_barrier_wait(hipshim_barrier* b) {
  const unsigned gen = b->generation.load(std::memory_order_acquire);
  if (b->count.fetch_add(1, std::memory_order_acq_rel) + 1 == b->n) {
    b->count.store(0, std::memory_order_relaxed);
    b->generation.store(gen + 1, std::memory_order_release);
  } else {
    while (b->generation.load(std::memory_order_acquire) == gen) sched_yield();
  }
}
inline namespace hipshim_threaded {
inline dim3 blockIdx, blockDim, gridDim;
inline thread_local dim3 threadIdx;
}
namespace hipshim {
struct Block {
  hipshim_barrier all;
  std::unique_ptr<hipshim_barrier[]> wave;
  std::vector<unsigned long long> slot;      // one 8-byte exchange slot per thread
  std::vector<unsigned long long> wide;      // 32 bytes per thread: the operands of a 16-bit MFMA
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
// The lanes of a wave run in lock step on the GPU: code may exchange data through LDS between the lanes of ONE wave without a
// barrier in the source.  Here they are independent threads; the tests' rewrites put this barrier of the wave where a kernel
// relies on it (the epilogues' transposition through a wave-private LDS region).
static inline void hipshim_wave_sync() { pthread_barrier_wait(&hipshim::cur->wave[hipshim::linear_tid() / 64]); }
// LDS-DMA of the split kernels (global_load_lds_dwordx4 with M0 = lds_dst): 16 bytes per lane to LDS byte lds_dst + lane * 16
static inline void hipshim_lds_dma16(const void* gsrc, unsigned lds_dst) {
  const unsigned lane = hipshim::linear_tid() % 64;
  memcpy(hipshim_dyn_smem_ptr + lds_dst + lane * 16, gsrc, 16);
}
// DPP row operations the kernels use (v_mov_b32_dpp with all rows and banks enabled): quad_perm (ctrl < 0x100), row_mirror
// (0x140), row_half_mirror (0x141) -- lane l of a 16-lane row reads another lane of its row
static inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
  const unsigned t = hipshim::linear_tid(), w = t / 64, lane = t % 64;
  hipshim::cur->slot[t] = (unsigned)src;
  pthread_barrier_wait(&hipshim::cur->wave[w]);
  unsigned from;
  if (ctrl < 0x100) from = (lane & ~3u) | (((unsigned)ctrl >> (2 * (lane & 3))) & 3u);
  else if (ctrl == 0x140) from = (lane & ~15u) | (15u - (lane & 15u));
  else if (ctrl == 0x141) from = (lane & ~7u) | (7u - (lane & 7u));
  else { from = lane; __builtin_trap(); }
  const int got = (int)(unsigned)hipshim::cur->slot[w * 64 + from];
  pthread_barrier_wait(&hipshim::cur->wave[w]);
  return got;
}
// v_mfma_f32_32x32x16_{f16,bf16}: lane l supplies A[l & 31][8 (l >> 5) + 0..7] and B[8 (l >> 5) + 0..7][l & 31]; the result
// layout is that of v_mfma_f32_32x32x2_f32.  Products of 16-bit operands are exact in fp32; they are added to the accumulator in
// the order of k (the hardware's internal order and width are not modelled: results are compared within tolerances, never bits).
template <typename V8, typename V16>
static inline V16 hipshim_mfma_32x32x16(V8 a, V8 b, V16 c) {
  const unsigned t = hipshim::linear_tid(), w = t / 64, lane = t % 64;
  static_assert(sizeof(V8) == 16, "eight 16-bit operands per lane");
  memcpy(&hipshim::cur->wide[4 * (size_t)t], &a, 16);
  memcpy(&hipshim::cur->wide[4 * (size_t)t + 2], &b, 16);
  pthread_barrier_wait(&hipshim::cur->wave[w]);
  const unsigned j = lane & 31;
  for (unsigned r = 0; r < 16; ++r) {
    const unsigned i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (unsigned kh = 0; kh < 2; ++kh) {
      V8 av, bv;
      memcpy(&av, &hipshim::cur->wide[4 * (size_t)(w * 64 + 32 * kh + i)], 16);
      memcpy(&bv, &hipshim::cur->wide[4 * (size_t)(w * 64 + 32 * kh + j) + 2], 16);
      for (int e = 0; e < 8; ++e) acc += (float)av[e] * (float)bv[e];
    }
    c[r] = acc;
  }
  pthread_barrier_wait(&hipshim::cur->wave[w]);
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipshim_mfma_32x32x16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipshim_mfma_32x32x16(a, b, c)
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
  if (getenv("HIPSHIM_TRACE")) fprintf(stderr, "hipshim launch: grid %u x %u x %u, block %u threads, %zu bytes of dynamic LDS\n", grid.x, grid.y, grid.z, nthreads, (size_t)shmem);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        hipshim_smem_buffer smem_(shmem);
        hipshim::Block blk;
        pthread_barrier_init(&blk.all, nullptr, nthreads);
        blk.wave.reset(new hipshim_barrier[nwaves]);
        for (unsigned w = 0; w < nwaves; ++w) {
          const unsigned n = w + 1 < nwaves ? 64 : nthreads - 64 * w;
          pthread_barrier_init(&blk.wave[w], nullptr, n);
        }
        blk.slot.assign(nthreads, 0ull);
        blk.wide.assign(4 * (size_t)nthreads, 0ull);
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
