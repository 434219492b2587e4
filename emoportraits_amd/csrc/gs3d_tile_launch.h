// Host side of the LDS-staged 3-D grid_sample (gs3d_tile.h): kernel, tuning word, launch.  Included by one translation unit
// per padding mode (gs3d_tile_pad*.hip) so that the 3 x 3 x 3 x 3 instantiations compile in parallel.
#pragma once
#include <mutex>
#include <set>
#include <utility>
#include "gs3d_tile.h"

namespace gs3d {

constexpr int TILE_MAXI = 10;          // slots per thread and unit: stage capacity up to 10 * THREADS slots (40 / 80 KiB)

// 4 waves per SIMD: 4 blocks of 256 threads or 2 of 512 per CU (LDS: 4 x 40 KiB or 2 x 80 KiB)
template <int PAD, int MODE, bool IN_P4, bool OUT_P4, int THREADS, int VPT>
__global__ __launch_bounds__(THREADS, 4) void tile_kernel(const TileParams prm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gs3d_smem[];
  TileThread<PAD, MODE, IN_P4, OUT_P4, THREADS, VPT, TILE_MAXI> t;
  t.init(prm, gs3d_smem, blockIdx.x, gridDim.x, threadIdx.x);
  __syncthreads();
  t.taps();
  __syncthreads();
  const int npass = t.plan_passes();
  for (int ps = 0; ps < npass; ++ps) {
    if (!t.plan(ps)) { t.direct(ps); continue; }
    for (int u0 = t.u_begin; u0 < t.u_end; u0 += t.nu) {
      t.fill(u0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
      __syncthreads();                                    // ... and everybody else's
      t.gather(ps, u0);
      __syncthreads();                                    // the stage may be overwritten
    }
  }
}

// opt in to more than 64 KiB of dynamic LDS once per (kernel, device); keyed by the function pointer value
template <typename K>
static int raise_dynamic_lds(K kern) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> raised;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return EMO_ERR_UNSUPPORTED;
  const std::pair<const void*, int> key(reinterpret_cast<const void*>(kern), dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!raised.count(key)) {
    e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return EMO_ERR_UNSUPPORTED;   // (not a raw hipError_t: callers see EMO_* codes only)
    raised.insert(key);
  }
  return EMO_OK;
}

template <int PAD, int MODE, bool IN_P4, bool OUT_P4, int THREADS, int VPT>
int launch_tile_cfg(const TileParams& p, size_t lds_bytes, hipStream_t s) {
  const long total = (long)p.ngroups * p.N * p.ntx * p.nty * p.ntz;
  if (total > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  auto kern = tile_kernel<PAD, MODE, IN_P4, OUT_P4, THREADS, VPT>;
  if (lds_bytes > 64 * 1024) {
    const int rc = raise_dynamic_lds(kern);
    if (rc != EMO_OK) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(THREADS), lds_bytes, s, p);
  return emo_launch_status();
}

// Tuning word of the tile kernels (the `variant` argument of emo_grid_sample3d_f32; 0 = defaults):
//   bits  3..0  log2 tile x     7..4  log2 tile y     11..8  log2 tile z     (all three 0: default tile)
//   bits 16..12 channel units per block (0: default)  24..17 LDS per block in KiB (0: default)
//   bit  25     512 threads per block instead of 256
template <int PAD, int MODE, bool IN_P4, bool OUT_P4>
int launch_tile(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                long vol_bstride, int variant, hipStream_t s) {
  if (IN_P4 && (C % 4)) return EMO_ERR_UNSUPPORTED;
  if (!IN_P4 && (W % 4)) return EMO_ERR_UNSUPPORTED;                            // 16-byte pieces of the planar box rows
  if (W > 2046 || H > 2046 || D > 1022) return EMO_ERR_UNSUPPORTED;              // packed floor corner
  if ((long)D * H * W * (IN_P4 ? 16 : 4) >= (1L << 31)) return EMO_ERR_UNSUPPORTED;   // 32-bit byte offsets inside a unit
  if ((long)Do * Ho * Wo * 16 >= (1L << 32)) return EMO_ERR_UNSUPPORTED;         // 32-bit store offsets
  TileParams p;
  p.vol = vol; p.grid = grid; p.theta = theta; p.lin_x = lin_x; p.lin_y = lin_y; p.lin_z = lin_z; p.out = out;
  p.N = N; p.C = C; p.D = D; p.H = H; p.W = W; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
  p.vol_bstride = vol_bstride;
  p.units = IN_P4 ? C / 4 : C;
  int txs = variant & 15, tys = (variant >> 4) & 15, tzs = (variant >> 8) & 15;
  int upb = (variant >> 12) & 31, lds_kib = (variant >> 17) & 255;
  int threads = (variant >> 25) & 1 ? 512 : 256;
  if (txs + tys + tzs == 0) {
    // default tile: 4 x 8 x 8 output voxels, one per thread -- the best of the sweeps for both calls at 16 frames per launch
    // (archive/profiles/r3_sampler_tile_sweep_n16.jsonl: rotation 15.6 us P4 -> P4, uv 11.6 us); larger tiles stage smaller boxes per
    // voxel but leave too few blocks
    txs = 3; tys = 3; tzs = 2; threads = 256;
    while (tzs > 0 && (1 << tzs) >= 2 * Do) { --tzs; ++tys; }
    while (tys > 0 && (1 << tys) >= 2 * Ho) { --tys; ++txs; }
  }
  int lv = txs + tys + tzs - (threads == 512 ? 9 : 8);                            // log2(VPT)
  if (lv < 0 || lv > 1) return EMO_ERR_BAD_ARG;
  // all channel units of a tile in one block: the per-block planning (taps, box, slot map: ~800 VALU + 3900 SALU instructions
  // per wave) is the kernel's largest fixed cost (3 units per block: 43 us per frame, all 24: 23 us)
  if (upb == 0) upb = p.units < 31 ? p.units : 31;
  if (upb > p.units) upb = p.units;
  if (lds_kib == 0) lds_kib = threads == 512 ? 80 : 40;
  if (lds_kib > 160 || lds_kib < 1) return EMO_ERR_BAD_ARG;
  p.txs = txs; p.tys = tys; p.tzs = tzs;
  p.ntx = emo_cdiv(Wo, 1 << txs); p.nty = emo_cdiv(Ho, 1 << tys); p.ntz = emo_cdiv(Do, 1 << tzs);
  p.upb = upb;
  p.ngroups = emo_cdiv(p.units, upb);
  const size_t lds = (size_t)lds_kib * 1024;
  // header + scratch + a stage of at least 64 slots, or the tuning word is rejected (an unsigned underflow here would let the
  // kernel stage far beyond the LDS it was given; a tile whose box does not fit the stage takes the kernel's direct-gather path)
  const size_t lds_fixed = (size_t)TILE_HDR_BYTES + tile_scratch_bytes(IN_P4, OUT_P4, threads);
  if (lds < lds_fixed + 1024) return EMO_ERR_BAD_ARG;
  p.cap_slots = (int)((lds - lds_fixed) / 16);
  if (p.cap_slots > TILE_MAXI * threads) p.cap_slots = TILE_MAXI * threads;
  if (threads == 256) {
    if (lv == 0) return launch_tile_cfg<PAD, MODE, IN_P4, OUT_P4, 256, 1>(p, lds, s);
    return launch_tile_cfg<PAD, MODE, IN_P4, OUT_P4, 256, 2>(p, lds, s);
  }
  if (lv == 0) return EMO_ERR_BAD_ARG;
  return launch_tile_cfg<PAD, MODE, IN_P4, OUT_P4, 512, 2>(p, lds, s);
}

template <int PAD, int MODE>
int launch_tile_layout(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                       const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                       long vol_bstride, int in_layout, int out_layout, int variant, hipStream_t s) {
#define EMO_GS3D_ARGS vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride, variant, s
  if (in_layout == EMO_LAYOUT_P4 && out_layout == EMO_LAYOUT_P4) return launch_tile<PAD, MODE, true, true>(EMO_GS3D_ARGS);
  if (in_layout == EMO_LAYOUT_P4 && out_layout == EMO_LAYOUT_NCDHW) return launch_tile<PAD, MODE, true, false>(EMO_GS3D_ARGS);
  if (in_layout == EMO_LAYOUT_NCDHW && out_layout == EMO_LAYOUT_NCDHW) return launch_tile<PAD, MODE, false, false>(EMO_GS3D_ARGS);
#undef EMO_GS3D_ARGS
  return EMO_ERR_UNSUPPORTED;
}

template <int PAD>
int launch_tile_pad(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                    const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                    long vol_bstride, int in_layout, int out_layout, int variant, int grid_kind, hipStream_t s) {
  if (theta)
    return launch_tile_layout<PAD, MODE_THETA>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo,
                                               vol_bstride, in_layout, out_layout, variant, s);
  if (grid_kind == 1)
    return launch_tile_layout<PAD, MODE_DELTA>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo,
                                               vol_bstride, in_layout, out_layout, variant, s);
  return launch_tile_layout<PAD, MODE_GRID>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo,
                                            vol_bstride, in_layout, out_layout, variant, s);
}

}  // namespace gs3d

#define EMO_GS3D_TILE_PAD_SIGNATURE(name)                                                                              \
  int name(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,            \
           const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo, long vol_bstride, \
           int in_layout, int out_layout, int variant, int grid_kind, hipStream_t s)
