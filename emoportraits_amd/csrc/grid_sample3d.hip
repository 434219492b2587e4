// 3-D trilinear grid_sample (align_corners=False) for gfx950 -- SURVEY.md section 8 rows a1 + a2.
//
// Replaces  F.grid_sample(inputs.float(), grid.float(), padding_mode=...)  on 5-D tensors
// (reference: models/stage_1/volumetric_avatar/va.py:264-265; call sites notebooks/infer.py:499-500,
// :618-619).  The arithmetic restated here is ATen's CPU grid_sampler_3d (ATen/native/GridSampler.h:
// grid_sampler_unnormalize :27-36, clip_coordinates :58-60, reflect_coordinates :89-106) and is kept
// bit-identical to it: all coordinate/weight math is fp32 with explicit round-to-nearest intrinsics (no FMA
// contraction), corner weights are (wx*wy)*wz, corners accumulate in the order tnw,tne,tsw,tse,bnw,bne,
// bsw,bse with separate multiply and add, and out-of-range corners contribute nothing (zeros padding).
//
// Two data layouts:
//   NCDHW (reference layout): one lane per output voxel, loop over a chunk of channels.  Taps/weights are
//     computed once per voxel and reused for every channel; output stores are coalesced along x.
//   NDHWC (channels-last, internal): the C channels of a voxel are contiguous, so every corner is ONE
//     contiguous C*4-byte read (384 B for C=96) regardless of how the warp scatters neighbouring voxels --
//     each lane owns 4 channels (float4) of one voxel.  The canonical volume is repacked once per identity.
//
// The analytic variant (theta != NULL) fuses the head-pose affine of the identity lattice into the sampler
// (row a2: notebooks/infer.py:441-444, :583-588), removing the 0.79 MB grid tensor from HBM.
#include "gs3d_coord.h"

#define EMO_GS3D_TILE_FLAG (1 << 30)   /* variant bit: NCDHW -> NCDHW through the LDS-staged planar kernel (gs3d_tile.h) */

#ifndef EMO_GS3D_NT_STORES
#define EMO_GS3D_NT_STORES 0   /* A/B: non-temporal stores of the channels-last sampler output.  Measured (profiles/
                                  r3_sampler_nt_stores_ab.jsonl): the uv call itself does not change (12.8 vs 13.1 us per frame), the
                                  rotation call that reads the intermediate back in chunks of 4 frames goes from 13.1 to 17.3 us --
                                  the plain stores leave it in the Infinity Cache, the non-temporal ones do not.  Off. */
#endif
typedef float emo_f32x4 __attribute__((ext_vector_type(4)));
#ifndef EMO_GS3D_ABLATE
#define EMO_GS3D_ABLATE 0      /* timing experiments only (results are WRONG for any value != 0): 1 no corner gathers, 2 no output
                                  stores of the channels-last kernels, 4 no NCDHW stores of the rotation kernel (archive/profiles/r3_sampler_ablation.jsonl) */
#endif
#if EMO_GS3D_ABLATE & 2
#define EMO_GS3D_STORE(ptr, val)                                                                                       \
  do {                                                                                                                 \
    const float4 v_ = (val);                                                                                           \
    if (v_.x == 12345.678f) *(ptr) = v_;                                                                               \
  } while (0)
#elif EMO_GS3D_NT_STORES
#define EMO_GS3D_STORE(ptr, val)                                                                                       \
  do {                                                                                                                 \
    const float4 v_ = (val);                                                                                           \
    emo_f32x4 n_ = {v_.x, v_.y, v_.z, v_.w};                                                                          \
    __builtin_nontemporal_store(n_, reinterpret_cast<emo_f32x4*>(ptr));                                               \
  } while (0)
#else
#define EMO_GS3D_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace {

struct Taps {
  int off[8];      // spatial offset (z*H + y)*W + x of each corner, 0 when the corner is out of range
  float w[8];      // corner weights, reference order tnw,tne,tsw,tse,bnw,bne,bsw,bse
  unsigned inb;    // bit k set <=> corner k is inside the volume
};

using gs3d::MODE_DELTA;
using gs3d::MODE_GRID;
using gs3d::MODE_THETA;
using gs3d::load_coord;
using gs3d::xcd_remap;

// floor corner + weights (gs3d_coord.h, bit-identical to ATen) -> per-corner offsets and the in-range mask
template <int PAD>
__device__ __forceinline__ void compute_taps(float gx, float gy, float gz, int D, int H, int W, Taps& t) {
  int x0, y0, z0;
  gs3d::corner_weights<PAD>(gx, gy, gz, D, H, W, x0, y0, z0, t.w);
  const int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
  const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
  const bool vz0 = (unsigned)z0 < (unsigned)D, vz1 = (unsigned)z1 < (unsigned)D;
  const bool v[8] = {vz0 && vy0 && vx0, vz0 && vy0 && vx1, vz0 && vy1 && vx0, vz0 && vy1 && vx1,
                     vz1 && vy0 && vx0, vz1 && vy0 && vx1, vz1 && vy1 && vx0, vz1 && vy1 && vx1};
  const int HW = H * W;
  const int o[8] = {z0 * HW + y0 * W + x0, z0 * HW + y0 * W + x1, z0 * HW + y1 * W + x0, z0 * HW + y1 * W + x1,
                    z1 * HW + y0 * W + x0, z1 * HW + y0 * W + x1, z1 * HW + y1 * W + x0, z1 * HW + y1 * W + x1};
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    t.off[k] = v[k] ? o[k] : 0;
    m |= v[k] ? (1u << k) : 0u;
  }
  t.inb = m;
}

// ------------------------------------------------------------------------------------------------------
// NCDHW -> NCDHW.  grid = (ceil(nvox/256), channel chunks, N); one lane per output voxel.
// ------------------------------------------------------------------------------------------------------
template <int PAD, int MODE>
__global__ __launch_bounds__(256) void gs3d_ncdhw_kernel(
    const float* __restrict__ vol, const float* __restrict__ grid, const float* __restrict__ theta,
    const float* __restrict__ lin_x, const float* __restrict__ lin_y, const float* __restrict__ lin_z,
    float* __restrict__ out, int C, int D, int H, int W, int Do, int Ho, int Wo, long vol_bstride, int c_per_block) {
  const int nvox = Do * Ho * Wo;
  const int vox = blockIdx.x * 256 + threadIdx.x;
  if (vox >= nvox) return;
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * c_per_block;
  const int c1 = min(c0 + c_per_block, C);
  float gx, gy, gz;
  load_coord<MODE>(grid, theta, lin_x, lin_y, lin_z, n, vox, nvox, Ho, Wo, gx, gy, gz);
  Taps t;
  compute_taps<PAD>(gx, gy, gz, D, H, W, t);
  const long DHW = (long)D * H * W;
  const float* vp = vol + (long)n * vol_bstride + (long)c0 * DHW;
  float* op = out + ((long)n * C + c0) * nvox + vox;
  int c = c0;
  for (; c + 4 <= c1; c += 4) {
    float v[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[j][k] = vp[j * DHW + t.off[k]];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float val = ((t.inb >> k) & 1u) ? v[j][k] : 0.0f;
        acc = __fadd_rn(acc, __fmul_rn(val, t.w[k]));
      }
      op[(long)j * nvox] = acc;
    }
    vp += 4 * DHW;
    op += 4L * nvox;
  }
  for (; c < c1; ++c) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float raw = vp[t.off[k]];
      const float val = ((t.inb >> k) & 1u) ? raw : 0.0f;
      acc = __fadd_rn(acc, __fmul_rn(val, t.w[k]));
    }
    op[0] = acc;
    vp += DHW;
    op += nvox;
  }
}

// ------------------------------------------------------------------------------------------------------
// v2 channels-last kernels: taps are computed ONCE per voxel (64 voxels per block, one lane each), parked in
// LDS, and every (voxel, channel-quad) item then only does: 5 broadcast LDS reads, 8 x 16-byte gathers,
// 64 mul/add.  v1 recomputed the ~150-instruction tap math in each of the C/4 lanes of a voxel and was
// VALU-bound (15 us/sample vs a 5 us L1-bandwidth bound).  Out-of-range corners are handled by a wave-uniform
// branch: the common all-in-range wave takes a path with no per-component selects.
// ------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) TapRec {
  int off[8];
  float w[8];
  unsigned inb;
  unsigned pad[3];
};
// Block order of one launch.  ORDER 1: XCD-contiguous (gs3d::xcd_remap: block b runs on XCD b % 8, each XCD has a private
// 4 MiB L2; remapped, XCD k walks one contiguous eighth of the (sample, z, y, x) ordered work, so a corner row is fetched
// once and re-used out of that XCD's L2 by the following rows and the next z-slice).  ORDER 3: XCD-contiguous AND, inside an
// XCD's share, (z-slice, group of 8 blocks, sample, block in group) -- needs N % 8 == 0: when N driver frames sample ONE
// shared canonical volume with near-identity warps, a group's corner rows (a few hundred KB of the shared volume) stay in
// L2 while all of the XCD's samples pass over them.  (Other orders, 128 / 256 voxels per block, the first-generation
// kernels and the channel-group-per-XCD layout were measured in rounds 1-2 and retired: DESIGN.md section 3.2.)
template <int ORDER>
__device__ __forceinline__ void block_to_work(int b, int total, int bps, int nslices, int& n, int& blk) {
  constexpr int G = 8;                          // blocks (= output rows at Wo = 64) per row group
  if (ORDER == 1) { const int L = xcd_remap(b, total); n = L / bps; blk = L - n * bps; return; }
  const int xcd = b & 7, idx = b >> 3;
  const int spx = (total / bps) >> 3;          // samples per XCD
  const int per_group = spx * G;
  const int grp = idx / per_group;              // global group index = z * (bpz / G) + row group
  const int rem = idx - grp * per_group;
  const int j = rem / G;
  n = xcd * spx + j;
  blk = grp * G + (rem - j * G);
}

// voxel i (0..63) of 4x4x4 brick `blk` of a Do x Ho x Wo output lattice (all three multiples of 4): bricks x-fastest
__device__ __forceinline__ int brick_voxel(int blk, int i, int Ho, int Wo) {
  const int bw = Wo >> 2, bh = Ho >> 2;
  const int bx = blk % bw;
  const int t = blk / bw;
  const int by = t % bh, bz = t / bh;
  return (((bz << 2) + (i >> 4)) * Ho + (by << 2) + ((i >> 2) & 3)) * Wo + (bx << 2) + (i & 3);
}

// BRICK: the block's voxels are a 4x4x4 brick (vox0 = brick index) instead of VPB consecutive voxels
template <int PAD, int MODE, int VPB, bool BRICK = false>
__device__ __forceinline__ void stage_taps(TapRec* __restrict__ recs, const float* __restrict__ grid,
                                           const float* __restrict__ theta, const float* __restrict__ lin_x,
                                           const float* __restrict__ lin_y, const float* __restrict__ lin_z, int n,
                                           int vox0, int nvox, int D, int H, int W, int Ho, int Wo) {
  for (int i = threadIdx.x; i < VPB; i += 256) {
    const int vox = BRICK ? brick_voxel(vox0, i, Ho, Wo) : vox0 + i;
    Taps t;
    if (vox < nvox) {
      float gx, gy, gz;
      load_coord<MODE>(grid, theta, lin_x, lin_y, lin_z, n, vox, nvox, Ho, Wo, gx, gy, gz);
      compute_taps<PAD>(gx, gy, gz, D, H, W, t);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) { t.off[k] = 0; t.w[k] = 0.0f; }
      t.inb = 0;
    }
    TapRec r;
#pragma unroll
    for (int k = 0; k < 8; ++k) { r.off[k] = t.off[k]; r.w[k] = t.w[k]; }
    r.inb = t.inb; r.pad[0] = r.pad[1] = r.pad[2] = 0;
    recs[i] = r;
  }
}

// FMA (variant bit 4 of the channels-last kernels, opt-in): coordinates, floor, corner indices and the eight weights are the
// same bit-exact arithmetic; only the accumulation changes from ATen's separate multiply and add per corner (64 VALU
// instructions per item) to acc = fma(v, w, acc) in packed form (16 v_pk_fma_f32).  Values then differ from ATen's CPU kernel
// by the product roundings that are skipped: <= 8 * 2^-24 * max |v w| absolute (tests/test_grid_sample_gpu.py).
typedef float emo_f2 __attribute__((ext_vector_type(2)));
template <bool FMA = false>
__device__ __forceinline__ float4 gather_quad(const char* __restrict__ vbytes, const TapRec& r, unsigned row_bytes,
                                              unsigned qbyte) {
  float4 v[8];
#if EMO_GS3D_ABLATE & 1     /* timing experiment (wrong results): no corner gathers */
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = make_float4(r.w[k], r.w[(k + 1) & 7], (float)r.off[k], (float)qbyte);
#else
#pragma unroll
  for (int k = 0; k < 8; ++k)
    v[k] = *reinterpret_cast<const float4*>(vbytes + ((unsigned)r.off[k] * row_bytes + qbyte));
#endif
  if constexpr (FMA) {
    emo_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    if (__all(r.inb == 0xffu)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const emo_f2 w2 = {r.w[k], r.w[k]};
        a01 = __builtin_elementwise_fma(emo_f2{v[k].x, v[k].y}, w2, a01);
        a23 = __builtin_elementwise_fma(emo_f2{v[k].z, v[k].w}, w2, a23);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool in = (r.inb >> k) & 1u;
        const emo_f2 w2 = {r.w[k], r.w[k]};
        a01 = __builtin_elementwise_fma(emo_f2{in ? v[k].x : 0.0f, in ? v[k].y : 0.0f}, w2, a01);
        a23 = __builtin_elementwise_fma(emo_f2{in ? v[k].z : 0.0f, in ? v[k].w : 0.0f}, w2, a23);
      }
    }
    return make_float4(a01[0], a01[1], a23[0], a23[1]);
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // wave-uniform: does every lane of this wave have all 8 corners in range?
  if (__all(r.inb == 0xffu)) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = r.w[k];
      acc.x = __fadd_rn(acc.x, __fmul_rn(v[k].x, w));
      acc.y = __fadd_rn(acc.y, __fmul_rn(v[k].y, w));
      acc.z = __fadd_rn(acc.z, __fmul_rn(v[k].z, w));
      acc.w = __fadd_rn(acc.w, __fmul_rn(v[k].w, w));
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool in = (r.inb >> k) & 1u;
      const float w = r.w[k];
      acc.x = __fadd_rn(acc.x, __fmul_rn(in ? v[k].x : 0.0f, w));
      acc.y = __fadd_rn(acc.y, __fmul_rn(in ? v[k].y : 0.0f, w));
      acc.z = __fadd_rn(acc.z, __fmul_rn(in ? v[k].z : 0.0f, w));
      acc.w = __fadd_rn(acc.w, __fmul_rn(in ? v[k].w : 0.0f, w));
    }
  }
  return acc;
}

// (voxel, quad) of item = threadIdx.x + 256 * k without a division per item: the runtime divisor LPV = C / 4 costs ~25 VALU
// instructions per division, a third of an item's instruction count (the kernels' arithmetic skeleton is 40 % of their time,
// archive/profiles/r3_sampler_ablation.jsonl); one division per thread, then (v, q) advance by (256 / LPV, 256 % LPV) with a carry
struct ItemWalk {
  int v, q, dv, dq, lpv;
  __device__ __forceinline__ ItemWalk(int tid, int LPV) : lpv(LPV) {
    v = tid / LPV; q = tid - v * LPV;
    dv = 256 / LPV; dq = 256 - dv * LPV;
  }
  __device__ __forceinline__ void next() {
    v += dv; q += dq;
    if (q >= lpv) { q -= lpv; ++v; }
  }
};

// NDHWC -> NDHWC, 1-D grid of N * ceil(nvox/VPB) blocks
template <int PAD, int MODE, int VPB, int ORDER, bool FMA>
__global__ __launch_bounds__(256) void gs3d_cl_v2_kernel(
    const float* __restrict__ vol, const float* __restrict__ grid, const float* __restrict__ theta,
    const float* __restrict__ lin_x, const float* __restrict__ lin_y, const float* __restrict__ lin_z,
    float* __restrict__ out, int C, int D, int H, int W, int Do, int Ho, int Wo, long vol_bstride, int bps) {
  __shared__ TapRec recs[VPB];
  const int LPV = C >> 2;
  const int nvox = Do * Ho * Wo;
  int n, blk;
  block_to_work<ORDER>(blockIdx.x, gridDim.x, bps, Do, n, blk);
  const int vox0 = blk * VPB;
  stage_taps<PAD, MODE, VPB>(recs, grid, theta, lin_x, lin_y, lin_z, n, vox0, nvox, D, H, W, Ho, Wo);
  __syncthreads();
  const char* vbytes = reinterpret_cast<const char*>(vol + (long)n * vol_bstride);
  const unsigned row_bytes = (unsigned)C * 4u;
  const int nitems = min(VPB, nvox - vox0) * LPV;
  float4* obase = reinterpret_cast<float4*>(out) + ((long)n * nvox + vox0) * LPV;
  ItemWalk w(threadIdx.x, LPV);
  for (int item = threadIdx.x; item < nitems; item += 256, w.next()) {
    const TapRec r = recs[w.v];
    EMO_GS3D_STORE(&obase[item], gather_quad<FMA>(vbytes, r, row_bytes, (unsigned)w.q * 16u));
  }
}

// NDHWC -> NDHWC with 4x4x4 output bricks per block: the 8 corners of a brick's 64 voxels fall into ~5x5x5 source voxels
// (48 KB at C = 96) instead of the 2x2x65 of an x-row (100 KB): half the L1 fills per output voxel.  The samplers are
// bound by the per-CU L1 path (64-byte granules per clock + outstanding-miss capacity), not by L2 or HBM
// (archive/profiles/r2_pmc_sampler_*.json), so fewer fills per voxel is the lever.  Needs Do, Ho, Wo multiples of 4.
template <int PAD, int MODE, int ORDER>
__global__ __launch_bounds__(256) void gs3d_cl_brick_kernel(
    const float* __restrict__ vol, const float* __restrict__ grid, const float* __restrict__ theta,
    const float* __restrict__ lin_x, const float* __restrict__ lin_y, const float* __restrict__ lin_z,
    float* __restrict__ out, int C, int D, int H, int W, int Do, int Ho, int Wo, long vol_bstride, int bps) {
  __shared__ TapRec recs[64];
  __shared__ int vox_of[64];
  const int LPV = C >> 2;
  const int nvox = Do * Ho * Wo;
  int n, blk;
  block_to_work<ORDER>(blockIdx.x, gridDim.x, bps, Do >> 2, n, blk);
  stage_taps<PAD, MODE, 64, true>(recs, grid, theta, lin_x, lin_y, lin_z, n, blk, nvox, D, H, W, Ho, Wo);
  if (threadIdx.x < 64) vox_of[threadIdx.x] = brick_voxel(blk, threadIdx.x, Ho, Wo);
  __syncthreads();
  const char* vbytes = reinterpret_cast<const char*>(vol + (long)n * vol_bstride);
  const unsigned row_bytes = (unsigned)C * 4u;
  float4* obase = reinterpret_cast<float4*>(out) + (long)n * nvox * LPV;
  ItemWalk w(threadIdx.x, LPV);
  for (int item = threadIdx.x; item < 64 * LPV; item += 256, w.next()) {
    const TapRec r = recs[w.v];
    EMO_GS3D_STORE(&obase[(long)vox_of[w.v] * LPV + w.q], gather_quad(vbytes, r, row_bytes, (unsigned)w.q * 16u));
  }
}

// NDHWC -> NCDHW, 1-D grid; dynamic LDS = VPB tap records + C * (VPB + 1) floats (transpose tile)
template <int PAD, int MODE, int VPB, int ORDER, bool FMA>
__global__ __launch_bounds__(256) void gs3d_cl2ncdhw_v2_kernel(
    const float* __restrict__ vol, const float* __restrict__ grid, const float* __restrict__ theta,
    const float* __restrict__ lin_x, const float* __restrict__ lin_y, const float* __restrict__ lin_z,
    float* __restrict__ out, int C, int D, int H, int W, int Do, int Ho, int Wo, long vol_bstride, int bps, int nt_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  TapRec* recs = reinterpret_cast<TapRec*>(smem);                 // VPB * 80 B
  float* tile = smem + VPB * (sizeof(TapRec) / 4);                 // [C][VPB + 1]
  constexpr int LD = VPB + 1;
  const int LPV = C >> 2;
  const int nvox = Do * Ho * Wo;
  int n, blk;
  block_to_work<ORDER>(blockIdx.x, gridDim.x, bps, Do, n, blk);
  const int vox0 = blk * VPB;
  stage_taps<PAD, MODE, VPB>(recs, grid, theta, lin_x, lin_y, lin_z, n, vox0, nvox, D, H, W, Ho, Wo);
  __syncthreads();
  const char* vbytes = reinterpret_cast<const char*>(vol + (long)n * vol_bstride);
  const unsigned row_bytes = (unsigned)C * 4u;
  const int nv = min(VPB, nvox - vox0);
  const int nitems = nv * LPV;
  ItemWalk w(threadIdx.x, LPV);
  for (int item = threadIdx.x; item < nitems; item += 256, w.next()) {
    const int v = w.v, q = w.q;
    const TapRec r = recs[v];
    const float4 acc = gather_quad<FMA>(vbytes, r, row_bytes, (unsigned)q * 16u);
    const int c = q * 4;
    tile[(c + 0) * LD + v] = acc.x;
    tile[(c + 1) * LD + v] = acc.y;
    tile[(c + 2) * LD + v] = acc.z;
    tile[(c + 3) * LD + v] = acc.w;
  }
  __syncthreads();
  float* obase = out + (long)n * C * nvox + vox0;
  for (int i = threadIdx.x; i < C * VPB; i += 256) {
    const int c = i / VPB;
    const int v = i - c * VPB;
    if (v < nv && (!(EMO_GS3D_ABLATE & 4) || tile[c * LD + v] == 12345.678f)) {
      // nt_out: the NCDHW result is streamed past the caches -- in the driver pass it is read much later (by the decoder),
      // while the Infinity Cache is what serves the intermediate of the sampler pair (measured: pair - 5 % at chunks of 4)
      if (nt_out) __builtin_nontemporal_store(tile[c * LD + v], &obase[(long)c * nvox + v]);
      else obase[(long)c * nvox + v] = tile[c * LD + v];
    }
  }
}

constexpr int CL_VPB = 64;      // output voxels per block of the row-shaped kernels

template <int PAD, int MODE, int ORDER, bool FMA = false>
int launch_cl_v2(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                 const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                 long vol_bstride, bool out_cl, hipStream_t s, int nt_out = 0) {
  const int nvox = Do * Ho * Wo;
  const int bps = emo_cdiv(nvox, CL_VPB);
  const long total = (long)bps * N;
  if (total > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  if (ORDER == 3 && ((N & 7) || (bps % Do) || (nvox % CL_VPB) || ((bps / Do) % 8)))   // whole slices / groups, N % 8 == 0
    return launch_cl_v2<PAD, MODE, 1, FMA>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                           out_cl, s, nt_out);
  if (out_cl) {
    hipLaunchKernelGGL((gs3d_cl_v2_kernel<PAD, MODE, CL_VPB, ORDER, FMA>), dim3((unsigned)total), dim3(256), 0, s, vol, grid,
                       theta, lin_x, lin_y, lin_z, out, C, D, H, W, Do, Ho, Wo, vol_bstride, bps);
  } else {
    const size_t lds = CL_VPB * sizeof(TapRec) + (size_t)C * (CL_VPB + 1) * sizeof(float);
    if (lds > 64 * 1024) return EMO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gs3d_cl2ncdhw_v2_kernel<PAD, MODE, CL_VPB, ORDER, FMA>), dim3((unsigned)total), dim3(256), lds, s, vol,
                       grid, theta, lin_x, lin_y, lin_z, out, C, D, H, W, Do, Ho, Wo, vol_bstride, bps, nt_out);
  }
  return emo_launch_status();
}

// variant word of the channels-last kernels (bits): 1 = 4 x 4 x 4 output bricks per block instead of 64-voxel rows (NDHWC output,
// Do / Ho / Wo multiples of 4): half the L1 fills per voxel, the uv call alone 11.7 instead of 12.3 us per frame -- but the
// rotation call that reads its output back is then 0.6-0.8 us slower (its blocks walk the volume in row order), so the pair
// is 1.3 us per frame slower: opt-in.  2 = non-temporal stores of an NCDHW output (the driver pass's rotation call).
// 4 = fused multiply-add accumulation (gather_quad<true>: taps bit-identical, values within 8 * 2^-24 * max|v w| of ATen).
// Default: 64-voxel rows, XCD-contiguous, and for a volume shared by N % 8 == 0 samples also row-group-major over the XCD's
// samples.  (archive/profiles/r3_sampler_nt_stores_ab.jsonl, r3_sampler_nt_out_ab.jsonl)
template <int PAD, int MODE>
int dispatch_cl_v2(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                   const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                   long vol_bstride, bool out_cl, int variant, hipStream_t s) {
  if (variant < 0 || variant > 7) return EMO_ERR_BAD_ARG;
  const int nt_out = (variant & 2) && !out_cl;
  if ((variant & 1) && out_cl && !(Do & 3) && !(Ho & 3) && !(Wo & 3)) {
    const int bps = (Do * Ho * Wo) >> 6;
    const long total = (long)bps * N;
    if (total > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gs3d_cl_brick_kernel<PAD, MODE, 1>), dim3((unsigned)total), dim3(256), 0, s, vol, grid, theta, lin_x,
                       lin_y, lin_z, out, C, D, H, W, Do, Ho, Wo, vol_bstride, bps);
    return emo_launch_status();
  }
  if (variant & 4) {      // fused multiply-add accumulation (gather_quad<true>): row kernels only
    if (vol_bstride == 0 && N >= 8)
      return launch_cl_v2<PAD, MODE, 3, true>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                              out_cl, s, nt_out);
    return launch_cl_v2<PAD, MODE, 1, true>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                            out_cl, s, nt_out);
  }
  if (vol_bstride == 0 && N >= 8)
    return launch_cl_v2<PAD, MODE, 3>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                      out_cl, s, nt_out);
  return launch_cl_v2<PAD, MODE, 1>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                    out_cl, s, nt_out);
}

// ------------------------------------------------------------------------------------------------------
// layout repack [N][C][S] <-> [N][S][C] through a 64x64 LDS tile
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     int R, int Ccols) {
  // in: [n][R][Ccols] -> out: [n][Ccols][R]
  __shared__ float tile[64][65];
  const int n = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float* ip = in + (long)n * R * Ccols;
  float* op = out + (long)n * R * Ccols;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    if (r < R && c < Ccols) tile[i][tx] = ip[(long)r * Ccols + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < Ccols) op[(long)c * R + r] = tile[tx][i];
  }
}

template <int PAD, int MODE>
int launch(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
           const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
           long vol_bstride, int in_layout, int out_layout, int variant, hipStream_t s) {
  const int nvox = Do * Ho * Wo;
  if (in_layout == EMO_LAYOUT_NCDHW && out_layout == EMO_LAYOUT_NCDHW) {
    int cpb = variant > 0 ? variant : 8;
    if (cpb > C) cpb = C;
    dim3 g(emo_cdiv(nvox, 256), emo_cdiv(C, cpb), N);
    hipLaunchKernelGGL((gs3d_ncdhw_kernel<PAD, MODE>), g, dim3(256), 0, s, vol, grid, theta, lin_x, lin_y, lin_z,
                       out, C, D, H, W, Do, Ho, Wo, vol_bstride, cpb);
  } else if (in_layout == EMO_LAYOUT_NDHWC && (out_layout == EMO_LAYOUT_NDHWC || out_layout == EMO_LAYOUT_NCDHW)) {
    if (C % 4) return EMO_ERR_UNSUPPORTED;
    if ((long)D * H * W * C * 4 >= (1L << 32)) return EMO_ERR_UNSUPPORTED;   // 32-bit byte offsets
    return dispatch_cl_v2<PAD, MODE>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                     out_layout == EMO_LAYOUT_NDHWC, variant, s);
  } else {
    return EMO_ERR_UNSUPPORTED;
  }
  return emo_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// The rotation warp itself (row a2): grid[n,z,y,x,:] = theta[n,:3,:4] . (u_x, v_y, w_z, 1) -- exactly the coordinates
// the MODE_THETA samplers generate on the fly, materialised for callers that want the reference's cached
// `source_rotation_warp` tensor (notebooks/infer.py:441-444) and for the parity tests that compare them with the
// reference's `identity_grid_3d.bmm(theta[:, :3].transpose(1, 2))`.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_grid3d_kernel(const float* __restrict__ theta,
                                                            const float* __restrict__ lin_x,
                                                            const float* __restrict__ lin_y,
                                                            const float* __restrict__ lin_z, float* __restrict__ grid,
                                                            int Do, int Ho, int Wo) {
  const int nvox = Do * Ho * Wo;
  const int vox = blockIdx.x * 256 + threadIdx.x;
  if (vox >= nvox) return;
  const int n = blockIdx.y;
  float gx, gy, gz;
  load_coord<MODE_THETA>(nullptr, theta, lin_x, lin_y, lin_z, n, vox, nvox, Ho, Wo, gx, gy, gz);
  float* g = grid + ((long)n * nvox + vox) * 3;
  g[0] = gx; g[1] = gy; g[2] = gz;
}

template <int PAD>
int launch_pad(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
               const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
               long vol_bstride, int in_layout, int out_layout, int variant, int grid_kind, hipStream_t s) {
  if (theta)
    return launch<PAD, MODE_THETA>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                   in_layout, out_layout, variant, s);
  if (grid_kind == 1)
    return launch<PAD, MODE_DELTA>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                   in_layout, out_layout, variant, s);
  return launch<PAD, MODE_GRID>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_bstride,
                                in_layout, out_layout, variant, s);
}

}  // namespace

int emo_gs3d_tile_dispatch(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                           const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                           int64_t vol_batch_stride, int padding_mode, int in_layout, int out_layout, int variant,
                           int grid_kind, void* stream);
int emo_repack_p4_dispatch(const float* in, float* out, int N, int C, int DHW, int to_p4, void* stream);

extern "C" int emo_grid_sample3d_f32(const float* vol, const float* grid, const float* theta, const float* lin_x,
                                     const float* lin_y, const float* lin_z, float* out, int N, int C, int D, int H,
                                     int W, int Do, int Ho, int Wo, int64_t vol_batch_stride, int padding_mode,
                                     int in_layout, int out_layout, int variant, int grid_kind, void* stream) {
  if (!vol || !out || (!grid && !theta)) return EMO_ERR_BAD_ARG;
  if (grid_kind != 0 && grid_kind != 1) return EMO_ERR_BAD_ARG;
  if ((theta || grid_kind == 1) && (!lin_x || !lin_y || !lin_z)) return EMO_ERR_BAD_ARG;
  if (theta && grid_kind == 1) return EMO_ERR_BAD_ARG;
  if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return EMO_ERR_BAD_ARG;
  if (vol_batch_stride < 0) return EMO_ERR_BAD_ARG;
  if (N > 65535) return EMO_ERR_UNSUPPORTED;
  if ((long)D * H * W >= (1L << 31) / 4 || (long)Do * Ho * Wo >= (1L << 31) / 4) return EMO_ERR_UNSUPPORTED;
  if (!emo_aligned16(vol) || !emo_aligned16(out)) return EMO_ERR_ALIGN;
  if (in_layout == EMO_LAYOUT_P4)
    return emo_gs3d_tile_dispatch(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_batch_stride,
                                  padding_mode, in_layout, out_layout, variant & ~EMO_GS3D_TILE_FLAG, grid_kind, stream);
  if (in_layout == EMO_LAYOUT_NCDHW && out_layout == EMO_LAYOUT_NCDHW && (variant & EMO_GS3D_TILE_FLAG)) {
    // LDS-staged planar kernel; shapes it does not take (W % 4, huge extents) fall through to the direct-gather kernel
    const int rc = emo_gs3d_tile_dispatch(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo, vol_batch_stride,
                                          padding_mode, in_layout, out_layout, variant & ~EMO_GS3D_TILE_FLAG, grid_kind, stream);
    if (rc != EMO_ERR_UNSUPPORTED) return rc;
    variant = 0;
  }
  hipStream_t s = (hipStream_t)stream;
  switch (padding_mode) {
    case EMO_PAD_ZEROS:
      return launch_pad<EMO_PAD_ZEROS>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo,
                                       vol_batch_stride, in_layout, out_layout, variant, grid_kind, s);
    case EMO_PAD_BORDER:
      return launch_pad<EMO_PAD_BORDER>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo,
                                        vol_batch_stride, in_layout, out_layout, variant, grid_kind, s);
    case EMO_PAD_REFLECTION:
      return launch_pad<EMO_PAD_REFLECTION>(vol, grid, theta, lin_x, lin_y, lin_z, out, N, C, D, H, W, Do, Ho, Wo,
                                            vol_batch_stride, in_layout, out_layout, variant, grid_kind, s);
    default:
      return EMO_ERR_BAD_ARG;
  }
}

extern "C" int emo_affine_grid3d_f32(const float* theta, const float* lin_x, const float* lin_y, const float* lin_z,
                                     float* grid, int N, int Do, int Ho, int Wo, void* stream) {
  if (!theta || !lin_x || !lin_y || !lin_z || !grid) return EMO_ERR_BAD_ARG;
  if (N <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return EMO_ERR_BAD_ARG;
  if (N > 65535 || (long)Do * Ho * Wo >= (1L << 31) / 4) return EMO_ERR_UNSUPPORTED;
  dim3 g(emo_cdiv((long)Do * Ho * Wo, 256), N);
  hipLaunchKernelGGL(affine_grid3d_kernel, g, dim3(256), 0, (hipStream_t)stream, theta, lin_x, lin_y, lin_z, grid, Do, Ho, Wo);
  return emo_launch_status();
}

extern "C" int emo_volume_repack_f32(const float* in, float* out, int N, int C, int DHW, int to_channels_last,
                                     void* stream) {
  if (!in || !out || N <= 0 || C <= 0 || DHW <= 0) return EMO_ERR_BAD_ARG;
  if (N > 65535) return EMO_ERR_UNSUPPORTED;
  if (to_channels_last == 4 || to_channels_last == 5)     // NCDHW <-> packed-4
    return emo_repack_p4_dispatch(in, out, N, C, DHW, to_channels_last == 4, stream);
  if (to_channels_last != 0 && to_channels_last != 1) return EMO_ERR_BAD_ARG;
  const int R = to_channels_last ? C : DHW;       // rows of the input matrix
  const int Ccols = to_channels_last ? DHW : C;   // columns of the input matrix
  dim3 g(emo_cdiv(Ccols, 64), emo_cdiv(R, 64), N);
  if (g.y > 65535) return EMO_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(repack_kernel, g, dim3(256), 0, (hipStream_t)stream, in, out, R, Ccols);
  return emo_launch_status();
}
