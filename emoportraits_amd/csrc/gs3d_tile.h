// LDS-staged 3-D trilinear grid_sample for gfx950 -- SURVEY.md section 8 rows a1 + a2, the kernel BASELINE.json's north_star
// names ("a fused NCDHW 3D trilinear grid_sample with LDS-staged volume tiles and coalesced HBM reads").
// Replaces F.grid_sample(inputs.float(), grid.float(), padding_mode=...) on 5-D tensors
// (models/stage_1/volumetric_avatar/va.py:264-265; call sites notebooks/infer.py:499-500, :618-619).
//
// Why: a direct gather asks the per-CU vector L1 for 8 corners per output value -- 8 x the output bytes, 64 bytes per clock
// per CU -- and that request rate, not HBM, bounded the round-1/2 samplers.  Here a workgroup owns a
// TILE of output voxels; the source BOX that holds all corners of the tile is brought into LDS once, by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip, many KiB in flight per CU), and the 8 x gather
// runs out of LDS (ds_read_b128: 256 B per clock per CU).  The L1 sees box/tile ~ 2-3.5 x the output bytes instead of 8 x,
// all of it as full-granule coalesced rows.
//
//   work item   (channel group g, sample n, output tile): THREADS * VPT output voxels, tile extents are powers of two;
//               thread t owns voxels j * THREADS + t (j < VPT) of the tile -- "brick" j is a contiguous slab of the tile.
//   taps        per owned voxel, once: floor corner (packed x0+1 | y0+1 | z0+1), 8 weights (gs3d_coord.h, bit-identical
//               to ATen), in registers for all channel units of the block.  Voxels whose 8 corners are all outside the
//               volume are "dead" (output exactly 0) and do not widen the box.
//   box         integer bounding box of the live floor corners (+1), by LDS atomics per brick.  It includes one voxel of
//               zeros around the volume where corners stick out (zeros padding: an out-of-range corner adds 0 * w, which
//               leaves the fp32 accumulator unchanged bit for bit).
//   passes      if the union box of all bricks fits the LDS stage it is staged once for all VPT voxels of a thread;
//               otherwise brick by brick; a brick whose own box does not fit (wild warps) is sampled straight from
//               global memory with the same arithmetic -- correct for any input, fast for spatially coherent warps.
//   units       IN_P4: a unit is a channel QUAD of the packed-4 layout [N][C/4][D][H][W][4] -- one box voxel = one 16-byte
//               slot, one lane of the DMA.  Planar (NCDHW): a unit is one channel, a slot = 4 x-consecutive floats (box
//               x-range aligned to 4).  As many units as fit the stage are filled together, then gathered: one barrier
//               pair per stage.
//   output      OUT_P4: 16-byte stores in the packed-4 layout (feeds the next sampler); else NCDHW, the reference layout.
//
// The same source also compiles as host C++ (GS3D_HOST_EMULATION): tests/emul runs the phases thread by thread and checks
// them against the oracle bit for bit without a GPU.
#pragma once
#include "gs3d_coord.h"

#ifndef GS3D_HOST_EMULATION
#include <limits.h>
#endif

namespace gs3d {

struct TileParams {
  const float* vol;
  const float* grid;       // MODE_GRID: [N,Do,Ho,Wo,3]; MODE_DELTA: [N,3,Do,Ho,Wo]
  const float* theta;      // MODE_THETA: [N,3,4]
  const float* lin_x;
  const float* lin_y;
  const float* lin_z;
  float* out;
  int N, C, D, H, W, Do, Ho, Wo;
  long vol_bstride;        // floats between consecutive volumes (0: one volume shared by all samples)
  int txs, tys, tzs;       // log2 of the tile extents; 1 << (txs + tys + tzs) == THREADS * VPT
  int ntx, nty, ntz;       // tiles per axis
  int units;               // channel units: C / 4 (packed-4 input) or C (planar input)
  int upb;                 // units per block
  int ngroups;             // ceil(units / upb)
  int cap_slots;           // capacity of the LDS stage in 16-byte slots
};

constexpr int TILE_HDR_BYTES = 128;   // box min/max of up to 4 bricks
// LDS of a block: [header][16 bytes per thread: output transpose scratch, packed-4 -> NCDHW only][stage]
constexpr int tile_scratch_bytes(bool in_p4, bool out_p4, int threads) { return (in_p4 && !out_p4) ? threads * 16 : 0; }

#if defined(GS3D_HOST_EMULATION)
#define GS_UNIFORM(x) (x)
#else
#define GS_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   /* value is the same in every lane: keep it in an SGPR */
#endif

#if defined(GS3D_HOST_EMULATION)
#define GS_SCHED_BARRIER()
#else
#define GS_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

GS_FN int imin(int a, int b) { return a < b ? a : b; }
GS_FN int imax(int a, int b) { return a > b ? a : b; }
// floor / ceil to multiples of 4 for possibly negative values
GS_FN int floor4(int v) { return v & ~3; }

#if defined(GS3D_HOST_EMULATION)
struct f4 { float x, y, z, w; };
GS_FN f4 lds_read16(const unsigned char* lds, int off) { f4 v; memcpy(&v, lds + off, 16); return v; }
GS_FN float lds_read4(const unsigned char* lds, int off) { float v; memcpy(&v, lds + off, 4); return v; }
GS_FN void lds_zero16(unsigned char* lds, int off) { memset(lds + off, 0, 16); }
GS_FN void lds_write16(unsigned char* lds, int off, f4 v) { memcpy(lds + off, &v, 16); }
GS_FN int lds_geti(const unsigned char* lds, int off) { int v; memcpy(&v, lds + off, 4); return v; }
GS_FN void lds_seti(unsigned char* lds, int off, int v) { memcpy(lds + off, &v, 4); }
GS_FN void lds_atomic_min(unsigned char* lds, int off, int v) { int c = lds_geti(lds, off); if (v < c) lds_seti(lds, off, v); }
GS_FN void lds_atomic_max(unsigned char* lds, int off, int v) { int c = lds_geti(lds, off); if (v > c) lds_seti(lds, off, v); }
GS_FN f4 glob_read16(const float* p) { f4 v; memcpy(&v, p, 16); return v; }
GS_FN void glob_write16(float* p, f4 v) { memcpy(p, &v, 16); }
// LDS-DMA of one lane: 16 bytes from (base + byte offset) to (wave base + lane * 16)
GS_FN void dma16(unsigned char* lds, int lds_wave_base, int lane, const float* sbase, int voff) {
  memcpy(lds + lds_wave_base + lane * 16, reinterpret_cast<const unsigned char*>(sbase) + voff, 16);
}
#else
typedef float4 f4;
GS_FN f4 lds_read16(const unsigned char* lds, int off) { return *reinterpret_cast<const f4*>(lds + off); }
GS_FN float lds_read4(const unsigned char* lds, int off) { return *reinterpret_cast<const float*>(lds + off); }
GS_FN void lds_zero16(unsigned char* lds, int off) { *reinterpret_cast<f4*>(lds + off) = make_float4(0.f, 0.f, 0.f, 0.f); }
GS_FN void lds_write16(unsigned char* lds, int off, f4 v) { *reinterpret_cast<f4*>(lds + off) = v; }
GS_FN int lds_geti(const unsigned char* lds, int off) { return *reinterpret_cast<const int*>(lds + off); }
GS_FN void lds_seti(unsigned char* lds, int off, int v) { *reinterpret_cast<int*>(lds + off) = v; }
GS_FN void lds_atomic_min(unsigned char* lds, int off, int v) { atomicMin(reinterpret_cast<int*>(lds + off), v); }
GS_FN void lds_atomic_max(unsigned char* lds, int off, int v) { atomicMax(reinterpret_cast<int*>(lds + off), v); }
GS_FN f4 glob_read16(const float* p) { return *reinterpret_cast<const f4*>(p); }
GS_FN void glob_write16(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// global_load_lds_dwordx4, saddr form: LDS destination = M0 (wave-uniform byte address) + lane * 16; source = SGPR base +
// 32-bit VGPR byte offset.  M0 is compiler-reserved: written and restored inside the statement (cdna_hip_programming.md,
// inline-asm rules).  hipcc does not count this load: the caller waits with s_waitcnt vmcnt(0) before the barrier.
// (s_mov, s_mov, s_nop 2 = the five wait states a vector-memory read of a scalar register needs behind a vector-ALU write of
// it -- sbase may have just been reloaded by v_readlane; the compiler does not pad asm statements)
GS_FN void dma16(unsigned char* lds, int lds_wave_base, int /*lane*/, const float* sbase, int voff) {
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)lds_wave_base;
  const unsigned dst_u = __builtin_amdgcn_readfirstlane(dst);
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(dst_u), "s"(sbase)
      : "memory");
}
#endif

template <int PAD, int MODE, bool IN_P4, bool OUT_P4, int THREADS, int VPT, int MAXI>
struct TileThread {
  static_assert(VPT >= 1 && VPT <= 4, "up to 4 bricks per tile");
  static_assert(IN_P4 || !OUT_P4, "packed-4 output needs packed-4 input");
  static constexpr int ELEM = IN_P4 ? 16 : 4;       // LDS bytes of one box voxel of one unit
  static constexpr int XPS = IN_P4 ? 1 : 4;          // box voxels (along x) per 16-byte slot
  static constexpr int SCRATCH0 = TILE_HDR_BYTES;
  static constexpr int DATA0 = TILE_HDR_BYTES + tile_scratch_bytes(IN_P4, OUT_P4, THREADS);
  static constexpr bool TRANSPOSED_OUT = IN_P4 && !OUT_P4;   // NCDHW output of packed-4 accumulators

  // ---- uniform over the block ----
  TileParams p;
  unsigned char* lds;
  int n, u_begin, u_end;
  int ox, oy, oz;                                    // tile origin in the output lattice
  int nvox, DHW;
  bool union_mode;
  int bx0, by0, bz0, bw, bh, bd, nslots, nu;         // box of the current pass; nu = units per stage
  // ---- per thread ----
  int tid;
  float w[VPT][8];
  int pk[VPT];                                       // (z0+1) << 22 | (y0+1) << 11 | (x0+1); -1: dead
  int vox[VPT];                                      // output voxel index; -1: outside the output lattice
  int ebase[VPT];                                    // box element index of the floor corner (current pass)
  int goff[MAXI];                                    // byte offset of this thread's slot i in a unit's volume; -1 zero; -2 none

  GS_MFN void init(const TileParams& prm, unsigned char* lds_, int block, int nblocks, int tid_) {
    p = prm; lds = lds_; tid = tid_;
    nvox = p.Do * p.Ho * p.Wo;
    DHW = p.D * p.H * p.W;
    const int ntiles = p.ntx * p.nty * p.ntz;
    const int L = xcd_remap(block, nblocks);          // order (group, sample, tile): XCD k walks a contiguous eighth
    const int per_group = p.N * ntiles;
    const int g = L / per_group;
    const int r = L - g * per_group;
    n = r / ntiles;
    const int tile = r - n * ntiles;
    const int tzi = tile / (p.ntx * p.nty);
    const int r2 = tile - tzi * (p.ntx * p.nty);
    const int tyi = r2 / p.ntx;
    const int txi = r2 - tyi * p.ntx;
    ox = txi << p.txs; oy = tyi << p.tys; oz = tzi << p.tzs;
    u_begin = g * p.upb;
    u_end = imin(p.units, u_begin + p.upb);
    if (tid < VPT * 6) lds_seti(lds, tid * 4, (tid % 6) < 3 ? INT_MAX : INT_MIN);
  }

  // ---- phase: taps (after a barrier behind init) ----
  GS_MFN void taps() {
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int v = j * THREADS + tid;
      const int rx = v & ((1 << p.txs) - 1);
      const int ry = (v >> p.txs) & ((1 << p.tys) - 1);
      const int rz = v >> (p.txs + p.tys);
      const int xo = ox + rx, yo = oy + ry, zo = oz + rz;
      const bool inlat = xo < p.Wo && yo < p.Ho && zo < p.Do;
      vox[j] = inlat ? (zo * p.Ho + yo) * p.Wo + xo : -1;
      pk[j] = -1;
#pragma unroll
      for (int k = 0; k < 8; ++k) w[j][k] = 0.0f;
      if (inlat) {
        float gx, gy, gz;
        load_coord_xyz<MODE>(p.grid, p.theta, p.lin_x, p.lin_y, p.lin_z, n, xo, yo, zo, vox[j], nvox, gx, gy, gz);
        int x0, y0, z0;
        float wt[8];
        corner_weights<PAD>(gx, gy, gz, p.D, p.H, p.W, x0, y0, z0, wt);
        // live <=> at least one corner inside the volume on every axis
        const bool live = x0 >= -1 && x0 < p.W && y0 >= -1 && y0 < p.H && z0 >= -1 && z0 < p.D;
        if (live) {
          pk[j] = ((z0 + 1) << 22) | ((y0 + 1) << 11) | (x0 + 1);
#pragma unroll
          for (int k = 0; k < 8; ++k) w[j][k] = wt[k];
          const int b = j * 24;
          lds_atomic_min(lds, b + 0, x0); lds_atomic_min(lds, b + 4, y0); lds_atomic_min(lds, b + 8, z0);
          lds_atomic_max(lds, b + 12, x0); lds_atomic_max(lds, b + 16, y0); lds_atomic_max(lds, b + 20, z0);
        }
      }
      GS_SCHED_BARRIER();
    }
  }

  // box of brick j (or the union) -> origin / extents / slots.  An empty box (no live voxel) has 0 slots.
  GS_MFN void set_box(int lox, int loy, int loz, int hix, int hiy, int hiz) {
    if (lox > hix) { bx0 = by0 = bz0 = 0; bw = XPS == 1 ? 2 : 4; bh = bd = 2; nslots = 0; return; }
    bx0 = XPS == 1 ? lox : floor4(lox);
    by0 = loy; bz0 = loz;
    bw = hix + 2 - bx0;
    if (XPS == 4) bw = (bw + 3) & ~3;
    bh = hiy + 2 - by0;
    bd = hiz + 2 - bz0;
    // slots of one unit; saturate (huge boxes of wild warps must not overflow the product)
    const long s = (long)(bw / XPS) * bh * bd;
    nslots = s > 0x3fffffffL ? 0x3fffffff : (int)s;
  }

  GS_MFN void load_box(int j, int (&b)[6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = GS_UNIFORM(lds_geti(lds, j * 24 + k * 4));
  }

  // ---- phase: number of passes (after the barrier behind taps); uniform ----
  GS_MFN int plan_passes() {
    int u[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      int b[6];
      load_box(j, b);
#pragma unroll
      for (int k = 0; k < 3; ++k) { u[k] = imin(u[k], b[k]); u[k + 3] = imax(u[k + 3], b[k + 3]); }
    }
    set_box(u[0], u[1], u[2], u[3], u[4], u[5]);
    union_mode = (VPT == 1) || nslots <= p.cap_slots;
    return union_mode ? 1 : VPT;
  }

  GS_MFN bool in_pass(int j, int ps) const { return union_mode || j == ps; }

  // ---- phase: plan of pass ps; returns false (uniform) when the pass's box does not fit the stage ----
  // Writes the zero slots (box positions outside the volume) of every unit region of the stage.
  GS_MFN bool plan(int ps) {
    if (!union_mode) {
      int b[6];
      load_box(ps, b);
      set_box(b[0], b[1], b[2], b[3], b[4], b[5]);
    }
    if (nslots > p.cap_slots) return false;
    const int nunits = u_end - u_begin;
    nu = nslots > 0 ? imin(nunits, p.cap_slots / nslots) : nunits;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      ebase[j] = 0;
      if (in_pass(j, ps) && pk[j] != -1) {
        const int x0 = (pk[j] & 0x7ff) - 1, y0 = ((pk[j] >> 11) & 0x7ff) - 1, z0 = ((pk[j] >> 22) & 0x3ff) - 1;
        ebase[j] = ((z0 - bz0) * bh + (y0 - by0)) * bw + (x0 - bx0);
      }
    }
    // slot s = i * THREADS + tid -> (xr, yr, zr) of the box, stepped without divisions
    const int bws = bw / XPS;
    const int dx = THREADS % bws, dt = THREADS / bws;
    const int dty = dt % bh, dtz = dt / bh;
    int xr = tid % bws;
    const int t0 = tid / bws;
    int yr = t0 % bh, zr = t0 / bh;
    bool any_zero = false;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int x = bx0 + xr * XPS, y = by0 + yr, z = bz0 + zr;
      const bool inside = (unsigned)x < (unsigned)p.W && (unsigned)y < (unsigned)p.H && (unsigned)z < (unsigned)p.D;
      const bool have = i * THREADS + tid < nslots;
      goff[i] = have ? (inside ? ((z * p.H + y) * p.W + x) * ELEM : -1) : -2;
      any_zero = any_zero || (have && !inside);
      xr += dx;
      int cy = dty;
      if (xr >= bws) { xr -= bws; cy += 1; }
      yr += cy;
      zr += dtz;
      if (yr >= bh) { yr -= bh; zr += 1; }
    }
    if (any_zero) {          // only tiles whose box sticks out of the volume
      for (int k = 0; k < nu; ++k) {
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
          if (goff[i] == -1) lds_zero16(lds, DATA0 + (k * nslots + i * THREADS + tid) * 16);
      }
    }
    return true;
  }

  GS_MFN const float* unit_base(int unit) const {
    return p.vol + (long)n * p.vol_bstride + (long)unit * DHW * (IN_P4 ? 4 : 1);
  }

  // ---- phase: fill the stage with units [u0, u0 + nu) (LDS-DMA; the caller waits vmcnt(0) and barriers) ----
  GS_MFN void fill(int u0) {
    const int nuc = imin(nu, u_end - u0);
    const int wave_slot0 = (tid >> 6) << 6;
    for (int k = 0; k < nuc; ++k) {
      const float* sb = unit_base(u0 + k);
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        if (i * THREADS < nslots) {                                   // uniform
          if (goff[i] >= 0) dma16(lds, DATA0 + (k * nslots + i * THREADS + wave_slot0) * 16, tid & 63, sb, goff[i]);
        }
      }
    }
  }

  // stores: uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset -- no per-lane 64-bit address arithmetic
  GS_MFN void store(int unit, int j, f4 acc) {
    if (vox[j] < 0) return;
    if (OUT_P4) {
      unsigned char* ub = reinterpret_cast<unsigned char*>(p.out + ((long)n * p.units + unit) * nvox * 4);
      glob_write16(reinterpret_cast<float*>(ub + (unsigned)vox[j] * 16u), acc);
    } else {
      unsigned char* ub = reinterpret_cast<unsigned char*>(p.out + ((long)n * p.C + 4 * unit) * nvox);
      const unsigned vo = (unsigned)vox[j] * 4u;
      const unsigned long plane = (unsigned long)nvox * 4u;
      *reinterpret_cast<float*>(ub + vo) = acc.x;
      *reinterpret_cast<float*>(ub + plane + vo) = acc.y;
      *reinterpret_cast<float*>(ub + 2 * plane + vo) = acc.z;
      *reinterpret_cast<float*>(ub + 3 * plane + vo) = acc.w;
    }
  }
  GS_MFN void store1(int unit, int j, float acc) {
    if (vox[j] < 0) return;
    unsigned char* ub = reinterpret_cast<unsigned char*>(p.out + ((long)n * p.C + unit) * nvox);
    *reinterpret_cast<float*>(ub + (unsigned)vox[j] * 4u) = acc;
  }

  // ---- phase: gather units [u0, u0 + nu) of pass ps out of the stage and store ----
  // Voxel-outer, unit-inner: only one voxel's weights and corner values are live inside the inner loop (the scheduling
  // barrier keeps hipcc from interleaving the voxels of a thread, which costs more registers than it hides latency:
  // the other waves of the CU cover the LDS latency).
  GS_MFN f4 gather_acc(int j, int a00) {
    const int sy = bw * ELEM, sz = bw * bh * ELEM;
    const int a01 = a00 + sy, a10 = a00 + sz, a11 = a10 + sy;
    f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
    if (IN_P4) {
      f4 v[8];
      v[0] = lds_read16(lds, a00); v[1] = lds_read16(lds, a00 + 16);
      v[2] = lds_read16(lds, a01); v[3] = lds_read16(lds, a01 + 16);
      v[4] = lds_read16(lds, a10); v[5] = lds_read16(lds, a10 + 16);
      v[6] = lds_read16(lds, a11); v[7] = lds_read16(lds, a11 + 16);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float wc = w[j][c];
        acc.x = gs_fadd(acc.x, gs_fmul(v[c].x, wc));
        acc.y = gs_fadd(acc.y, gs_fmul(v[c].y, wc));
        acc.z = gs_fadd(acc.z, gs_fmul(v[c].z, wc));
        acc.w = gs_fadd(acc.w, gs_fmul(v[c].w, wc));
      }
    } else {
      float v[8];
      v[0] = lds_read4(lds, a00); v[1] = lds_read4(lds, a00 + 4);
      v[2] = lds_read4(lds, a01); v[3] = lds_read4(lds, a01 + 4);
      v[4] = lds_read4(lds, a10); v[5] = lds_read4(lds, a10 + 4);
      v[6] = lds_read4(lds, a11); v[7] = lds_read4(lds, a11 + 4);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc.x = gs_fadd(acc.x, gs_fmul(v[c], w[j][c]));
    }
    if (pk[j] == -1) { acc.x = acc.y = acc.z = acc.w = 0.0f; }
    return acc;
  }

  // Is the packed-4 -> NCDHW output of this block written as 16-byte stores of 4 x-consecutive voxels of ONE channel?
  // The 4 lanes of an x-aligned quad exchange their (voxel, 4 channels) accumulators through a 16-byte LDS slot per
  // lane (same wave: no barrier): a dword store writes 32-byte runs of a tile row of 8, which the write path charges as
  // whole requests (measured: + 8 us per frame against 16-byte stores).  Needs whole quads inside the lattice.
  GS_MFN bool quad_stores() const { return TRANSPOSED_OUT && p.txs >= 2 && (p.Wo & 3) == 0; }

  GS_MFN void emit_a(f4 acc) {                       // stage 1 of a transposed store: park the accumulator
    lds_write16(lds, SCRATCH0 + tid * 16, acc);
  }
  GS_MFN void emit_b(int unit, int j) {              // stage 2: lane (quad g, channel e) stores voxels 4g..4g+3 of channel e
    if (vox[j] < 0) return;
    const int e = tid & 3;
    const int q0 = SCRATCH0 + (tid & ~3) * 16 + e * 4;
    f4 o;
    o.x = lds_read4(lds, q0); o.y = lds_read4(lds, q0 + 16); o.z = lds_read4(lds, q0 + 32); o.w = lds_read4(lds, q0 + 48);
    unsigned char* ub = reinterpret_cast<unsigned char*>(p.out + ((long)n * p.C + 4 * unit + e) * nvox);
    glob_write16(reinterpret_cast<float*>(ub + (unsigned)(vox[j] - e) * 4u), o);
  }
  GS_MFN void emit(int unit, int j, f4 acc) {
    if (!IN_P4) store1(unit, j, acc.x);
    else if (quad_stores()) { emit_a(acc); emit_b(unit, j); }
    else store(unit, j, acc);
  }

  GS_MFN void gather(int ps, int u0) {
    const int nuc = imin(nu, u_end - u0);
    const int ustride = nslots * 16;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      if (!in_pass(j, ps)) continue;
      int a00 = DATA0 + ebase[j] * ELEM;
      for (int k = 0; k < nuc; ++k, a00 += ustride) emit(u0 + k, j, gather_acc(j, a00));
      GS_SCHED_BARRIER();
    }
  }

  // ---- a pass whose box does not fit the stage: the same arithmetic straight from global memory ----
  // The cold path (wild warps): written for few live registers, not for speed -- one corner at a time.
  GS_MFN void direct(int ps) {
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      if (!in_pass(j, ps) || vox[j] < 0) continue;
      const bool dead = pk[j] == -1;
      const int x0 = (pk[j] & 0x7ff) - 1, y0 = ((pk[j] >> 11) & 0x7ff) - 1, z0 = ((pk[j] >> 22) & 0x3ff) - 1;
      for (int unit = u_begin; unit < u_end; ++unit) {
        const float* sb = unit_base(unit);
        f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int x = x0 + (c & 1), y = y0 + ((c >> 1) & 1), z = z0 + (c >> 2);
          const bool ok = !dead && x >= 0 && x < p.W && y >= 0 && y < p.H && z >= 0 && z < p.D;
          const int off = ok ? (z * p.H + y) * p.W + x : 0;
          const float wc = w[j][c];
          if (IN_P4) {
            const f4 v = glob_read16(sb + (long)off * 4);
            acc.x = gs_fadd(acc.x, gs_fmul(ok ? v.x : 0.0f, wc));
            acc.y = gs_fadd(acc.y, gs_fmul(ok ? v.y : 0.0f, wc));
            acc.z = gs_fadd(acc.z, gs_fmul(ok ? v.z : 0.0f, wc));
            acc.w = gs_fadd(acc.w, gs_fmul(ok ? v.w : 0.0f, wc));
          } else {
            const float v = sb[off];
            acc.x = gs_fadd(acc.x, gs_fmul(ok ? v : 0.0f, wc));
          }
          GS_SCHED_BARRIER();
        }
        if (dead) { acc.x = acc.y = acc.z = acc.w = 0.0f; }
        if (IN_P4) store(unit, j, acc);
        else store1(unit, j, acc.x);
      }
    }
  }
};

}  // namespace gs3d
