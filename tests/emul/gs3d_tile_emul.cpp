// TEST INFRASTRUCTURE ONLY -- runs the phases of the LDS-staged sampler (emoportraits_amd/csrc/gs3d_tile.h, the SAME source
// the HIP kernel is compiled from) thread by thread on the CPU, so that its index logic (tile / brick / box / slot
// decomposition, zero border, multi-unit stages, brick-by-brick and direct fallbacks) is checked bit for bit against the
// oracle without a GPU.  "LDS" is a byte array pre-filled with a NaN pattern: a gather that reads a slot nobody filled shows up.
// Build: g++ -O1 -ffp-contract=off -shared -fPIC (tests/test_sampler_emul.py).
#define GS3D_HOST_EMULATION 1
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../emoportraits_amd/csrc/gs3d_tile.h"

namespace {

struct Stats {
  long blocks, staged_passes, direct_passes, union_blocks, slots_filled, stages;
};

template <int PAD, int MODE, bool IN_P4, bool OUT_P4, int THREADS, int VPT>
int run(const gs3d::TileParams& p, Stats* st) {
  typedef gs3d::TileThread<PAD, MODE, IN_P4, OUT_P4, THREADS, VPT, 16> TT;
  const int nblocks = p.ngroups * p.N * p.ntx * p.nty * p.ntz;
  const size_t lds_bytes = gs3d::TILE_HDR_BYTES + gs3d::tile_scratch_bytes(IN_P4, OUT_P4, THREADS) + (size_t)p.cap_slots * 16;
  std::vector<unsigned char> lds(lds_bytes);
  std::vector<TT> th(THREADS);
  for (int b = 0; b < nblocks; ++b) {
    memset(lds.data(), 0xff, lds_bytes);
    for (int t = 0; t < THREADS; ++t) th[t].init(p, lds.data(), b, nblocks, t);
    for (int t = 0; t < THREADS; ++t) th[t].taps();
    const int npass = th[0].plan_passes();
    for (int t = 1; t < THREADS; ++t)
      if (th[t].plan_passes() != npass) return -10;
    st->blocks++;
    st->union_blocks += th[0].union_mode ? 1 : 0;
    for (int ps = 0; ps < npass; ++ps) {
      bool staged = th[0].plan(ps);
      for (int t = 1; t < THREADS; ++t)
        if (th[t].plan(ps) != staged) return -11;
      if (!staged) {
        st->direct_passes++;
        for (int t = 0; t < THREADS; ++t) th[t].direct(ps);
        continue;
      }
      st->staged_passes++;
      for (int u0 = th[0].u_begin; u0 < th[0].u_end; u0 += th[0].nu) {
        if (th[0].nu < 1) return -12;
        // a stage is reused: poison it so that stale data of the previous stage cannot satisfy a gather
        for (int t = 0; t < THREADS; ++t) th[t].fill(u0);
        st->stages++;
        st->slots_filled += (long)th[0].nslots * gs3d::imin(th[0].nu, th[0].u_end - u0);
        if (!th[0].quad_stores()) {
          for (int t = 0; t < THREADS; ++t) th[t].gather(ps, u0);
        } else {
          // the transposed store exchanges accumulators between the lanes of a wave: run its two stages in lock step
          const int nuc = gs3d::imin(th[0].nu, th[0].u_end - u0);
          for (int j = 0; j < VPT; ++j) {
            if (!th[0].in_pass(j, ps)) continue;
            for (int k = 0; k < nuc; ++k) {
              for (int t = 0; t < THREADS; ++t)
                th[t].emit_a(th[t].gather_acc(j, TT::DATA0 + th[t].ebase[j] * TT::ELEM + k * th[t].nslots * 16));
              for (int t = 0; t < THREADS; ++t) th[t].emit_b(u0 + k, j);
            }
          }
        }
      }
    }
  }
  return 0;
}

template <int PAD, int MODE, bool IN_P4, bool OUT_P4>
int run_tv(const gs3d::TileParams& p, int threads, int vpt, Stats* st) {
  if (threads == 256 && vpt == 1) return run<PAD, MODE, IN_P4, OUT_P4, 256, 1>(p, st);
  if (threads == 256 && vpt == 2) return run<PAD, MODE, IN_P4, OUT_P4, 256, 2>(p, st);
  if (threads == 512 && vpt == 1) return run<PAD, MODE, IN_P4, OUT_P4, 512, 1>(p, st);
  if (threads == 512 && vpt == 2) return run<PAD, MODE, IN_P4, OUT_P4, 512, 2>(p, st);
  return -2;
}

template <int PAD, int MODE>
int run_layout(const gs3d::TileParams& p, int in_p4, int out_p4, int threads, int vpt, Stats* st) {
  if (in_p4 && out_p4) return run_tv<PAD, MODE, true, true>(p, threads, vpt, st);
  if (in_p4 && !out_p4) return run_tv<PAD, MODE, true, false>(p, threads, vpt, st);
  if (!in_p4 && !out_p4) return run_tv<PAD, MODE, false, false>(p, threads, vpt, st);
  return -3;
}

template <int PAD>
int run_mode(const gs3d::TileParams& p, int mode, int in_p4, int out_p4, int threads, int vpt, Stats* st) {
  switch (mode) {
    case gs3d::MODE_GRID: return run_layout<PAD, gs3d::MODE_GRID>(p, in_p4, out_p4, threads, vpt, st);
    case gs3d::MODE_THETA: return run_layout<PAD, gs3d::MODE_THETA>(p, in_p4, out_p4, threads, vpt, st);
    case gs3d::MODE_DELTA: return run_layout<PAD, gs3d::MODE_DELTA>(p, in_p4, out_p4, threads, vpt, st);
  }
  return -4;
}

}  // namespace

extern "C" int emul_gs3d_tile(const float* vol, const float* grid, const float* theta, const float* lin_x, const float* lin_y,
                              const float* lin_z, float* out, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                              long vol_bstride, int pad, int mode, int in_p4, int out_p4, int threads, int vpt, int txs,
                              int tys, int tzs, int upb, int cap_slots, long* stats6) {
  if ((1 << (txs + tys + tzs)) != threads * vpt) return -1;
  gs3d::TileParams p;
  p.vol = vol; p.grid = grid; p.theta = theta; p.lin_x = lin_x; p.lin_y = lin_y; p.lin_z = lin_z; p.out = out;
  p.N = N; p.C = C; p.D = D; p.H = H; p.W = W; p.Do = Do; p.Ho = Ho; p.Wo = Wo;
  p.vol_bstride = vol_bstride;
  p.txs = txs; p.tys = tys; p.tzs = tzs;
  p.ntx = (Wo + (1 << txs) - 1) >> txs; p.nty = (Ho + (1 << tys) - 1) >> tys; p.ntz = (Do + (1 << tzs) - 1) >> tzs;
  p.units = in_p4 ? C / 4 : C;
  p.upb = upb;
  p.ngroups = (p.units + upb - 1) / upb;
  p.cap_slots = cap_slots;
  Stats st = {0, 0, 0, 0, 0, 0};
  int rc;
  switch (pad) {
    case 0: rc = run_mode<0>(p, mode, in_p4, out_p4, threads, vpt, &st); break;
    case 1: rc = run_mode<1>(p, mode, in_p4, out_p4, threads, vpt, &st); break;
    case 2: rc = run_mode<2>(p, mode, in_p4, out_p4, threads, vpt, &st); break;
    default: rc = -5;
  }
  if (stats6) {
    stats6[0] = st.blocks; stats6[1] = st.staged_passes; stats6[2] = st.direct_passes; stats6[3] = st.union_blocks;
    stats6[4] = st.slots_filled; stats6[5] = st.stages;
  }
  return rc;
}
