// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, bitwise a
// k-ordered fmaf chain, 64 FLOP/clk/SIMD = 157 TF/chip) -- SURVEY.md section 8 rows a5, a9, a10 (and a6-a8).
//
// Replaces the F.conv2d / F.conv3d calls inside ResBlock / ConvBlock of the reference
// (networks/volumetric_avatar/utils.py:661-788) together with the pointwise work around them:
//   * the GroupNorm-apply + ReLU that precedes every conv of a ResBlock (utils.py:711-731) is folded into the
//     conv's input staging as a per-(sample, channel) affine  x*scale + shift  followed by max(.,0)
//     (scale/shift come from emo_groupnorm_affine_f32); zero padding is applied AFTER that transform, exactly
//     as F.conv does on the normalised tensor;
//   * the nearest-neighbour x2 upsampling of the decoder's up-blocks (utils.py:684-688,764-781) is folded
//     into the input gather (source index = logical index >> 1);
//   * bias, the residual/skip addition (utils.py:783) and tanh/sigmoid heads are applied in the epilogue.
// Spectral norm / weight standardisation are folded into the weights once at load time.
//
// GEMM view: D[co][p] = sum_k A[co][k] * B[k][p],  k = (ci, kd, kh, kw), p = output position.
//   rows (MFMA "i") = output channels, columns (MFMA "j") = positions => NC(D)HW stores are coalesced.
//   One stage = KC input channels x one depth tap x all KHxKW taps.  The raw input patch
//   [KC][TZ][TR+KH-1][TW+KW-1] is staged in LDS once and the B operand is read straight from it (no im2col
//   expansion): lane (half=l>>5, j=l&31) reads patch[2*pair+half][.. + r][.. + s], i.e. the two k-values of an
//   MFMA step are two CHANNELS at the same tap, so every LDS address is lane_base + compile-time immediate.
//   Weights are pre-packed on the host as [co_tile][stage][pair][tap][half][BM] so a stage's A tile is one
//   contiguous block copied with 16-byte loads.
// Block = 256 threads = 4 waves; wave tile = (TM x 32) x (TP x 32); LDS double-buffered, one barrier/stage.
#pragma once
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));   // native vector: stays an SSA value (HIP's float4 struct kept the
                                                             // prefetched weight tile in a scratch alloca once it was loop-carried)

struct ConvArgs {
  const float* x;      // [N, Cin, D, H, W]  (source dims; logical dims are (D, 2H, 2W) when UPS)
  const float* wpk;    // packed weights
  const float* bias;   // [Cout] or null
  const float* scale;  // [N, Cin] or null: input transform x*scale + shift (GroupNorm folded)
  const float* shift;  // [N, Cin]
  const float* res;    // residual added in the epilogue (shape of out; pre-upsample shape if res_ups) or null
  float* out;          // [N, Cout, Dl, Hl, Wl]
  int N, Cin, Cout;
  int D, H, W;         // source dims
  int Dl, Hl, Wl;      // logical = output dims
  int KD;              // depth taps (1 or 3), padding KD/2
  int relu_in;         // relu after the input affine (also usable without scale)
  int act;             // EMO_ACT_*
  int res_ups;         // residual is read at (z, y>>1, x>>1) from a [N,Cout,Dl,Hl/2,Wl/2] tensor
  int n_cchunks;       // ceil(Cin / KC)
  int tiles_x, tiles_y, tiles_z;
  int n_cotiles;       // ceil(Cout / BM)
  int ksplit;          // >= 1: the (channel chunk x depth tap) stages are divided over ksplit blocks per output tile
  int stages_per_split;
  float* partial;      // ksplit > 1: raw partial sums [ksplit][N][Cout][Dl][Hl][Wl]; bias / residual / activation are
                       // applied by conv_splitk_epilogue_kernel (conv_api.hip), which adds the splits in fixed order
};

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
struct ConvCfg {
  static constexpr int BM = WGM * TM * 32;
  static constexpr int BP = WGP * TP * 32;
  static constexpr int TAPS = KH * KW;
  static constexpr int KLOC = KC * TAPS;
  static constexpr int PR = TR + KH - 1;
  static constexpr int PW = TW + KW - 1;
  static constexpr int CHS = TZ * PR * PW;          // patch floats per input channel
  static constexpr int PATCH = KC * CHS;
  static constexpr int ASZ = KLOC * BM;             // floats of one stage's weight tile
  static constexpr int BUF = ASZ + ((PATCH + 3) & ~3);
#ifndef EMO_CONV_PRODUCERS
#define EMO_CONV_PRODUCERS 0   /* 0: all 4 waves stage AND multiply; 2: two extra loader waves do all the staging
                                  (wave specialisation), the 4 MFMA waves only read LDS and multiply */
#endif
  static constexpr int NPW = EMO_CONV_PRODUCERS;    // loader ("producer") waves
  static constexpr int SW = NPW ? NPW : 4;          // waves that stage
  static constexpr int ST = SW * 64;                // threads that stage
  static constexpr int THREADS = 256 + 64 * NPW;
  static constexpr int WPC = KC >= SW ? 1 : SW / KC;           // staging waves that share one input channel's patch
  static constexpr int CPW = KC >= SW ? KC / SW : 1;           // channels staged per staging wave per stage
  static constexpr int EPC = (CHS + 64 * WPC - 1) / (64 * WPC); // patch elements per lane per channel
  static constexpr int NPE = CPW * EPC;             // patch elements per thread per stage
  static constexpr int NA4 = (ASZ / 4 + SW * 64 - 1) / (SW * 64); // float4 weight loads per staging thread
  static_assert(WGM * WGP == 4, "4 waves per block");
  static_assert(TZ * TR * TW == BP, "position tile must equal BP");
  static_assert(KC % 2 == 0 && (KC % SW == 0 || SW % KC == 0), "whole channels per wave, or whole waves per channel");
  static_assert(EMO_CONV_PRODUCERS == 0 || EMO_CONV_PRODUCERS == 2, "0 or 2 loader waves");
  static_assert(ASZ % 4 == 0, "weight tile must be float4-copyable");
  static_assert(TM * TP <= 4, "accumulator budget");
};

__device__ __forceinline__ float emo_act(float v, int act) {
  if (act == EMO_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == EMO_ACT_TANH) return tanhf(v);
  if (act == EMO_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
#ifndef EMO_CONV_MIN_WAVES
#define EMO_CONV_MIN_WAVES 2   /* __launch_bounds__ 2nd argument: minimum waves per SIMD the register allocation must allow */
#endif
#ifndef EMO_CONV_MAX_WAVES
#define EMO_CONV_MAX_WAVES 5   /* occupancy the register allocation is planned for: LDS holds 3-5 blocks of 4 waves per CU.  Measured
                                  on the 64-row config (64x64..256x256 layers): 4 -> 127.5, 5 -> 129.5, 6 -> 118 TF */
#endif
__global__ __launch_bounds__(256 + 64 * EMO_CONV_PRODUCERS) __attribute__((amdgpu_waves_per_eu(EMO_CONV_MIN_WAVES, EMO_CONV_MAX_WAVES)))
void conv_igemm_kernel(const ConvArgs a) {
  using Cfg = ConvCfg<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  constexpr int BM = Cfg::BM, TAPS = Cfg::TAPS, PR = Cfg::PR, PW = Cfg::PW, CHS = Cfg::CHS;
  constexpr int PATCH = Cfg::PATCH, ASZ = Cfg::ASZ, BUF = Cfg::BUF, NPE = Cfg::NPE, NA4 = Cfg::NA4;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave id as a scalar (SGPR): everything derived from it (staged channel, base pointers, GN scale/shift) is then
  // wave-uniform and handled by the scalar unit instead of costing VALU issue slots next to the MFMA stream
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  constexpr int NPW = Cfg::NPW, SW = Cfg::SW, ST = Cfg::ST;
  const bool is_producer = NPW > 0 && wave >= 4;          // wave-uniform role
  const bool is_consumer = wave < 4;
  const bool stages_data = NPW == 0 || is_producer;
  const int sw = NPW ? (wave >= 4 ? wave - 4 : 0) : wave; // index among the staging waves
  const int stid = sw * 64 + lane;                        // index among the staging threads
  constexpr int WPC = Cfg::WPC;
  const int chan0 = KC >= SW ? sw : sw % KC;           // channel (within a chunk) this wave stages; further ones at + g*SW
  const int part = KC >= SW ? 0 : sw / KC;               // which 64-element slices of that channel's patch (WPC waves share it)
  const int cwave = wave & 3;
  const int wm = cwave / WGP, wp = cwave % WGP;
  const int m0 = wm * TM * 32, p0 = wp * TP * 32;

#ifndef EMO_CONV_XCD_ORDER
#define EMO_CONV_XCD_ORDER 1   /* 1: 1-D grid, XCD-contiguous, output-channel tile fastest (see below); 0: (ptile, cotile, n) grid */
#endif
  int n, cotile, bx, ks = 0;
  if (EMO_CONV_XCD_ORDER) {
    // Block b runs on XCD b % 8 (private 4 MiB L2 each).  Re-map so that every XCD walks one contiguous eighth of the
    // (sample, position tile, output-channel tile) work with the channel tile fastest: all channel tiles of a position
    // tile then run back to back on ONE XCD and share the input patch out of its L2 (it was fetched from HBM /
    // Infinity Cache once per channel tile before: 2-5x read amplification), and x-adjacent position tiles share
    // their halo columns the same way.  Bijective for any grid size.
    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    cotile = L % a.n_cotiles;
    int rest = L / a.n_cotiles;
    if (a.ksplit > 1) {   // K splits of one tile sit next to each other: same patch rows, different channels
      ks = rest % a.ksplit;
      rest /= a.ksplit;
    }
    const int nptiles = a.tiles_x * a.tiles_y * a.tiles_z;
    n = rest / nptiles;
    bx = rest - n * nptiles;
  } else {
    n = blockIdx.z;
    cotile = blockIdx.y;
    bx = blockIdx.x;
  }
  const int tx = bx % a.tiles_x; bx /= a.tiles_x;
  const int ty = bx % a.tiles_y; bx /= a.tiles_y;
  const int tz = bx;
  const int x0 = tx * TW, y0 = ty * TR, z0 = tz * TZ;

  const int HW = a.H * a.W;
  const long DHW = (long)a.D * HW;
  const float* xn = a.x + (long)n * a.Cin * DHW;
  const bool has_affine = a.scale != nullptr;
  const bool relu_in = a.relu_in != 0;
  const int padD = a.KD >> 1;
  // Branch-free input transform: v = max(v * sc + sh, floor).  Without an affine the scalar loads still happen (from the
  // input itself, any readable address) and are replaced by (1, 0); without ReLU the floor is -inf.  No conditional code
  // in the K loop: a conditionally used load is sunk by the compiler next to its use, which serialises load -> wait ->
  // LDS write in the middle of the stage instead of prefetching a stage ahead.
  const float* scale_n = has_affine ? a.scale + (long)n * a.Cin : a.x;
  const float* shift_n = has_affine ? a.shift + (long)n * a.Cin : a.x;
  const float relu_floor = relu_in ? 0.0f : -__builtin_huge_valf();

  // ---- patch staging map: wave w stages input channels {w, w+4, ...} of the chunk; lane l element l + 64*i of the
  //      channel's [TZ][PR][PW] patch.  Per element only a plane offset and a validity bit are kept (constant over
  //      stages); the channel / depth part of the address is a scalar base per stage. ----
  constexpr int CPW = Cfg::CPW, EPC = Cfg::EPC;
  unsigned p_off[EPC]; // BYTE offset (ys*W + xs)*4 inside an input plane (0 when the element is outside the image);
                       // unsigned 32-bit so that loads take the scalar-base + vector-offset addressing form
  int p_pz[EPC];       // z within the tile (non-zero only for TZ > 1)
  bool p_ok[EPC];      // element exists and its (y, x) lies inside the logical image
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    const int e = lane + (i * WPC + part) * 64;
    const int pz = e / (PR * PW);
    const int rem2 = e - pz * (PR * PW);
    const int pr = rem2 / PW;
    const int pc = rem2 - pr * PW;
    const int yl = y0 + pr - (KH >> 1);
    const int xl = x0 + pc - (KW >> 1);
    const bool ok = (e < CHS) && ((unsigned)yl < (unsigned)a.Hl) && ((unsigned)xl < (unsigned)a.Wl);
    const int ys = UPS ? (yl >> 1) : yl;
    const int xs = UPS ? (xl >> 1) : xl;
    p_ok[i] = ok;
    p_off[i] = ok ? (unsigned)(ys * a.W + xs) * 4u : 0u;
    p_pz[i] = pz;
  }

  const int nstages_all = a.n_cchunks * a.KD;
  const int st_begin = ks * a.stages_per_split;                       // this block's share of the K loop
  const int st_end = min(nstages_all, st_begin + a.stages_per_split);
  const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk) + ((long)cotile * nstages_all) * (ASZ / 4);

  // per-lane dump slots behind the two stage buffers (written, never read)
  floatx4* const dump4 = reinterpret_cast<floatx4*>(smem + 2 * BUF) + lane;
  float* const dump1 = smem + 2 * BUF + lane;

  float pv[NPE];        // staged patch values (raw)
  bool pvz[NPE];        // per-element depth validity (only varies per element when TZ > 1)
  bool sv[CPW];         // wave-uniform: channel exists (and, for TZ == 1, the depth slice is inside the volume)
  float sc[CPW], sh[CPW];
  floatx4 av[NA4];
#ifndef EMO_CONV_GLDS_A
#define EMO_CONV_GLDS_A 1   /* 1: weight tile by LDS-DMA (global_load_lds), issued at the top of the stage straight into the idle
                               buffer: no VGPR round trip, and -- being a side-effecting builtin -- it stays where it is put;
                               0: through VGPRs + ds_write_b128 (the scheduler sinks those loads next to the ds_write) */
#endif
  constexpr int NGL = (ASZ * 4 + 1024 * SW - 1) / (1024 * SW);   // 1-KiB LDS-DMA pieces per staging wave

// Both staging halves are macros (not lambdas / conditionals) so that pv[] / av[] are unconditionally defined
// straight-line values and stay in VGPRs (a conditional or lambda-captured definition sent them to scratch).
#define EMO_ISSUE_LOADS(stage_, dst_) { EMO_ISSUE_PATCH(stage_); EMO_ISSUE_WEIGHTS(stage_, dst_); }

#define EMO_ISSUE_PATCH(stage_)                                                                       \
  {                                                                                                   \
    const int cc_ = (stage_) / a.KD;                                                                  \
    const int t_ = (stage_) - cc_ * a.KD;                                                             \
    const int ci0_ = cc_ * KC;                                                                        \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) {                                                 \
      const int c_ = ci0_ + g * SW + chan0;                                                           \
      const bool cv_ = c_ < a.Cin;                                                                    \
      const int cs_ = cv_ ? c_ : 0;                                                                   \
      const int zu_ = z0 + t_ - padD;           /* depth of tile slice 0 */                           \
      const bool zv_ = (unsigned)zu_ < (unsigned)a.D;                                                 \
      const float* base_ = xn + (long)cs_ * DHW + (long)((TZ == 1 && zv_) ? zu_ : 0) * HW;            \
      sv[g] = cv_ && (TZ > 1 || zv_);                                                                 \
      { const float s1_ = scale_n[cs_], s0_ = shift_n[cs_];                                           \
        sc[g] = has_affine ? s1_ : 1.0f; sh[g] = has_affine ? s0_ : 0.0f; }                           \
      _Pragma("unroll") for (int i = 0; i < EPC; ++i) {                                               \
        if (TZ == 1) {                                                                                \
          pvz[g * EPC + i] = true;                                                                    \
          pv[g * EPC + i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_) + p_off[i]); \
        } else {                                                                                      \
          const int zi = zu_ + p_pz[i];                                                               \
          const bool zok = (unsigned)zi < (unsigned)a.D;                                              \
          pvz[g * EPC + i] = zok;                                                                     \
          pv[g * EPC + i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_) + (p_off[i] + (unsigned)((zok ? zi : 0) * HW) * 4u)); \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
  }

#define EMO_ISSUE_WEIGHTS(stage_, dst_)                                                               \
  {                                                                                                   \
    const floatx4* ws_ = wsrc + (long)(stage_) * (ASZ / 4);                                           \
    if (EMO_CONV_GLDS_A) {                                                                            \
      /* weight tile: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), no VGPR round trip. */ \
      /* The LDS image is lane-linear = exactly the packed weight order.                               */ \
      _Pragma("unroll") for (int i = 0; i < NGL; ++i) {                                               \
        const int j = sw + SW * i;                                                                    \
        const int boff = j * 1024 + lane * 16;                                                        \
        if (boff < ASZ * 4)                                                                           \
          __builtin_amdgcn_global_load_lds(                                                           \
              (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(ws_) + boff), \
              (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(dst_) + j * 1024), 16, 0, 0); \
      }                                                                                               \
    } else {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < NA4; ++i) {                                               \
        const int idx = stid + i * ST;                                                                \
        av[i] = ws_[idx < ASZ / 4 ? idx : ASZ / 4 - 1];                                               \
      }                                                                                               \
    }                                                                                                 \
  }

/* Stores are unconditional: lanes beyond the tile write to a private dump slot (address select, no branch). */ \
#define EMO_STORE_STAGE(stage_, buf_)                                                                 \
  {                                                                                                   \
    if (!EMO_CONV_GLDS_A) {                                                                           \
      floatx4* As4_ = reinterpret_cast<floatx4*>(buf_);                                               \
      _Pragma("unroll") for (int i = 0; i < NA4; ++i) {                                               \
        const int idx = stid + i * ST;                                                                \
        floatx4* d4_ = ((i + 1) * ST <= ASZ / 4 || idx < ASZ / 4) ? As4_ + idx : dump4;               \
        *d4_ = av[i];                                                                                 \
      }                                                                                               \
    }                                                                                                 \
    float* Ps_ = (buf_) + ASZ;                                                                        \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) {                                                 \
      float* Pc_ = Ps_ + (g * SW + chan0) * CHS;                                                      \
      _Pragma("unroll") for (int i = 0; i < EPC; ++i) {                                               \
        const int e = lane + (i * WPC + part) * 64;                                                   \
        float v = fmaxf(__fmaf_rn(pv[g * EPC + i], sc[g], sh[g]), relu_floor);                        \
        /* zero padding applies to the transformed tensor */                                          \
        v = (p_ok[i] && sv[g] && pvz[g * EPC + i]) ? v : 0.0f;                                        \
        float* d_ = ((i + 1) * WPC * 64 <= CHS || e < CHS) ? Pc_ + e : dump1;                         \
        *d_ = v;                                                                                      \
      }                                                                                               \
    }                                                                                                 \
  }

#ifndef EMO_CONV_ABLATE
#define EMO_CONV_ABLATE 0   /* timing experiments only: 4 = global loads issued but never written to LDS, 5 = LDS writes of
                               stale registers without global loads, 1 = no global loads / LDS stores in the loop, 2 = also no barrier,
                               3 = MFMA stream only (operands read once) -- results are WRONG for any value != 0 */
#endif
#ifndef EMO_CONV_SCHED_FENCE
#define EMO_CONV_SCHED_FENCE 0   /* 1: __builtin_amdgcn_sched_barrier around the prefetch (measured: makes the backend spill the
                                    prefetched weight tile to scratch); 0: scheduler's choice */
#endif
#ifndef EMO_CONV_PIPE_W
#define EMO_CONV_PIPE_W 0   /* with EMO_CONV_PIPE and register-staged weights (EMO_CONV_GLDS_A == 0): 1 = the weight tile is loop-carried
                               in VGPRs like the patch; 0 = loaded at the top of the stage */
#endif
#ifndef EMO_CONV_PIPE
#define EMO_CONV_PIPE 1   /* where the global loads of stage s+1 are issued: 0 = at the top of stage s (the scheduler then sinks
                             them down to their first use at STORE_PAIR: ~0 prefetch distance, the memory latency is exposed once
                             per stage); 1 = at the end of stage s-1, BEFORE that stage's closing barrier -- loads cannot be moved
                             across the barrier's fences, so they are in flight for at least the first half of stage s */
#endif
#ifndef EMO_CONV_STORE_AT
#define EMO_CONV_STORE_AT 1   /* 0: write the next stage into LDS after all MFMAs of this stage; 1: after half of them */
#endif
  constexpr int STORE_PAIR = EMO_CONV_STORE_AT ? (KC / 4) : -1;   // the idle LDS buffer is free for the whole stage

  if (stages_data) {
    EMO_ISSUE_LOADS(st_begin, smem);
    EMO_STORE_STAGE(st_begin, smem);
    if (EMO_CONV_PIPE && NPW == 0) {   // second stage in flight across the barrier (registers are loop-carried)
      const int st1_ = (st_begin + 1) < st_end ? (st_begin + 1) : st_begin;
      EMO_ISSUE_PATCH(st1_);
      if (EMO_CONV_PIPE_W && !EMO_CONV_GLDS_A) { EMO_ISSUE_WEIGHTS(st1_, smem); }
    }
  }
  __syncthreads();

  if (is_producer) {
    // ---- loader waves (wave specialisation): the whole stage time to fetch, transform and park the next stage.
    //      Same number of barriers as the MFMA waves; no accumulators live on this path. ----
    for (int st = st_begin; st < st_end; ++st) {
      float* nxt = smem + ((st - st_begin + 1) & 1) * BUF;
      const int stn = (st + 1) < st_end ? (st + 1) : st;
      EMO_ISSUE_LOADS(stn, nxt);
      EMO_STORE_STAGE(stn, nxt);
      __syncthreads();
    }
    return;
  }

  floatx16 acc[TM][TP];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // lane bases into the LDS tiles
  const int a_base = half * BM + m0 + l32;
  int b_base[TP];
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW;
    const int row = (p / TW) % TR;
    const int pz = p / (TW * TR);
    b_base[j] = half * CHS + pz * (PR * PW) + row * PW + col;
  }

// all MFMAs of one channel pair of the current stage: per tap 1 A read + 1 B read per 32x32 tile, TM*TP MFMAs
#define EMO_MFMA_PAIR(pair_)                                                                          \
  {                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < KH; ++r) {                                                  \
      _Pragma("unroll") for (int s = 0; s < KW; ++s) {                                                \
        const int tap = r * KW + s;                                                                   \
        float av_[TM], bv_[TP];                                                                       \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
          av_[i] = As[a_base + (EMO_CONV_ABLATE == 3 ? 0 : (((pair_) * TAPS + tap) * 2) * BM) + i * 32]; \
        _Pragma("unroll") for (int j = 0; j < TP; ++j)                                                \
          bv_[j] = Ps[b_base[j] + (EMO_CONV_ABLATE == 3 ? 0 : ((pair_) * 2) * CHS + r * PW + s)];     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
          _Pragma("unroll") for (int j = 0; j < TP; ++j)                                              \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[i], bv_[j], acc[i][j], 0, 0, 0);     \
      }                                                                                               \
    }                                                                                                 \
  }

  for (int st = st_begin; st < st_end; ++st) {
    float* cur = smem + ((st - st_begin) & 1) * BUF;
    float* nxt = smem + ((st - st_begin + 1) & 1) * BUF;
    // prefetch the next stage while this one computes; on the last stage the (clamped) prefetch re-reads the
    // last stage and its LDS write lands in the idle buffer -- harmless, and it keeps the loop branch-free
    const int stn = (st + 1) < st_end ? (st + 1) : st;
    const float* As = cur;
    const float* Ps = cur + ASZ;
    if (NPW == 0) {
      // every wave stages and multiplies: loads first, LDS write of the next stage half way through the MFMAs
      if (EMO_CONV_ABLATE == 0 || EMO_CONV_ABLATE == 4) {
        if (!EMO_CONV_PIPE) { EMO_ISSUE_PATCH(stn); }
        if (!(EMO_CONV_PIPE && EMO_CONV_PIPE_W && !EMO_CONV_GLDS_A)) {
          EMO_ISSUE_WEIGHTS(stn, nxt);   // LDS-DMA straight into the idle buffer (or through VGPRs when EMO_CONV_GLDS_A == 0)
        }
      }
      // pin the software pipeline: the next stage's global loads are ISSUED here, before this stage's MFMAs, and are
      // first touched (s_waitcnt + transform + LDS write) at STORE_PAIR -- without the fences the scheduler moves the
      // loads down next to their use and the whole memory latency is exposed once per stage
      if (EMO_CONV_SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
      if (EMO_CONV_ABLATE == 4) {   /* keep the loads alive without touching LDS */
        _Pragma("unroll") for (int q = 0; q < NPE; ++q) asm volatile("" ::"v"(pv[q]));
        _Pragma("unroll") for (int q = 0; q < NA4; ++q) asm volatile("" ::"v"(av[q].x), "v"(av[q].w));
      }
#pragma unroll
      for (int pair = 0; pair < KC / 2; ++pair) {
        if (pair == STORE_PAIR && STORE_PAIR > 0 && (EMO_CONV_ABLATE == 0 || EMO_CONV_ABLATE == 5)) {
          if (EMO_CONV_SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
          EMO_STORE_STAGE(stn, nxt);
        }
        EMO_MFMA_PAIR(pair);
      }
      if (STORE_PAIR <= 0 && (EMO_CONV_ABLATE == 0 || EMO_CONV_ABLATE == 5)) {
        if (EMO_CONV_SCHED_FENCE) __builtin_amdgcn_sched_barrier(0);
        EMO_STORE_STAGE(stn, nxt);
      }
      if (EMO_CONV_PIPE && (EMO_CONV_ABLATE == 0 || EMO_CONV_ABLATE == 4)) {
        // the registers were consumed by EMO_STORE_STAGE above: refill them with stage st+2, before the barrier
        const int stn2 = (st + 2) < st_end ? (st + 2) : (st_end - 1);
        EMO_ISSUE_PATCH(stn2);
        if (EMO_CONV_PIPE_W && !EMO_CONV_GLDS_A) { EMO_ISSUE_WEIGHTS(stn2, nxt); }
      }
    } else {
      // MFMA waves of the wave-specialised variant: nothing but LDS reads and matrix instructions
#pragma unroll
      for (int pair = 0; pair < KC / 2; ++pair) { EMO_MFMA_PAIR(pair); }
    }
    if (EMO_CONV_ABLATE < 2 || EMO_CONV_ABLATE >= 4) __syncthreads();
  }
#undef EMO_MFMA_PAIR

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
  const long plane = (long)a.Hl * a.Wl;
  const long ovol = (long)a.Dl * plane;
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int p = p0 + j * 32 + l32;
    const int col = p % TW;
    const int row = (p / TW) % TR;
    const int pz = p / (TW * TR);
    const int z = z0 + pz, y = y0 + row, x = x0 + col;
    const long sp = (long)z * plane + (long)y * a.Wl + x;
    long rsp = sp;
    long rvol = ovol;
    if (a.res_ups) {
      const int Wr = a.Wl >> 1, Hr = a.Hl >> 1;
      rsp = ((long)z * Hr + (y >> 1)) * Wr + (x >> 1);
      rvol = (long)a.Dl * Hr * Wr;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cotile * BM + m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.Cout && a.partial) {
          a.partial[(((long)ks * a.N + n) * a.Cout + co) * ovol + sp] = acc[i][j][r];
        } else if (co < a.Cout) {
          float v = acc[i][j][r];
          if (a.bias) v += a.bias[co];
          if (a.res) v += a.res[((long)n * a.Cout + co) * rvol + rsp];
          v = emo_act(v, a.act);
          a.out[((long)n * a.Cout + co) * ovol + sp] = v;
        }
      }
    }
  }
}

#undef EMO_ISSUE_LOADS
#undef EMO_ISSUE_PATCH
#undef EMO_ISSUE_WEIGHTS
#undef EMO_STORE_STAGE

// host-side launcher for one instantiation
template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS>
int conv_igemm_launch(ConvArgs a, hipStream_t s) {
  using Cfg = ConvCfg<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (a.Wl % TW || a.Hl % TR || a.Dl % TZ) return EMO_ERR_UNSUPPORTED;
  a.tiles_x = a.Wl / TW;
  a.tiles_y = a.Hl / TR;
  a.tiles_z = a.Dl / TZ;
  a.n_cchunks = (a.Cin + KC - 1) / KC;
  const long nt = (long)a.tiles_x * a.tiles_y * a.tiles_z;
  if (nt > 0x7fffffffL || a.N > 65535) return EMO_ERR_UNSUPPORTED;
  const int cot = (a.Cout + Cfg::BM - 1) / Cfg::BM;
  const size_t lds = (size_t)(2 * Cfg::BUF + 256) * sizeof(float);   // two stage buffers + 64 float4 dump slots
  if (lds > 160 * 1024) return EMO_ERR_UNSUPPORTED;
  auto kern = conv_igemm_kernel<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  if (lds > 64 * 1024) {
    // opt in to more than 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU); idempotent, so a benign race
    static bool raised = false;
    if (!raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      raised = true;
    }
  }
  a.n_cotiles = cot;
  if (a.ksplit < 1 || (a.ksplit > 1 && (!EMO_CONV_XCD_ORDER || !a.partial))) return EMO_ERR_BAD_ARG;
  if (a.ksplit == 1) { a.stages_per_split = a.n_cchunks * a.KD; a.partial = nullptr; }
  if (nt * cot * a.N * a.ksplit > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  dim3 g = EMO_CONV_XCD_ORDER ? dim3((unsigned)(nt * cot * a.N * a.ksplit)) : dim3((unsigned)nt, cot, a.N);
  hipLaunchKernelGGL(kern, g, dim3(Cfg::THREADS), lds, s, a);
  return emo_launch_status();
}
