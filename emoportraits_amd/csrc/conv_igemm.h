// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, bitwise a
// k-ordered fmaf chain, 64 FLOP/clk/SIMD = 157 TF/chip) -- SURVEY.md section 8 rows a5, a9, a10 (and a6-a8).
//
// Replaces the F.conv2d / F.conv3d calls inside ResBlock / ConvBlock of the reference
// (networks/volumetric_avatar/utils.py:661-788) together with the pointwise work around them:
//   * the GroupNorm-apply + ReLU that precedes every conv of a ResBlock (utils.py:711-731) is folded into the
//     conv's input staging as a per-(sample, channel) affine  x*scale + shift  followed by max(.,0)
//     (scale/shift come from emo_groupnorm_affine_f32 / emo_groupnorm_affine_from_tiles_f32); zero padding is
//     applied AFTER that transform, exactly as F.conv does on the normalised tensor;
//   * the nearest-neighbour x2 upsampling of the decoder's up-blocks (utils.py:684-688,764-781) is folded
//     into the input gather (source index = logical index >> 1);
//   * bias, the residual/skip addition (utils.py:783) and tanh/sigmoid heads are applied in the epilogue;
//   * the statistics of the NEXT GroupNorm (utils.py:711-731 of the following block) are reduced from the
//     accumulators in the epilogue: per (sample, position tile, channel) the mean and the centred sum of squares of
//     the tile's 128 output values (gn_stats; combined exactly by emo_groupnorm_affine_from_tiles_f32), so the
//     output tensor is never re-read for its normalisation.
// Spectral norm / weight standardisation are folded into the weights once at load time.
//
// GEMM view: D[co][p] = sum_k A[co][k] * B[k][p],  k = (ci, kd, kh, kw), p = output position.
//   rows (MFMA "i") = output channels, columns (MFMA "j") = positions => NC(D)HW stores are coalesced.
//   One stage = KC input channels x one depth tap x all KHxKW taps.  The raw input patch
//   [KC][TZ][TR+KH-1][TW+KW-1] is staged in LDS once and the B operand is read straight from it (no im2col
//   expansion): lane (half=l>>5, j=l&31) reads patch[2*pair+half][.. + r][.. + s], i.e. the two k-values of an
//   MFMA step are two CHANNELS at the same tap, so every LDS address is lane_base + compile-time immediate.
//   Weights are pre-packed on the host as [co_tile][stage][pair][tap][half][BM] so a stage's A tile is one
//   contiguous block copied by LDS-DMA.
// Block = 256 threads = 4 waves; wave tile = (TM x 32) x (TP x 32); LDS double-buffered, one barrier/stage.
//
// Software pipeline of one stage s (EMO_CONV_PIPE == 2, the default):
//     top      LDS-DMA of the weight tile of stage s+1 into the idle buffer; global loads of the patch of stage s+1
//              into registers (inline asm: see below)
//     ...      MFMAs of stage s, operands from the current buffer
//     5/8      s_waitcnt for the patch registers, input transform, ds_write into the idle buffer
//     ...      remaining MFMAs
//     end      barrier
// Why the patch loads are inline asm: with an LDS-DMA in flight hipcc (ROCm 7.2) waits vmcnt(0) at the first use of
// any ordinary global load and before __syncthreads(), and its scheduler sinks ordinary loads down to their first
// use.  The round-1 kernel issued the loads of stage s+2 just before the closing barrier of stage s to keep them
// from sinking -- and the compiler drained them right there (s_waitcnt vmcnt(0); s_barrier): the full memory latency
// was exposed once per stage in every wave (ablation: 125 TF with the loads, 143 TF without).  Loads written as asm
// volatile stay where they are put, are invisible to the compiler's counters, and are waited for by hand.
#pragma once
#include "common.h"

#include <mutex>
#include <set>
#include <utility>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#ifndef EMO_CONV_MAX_WAVES
#define EMO_CONV_MAX_WAVES 5   /* blocks per CU (= waves per SIMD) the register allocation is planned for when LDS would hold more.
                                  Measured on the 64-row config (64x64..256x256 layers): 4 -> 127.5, 5 -> 129.5, 6 -> 118 TF */
#endif

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
  int cot0;            // conv_igemm_bf16x3.h / conv_igemm_f16x2_ct2.h / conv_igemm_f16.h: first channel tile of the launch (n_cotiles
                       // tiles from there: a layer's tile pairs run a two-tile kernel, an odd last tile a single-tile one); 0 elsewhere
  int cot_end;         // conv_igemm_f16x2_w8.h with plain fp16 operands: one past the last channel tile of the launch (even), or 0 =
                       // all of them, an odd last one in a half-empty pair (emo_conv_igemm_f16w8_rest, conv_api.hip)
  float in_scale, out_scale;   // conv_igemm_bf16x3.h, fp16 two-term split only: power of two applied to the staged input, and
                       // 1 / (in_scale * weight scale) applied to the accumulators before the epilogue
  int n_work;          // conv_igemm_bf16x3.h only: output tiles x K splits of the launch (a block walks the tiles of its
                       // XCD with a stride; gridDim.x == n_work unless EMO_CONV_BF16X3_PERSISTENT=1)
  int ksplit;          // >= 1: the (channel chunk x depth tap) stages are divided over ksplit blocks per output tile
  int stages_per_split;
  float* partial;      // ksplit > 1: raw partial sums [ksplit][N][Cout][Dl][Hl][Wl]; bias / residual / activation are
                       // applied by conv_splitk_epilogue_kernel (conv_api.hip), which adds the splits in fixed order
  float* gn_stats;     // or null (needs ksplit == 1): [N][position tiles][Cout][2] = (mean, centred sum of squares) of
                       // the 128 final output values of every (sample, position tile, channel)
  int* sat_flag;       // conv_igemm_bf16x3.h, fp16 two-term split only, or null: set to 1 when a staged value left the fp16 range
  const int* run_if;   // conv_igemm_bf16x3.h / conv_igemm.h, or null: the launch does nothing unless *run_if != 0 (guarded fallback of a layer
                       // whose fp16-split launch raised its sat_flag)
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
  static constexpr int SW = 4;                      // waves that stage (all of them)
  static constexpr int THREADS = 256;
  static constexpr int WPC = KC >= SW ? 1 : SW / KC;           // staging waves that share one input channel's patch
  static constexpr int CPW = KC >= SW ? KC / SW : 1;           // channels staged per staging wave per stage
  static constexpr int EPC = (CHS + 64 * WPC - 1) / (64 * WPC); // patch elements per lane per channel
  static constexpr int NPE = CPW * EPC;             // patch elements per thread per stage
  static constexpr int NSTEPS = (KC / 2) * TAPS;    // MFMA steps (one A/B operand fetch each) per stage
  static_assert(WGM * WGP == 4, "4 waves per block");
  static_assert(TZ * TR * TW == BP, "position tile must equal BP");
  static_assert(KC % 2 == 0 && (KC % SW == 0 || SW % KC == 0), "whole channels per wave, or whole waves per channel");
  static_assert(ASZ % 4 == 0, "weight tile must be float4-copyable");
  static_assert(TM * TP <= 8, "accumulator budget");
  static_assert(2 * BUF >= 2 * WGP * BM, "the GroupNorm tile statistics are exchanged through the stage buffers");
  static constexpr int LDS_BYTES = (2 * BUF + 256) * 4;             // two stage buffers + 64 float4 dump slots
  // blocks per CU that LDS admits, capped: the register allocator must fit that many waves per SIMD (1 wave per block
  // and SIMD) -- without a floor it spends up to 256 VGPRs on epilogue ILP and silently halves the occupancy
  static constexpr int BY_LDS = (160 * 1024) / LDS_BYTES;
  static constexpr int CAP = TM * TP >= 8 ? 2                                                    // 128 accumulator registers
                             : TM * TP >= 4 ? (EMO_CONV_MAX_WAVES < 4 ? EMO_CONV_MAX_WAVES : 4)   // 64: 4 waves per SIMD = 128 VGPRs
                                            : EMO_CONV_MAX_WAVES;
  static constexpr int OCC = BY_LDS < 2 ? 2 : (BY_LDS > CAP ? CAP : BY_LDS);
};

__device__ __forceinline__ float emo_act(float v, int act) {
  if (act == EMO_ACT_RELU) return fmaxf(v, 0.0f);
  if (act == EMO_ACT_TANH) return tanhf(v);
  if (act == EMO_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
  return v;
}

// A vector-memory instruction must not read a scalar register within five wait states of a VECTOR-ALU write of it (CDNA3 ISA,
// "manually inserted wait states": v_readlane / v_readfirstlane / v_cmp -> VMEM).  The compiler pads its own instructions; it
// cannot see into an asm statement, and it reloads spilled scalars with v_readlane wherever it likes -- also right in front of
// one (found in A/B builds of the split convolution that computed garbage or faulted: `v_readlane_b32 s85, ...` directly before
// `buffer_load_dwordx4 ..., s85 offen`; tools/kernel_resources.py --audit now checks every listing for it).  Every asm
// statement that names a scalar operand in a vector-memory instruction therefore starts with the five wait states itself.
#ifndef EMO_SGPR_HAZARD_NOP
#define EMO_SGPR_HAZARD_NOP "s_nop 4\n\t"     /* (-DEMO_SGPR_HAZARD_NOP='""': the audit's self-test) */
#endif


// global_load_dword with a scalar base and a 32-bit per-lane byte offset, hidden from the compiler's vmcnt
// bookkeeping and from its scheduler (stays where it is written; cdna_hip_programming.md section 5.7).  The destination
// is valid only after emo_wait_vmem0() + emo_touch().
__device__ __forceinline__ float emo_gload_pinned(const float* sbase, unsigned voff) {
  float v;
  asm volatile(EMO_SGPR_HAZARD_NOP "global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
// 16-byte form (e.g. four consecutive per-channel scale values through a wave-uniform address: every lane gets the same 4)
__device__ __forceinline__ floatx4 emo_gload4_pinned(const float* sbase, unsigned voff) {
  floatx4 v;
  asm volatile(EMO_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
__device__ __forceinline__ void emo_touch4(floatx4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void emo_wait_vmem0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// makes v opaque at this point: no consumer of v is scheduled above it (volatile asm statements keep their order, so
// after emo_wait_vmem0() this pins every use behind the wait)
__device__ __forceinline__ void emo_touch(float& v) { asm volatile("" : "+v"(v)); }

#define acc_at(i_, j_) ((j_) < TPH ? acc_lo[i_][(j_) < TPH ? (j_) : 0] : acc_hi[i_][(j_) >= TPH ? (j_) - TPH : 0])

// Epilogue shared by the fp32 and the fp16-operand kernels (both accumulate in fp32 with the same C/D layout).
// The accumulators live in two arrays of at most 64 floats each (see the kernel).
//
// The kernels issue their MFMAs with the operands swapped -- mfma(patch fragment, weight fragment) -- so a 32x32 result tile
// is [position][channel]: col = lane & 31 is the output CHANNEL, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) the
// POSITION.  Registers 4q .. 4q+3 of a lane are then 4 consecutive x positions of one channel: one 16-byte store (and one
// 16-byte residual load) instead of four 4-byte ones, one bias load per lane, and GroupNorm sums that run over a lane's own
// registers.  Global memory instructions are bound by their NUMBER on this chip (one wave-instruction per ~17 cycles and
// CU whatever its width: tools/microbench/vmem_rate.hip), and the epilogue of a 64 x 256 tile was 64 store instructions per
// wave; tools/fit_conv_overhead.py prices the per-block fixed cost at 7 % (fp32) to 30 % (fp16 operands) of a 128-channel
// layer at 512^2.
#ifndef EMO_CONV_NT_STORE
#define EMO_CONV_NT_STORE 0   /* 1: non-temporal output stores (A/B measurement) */
#endif
__device__ __forceinline__ void emo_store4(float* p, const floatx4& v, bool aligned) {
  if (aligned) {
    if (EMO_CONV_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<floatx4*>(p));
    else *reinterpret_cast<floatx4*>(p) = v;
  } else {
    p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
  }
}

template <int TZ, int TR, int TW, int TM, int TP, int WGP, int BM>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, floatx16 (&acc_lo)[TM][TP > 2 ? TP / 2 : TP],
                                              floatx16 (&acc_hi)[TM][TP > 2 ? TP / 2 : TP], float* smem, int n, int cotile,
                                              int ptile, int ks, int x0, int y0, int z0, int m0, int p0, int wp, int half,
                                              int l32, int tid) {
  constexpr int TPH = TP > 2 ? TP / 2 : TP;
  static_assert(TW % 4 == 0, "4 consecutive positions of a register quad lie in one tile row");
  const long plane = (long)a.Hl * a.Wl;
  const long ovol = (long)a.Dl * plane;
  const bool to_partial = a.partial != nullptr;
  const bool has_bias = a.bias != nullptr && !to_partial;
  const bool has_res = a.res != nullptr && !to_partial;
  // 16-byte accesses need 16-byte aligned tensors (every tensor the host code allocates is; a caller's view may not be)
  const bool out_al = ((reinterpret_cast<unsigned long long>(to_partial ? a.partial : a.out) & 15ull) == 0);
  const bool res_al = has_res && ((reinterpret_cast<unsigned long long>(a.res) & 15ull) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int co = cotile * BM + m0 + i * 32 + l32;     // this lane's output channel
    const bool co_ok = co < a.Cout;
    const int cs = co_ok ? co : a.Cout - 1;
    const float bv = has_bias ? a.bias[cs] : 0.0f;
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      // the four register quads of the tile: positions p4 .. p4 + 3
      long sp[4];
      floatx4 rv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p4 = p0 + j * 32 + 8 * q + 4 * half;
        const int col = p4 % TW;
        const int row = (p4 / TW) % TR;
        const int pz = p4 / (TW * TR);
        const int z = z0 + pz, y = y0 + row, x = x0 + col;
        sp[q] = (long)z * plane + (long)y * a.Wl + x;
        rv[q] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        if (has_res) {   // wave-uniform: all loads of the tile are issued before the first use
          if (a.res_ups) {
            const int Wr = a.Wl >> 1, Hr = a.Hl >> 1;
            const float* rp = a.res + ((long)n * a.Cout + cs) * ((long)a.Dl * Hr * Wr) + ((long)z * Hr + (y >> 1)) * Wr + (x >> 1);
            const float r0 = rp[0], r1 = rp[1];
            rv[q] = floatx4{r0, r0, r1, r1};
          } else {
            const float* rp = a.res + ((long)n * a.Cout + cs) * ovol + sp[q];
            if (res_al) rv[q] = *reinterpret_cast<const floatx4*>(rp);
            else rv[q] = floatx4{rp[0], rp[1], rp[2], rp[3]};
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        floatx4 v;
        if (to_partial) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc_at(i, j)[4 * q + e];
          if (co_ok) emo_store4(a.partial + (((long)ks * a.N + n) * a.Cout + co) * ovol + sp[q], v, out_al);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = emo_act(acc_at(i, j)[4 * q + e] + bv + rv[q][e], a.act);
            acc_at(i, j)[4 * q + e] = v[e];   // the stored value: what the next GroupNorm normalises
          }
          if (co_ok) emo_store4(a.out + ((long)n * a.Cout + co) * ovol + sp[q], v, out_al);
        }
      }
    }
  }

  // ---- GroupNorm statistics of the output tile (wave-uniform branch).  A lane holds 16 * TP values of ITS channel, its
  //      partner lane ^ 32 the other half of the wave's TP * 32 positions: mean = (own sum + partner's) / count, then the sum
  //      of squares centred at that mean -- a two-pass variance on values that are still in registers.  The WGP waves that
  //      share the channel combine their (mean, M2) through LDS with the pairwise update of Chan et al. (equal counts).
  //      No fp32 E[x^2] - mean^2 inside a tile; the cross-tile combine (gn_from_tiles_kernel) is done in fp64. ----
  if (a.gn_stats != nullptr && !to_partial) {
    float* st_lds = smem;   // [WGP][BM][2]: the stage buffers are idle (the K loop ended with a barrier)
    constexpr float inv_cnt = 1.0f / (float)(TP * 32);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc_at(i, j)[r];
      s += __shfl_xor(s, 32);
      const float mean = s * inv_cnt;
      float m2 = 0.0f;
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc_at(i, j)[r] - mean; m2 = __fmaf_rn(d, d, m2); }
      m2 += __shfl_xor(m2, 32);
      if (half == 0) *reinterpret_cast<float2*>(st_lds + (wp * BM + m0 + i * 32 + l32) * 2) = make_float2(mean, m2);
    }
    __syncthreads();
    if (tid < BM) {
      const int co = cotile * BM + tid;
      if (co < a.Cout) {
        float mean = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int w = 0; w < WGP; ++w) mean += st_lds[(w * BM + tid) * 2 + 0];
        mean *= 1.0f / (float)WGP;
#pragma unroll
        for (int w = 0; w < WGP; ++w) {
          const float d = st_lds[(w * BM + tid) * 2 + 0] - mean;
          m2 += st_lds[(w * BM + tid) * 2 + 1] + (float)(TP * 32) * d * d;
        }
        const long nptiles = (long)a.tiles_x * a.tiles_y * a.tiles_z;
        float2* dst = reinterpret_cast<float2*>(a.gn_stats) + ((long)n * nptiles + ptile) * a.Cout + co;
        *dst = make_float2(mean, m2);
      }
    }
  }
}
#undef acc_at

#ifndef EMO_CONV_XCD_ORDER
#define EMO_CONV_XCD_ORDER 1   /* 1: 1-D grid, XCD-contiguous, output-channel tile fastest (see below); 0: (ptile, cotile, n) grid */
#endif
#ifndef EMO_CONV_PIPE
#define EMO_CONV_PIPE 2   /* 2: patch loads of stage s+1 as pinned asm at the top of stage s (header comment);
                             1: round-1 schedule, kept for A/B measurements: ordinary loads of stage s+2 issued before the
                                closing barrier of stage s (drained there by the compiler's vmcnt(0)) */
#endif
#ifndef EMO_CONV_STORE_EIGHTHS
#define EMO_CONV_STORE_EIGHTHS 5   /* the next stage's patch is transformed and written to LDS after this many eighths of the
                                      stage's MFMA steps (PIPE 1 used 4) */
#endif
#ifndef EMO_CONV_LDS_PREFETCH
#define EMO_CONV_LDS_PREFETCH 0   /* 1: the LDS operands of MFMA step k+1 are read before the MFMAs of step k are issued.
                                     Measured neutral (bench 132.0 vs 131.9 frames/s, archive/profiles/r2_conv_mfma_stream_variants.jsonl):
                                     with 4-5 waves per SIMD the operand latency is already covered by the other waves */
#endif
#ifndef EMO_CONV_SETPRIO
#define EMO_CONV_SETPRIO 1   /* 1: s_setprio 1 while a wave is in its MFMA stream, 0 around the staging of the next stage: the
                                blocks of a CU are at independent phases, so the arbiter prefers a wave that has MFMAs to issue over
                                one that is transforming / storing its patch (cdna_hip_programming.md T5).  Measured +0.4 %
                                (132.4 vs 131.8 frames/s, twice each) */
#endif
#ifndef EMO_CONV_QUAD_1X1
#define EMO_CONV_QUAD_1X1 0   /* 1: 1x1 kernels stage their patch by 16-byte quads (see the kernel template); 0: element by element.
                                 Measured neutral (tools/session/r2_call22.sh: 1536->512 at 64^2 115.6 vs 114.3 TF, 512->320 98.5 /
                                 111.5 vs 93.7 / 113.1, all 225 GPU tests green with it on): the loop shrinks from 252 to 175
                                 instructions per stage and the rate does not move -- the 1x1 layers are not bound by staging
                                 instructions; their 128 x 128 tile re-reads 9x more patch per MFMA than a 3x3 tile from L2 */
#endif
#ifndef EMO_CONV_ABLATE
#define EMO_CONV_ABLATE 0   /* timing experiments only (results are WRONG for any value != 0): 1 = no global loads, no LDS-DMA and
                               no LDS stores in the loop; 5 = LDS stores of stale registers, no global loads, no LDS-DMA */
#endif

// QUAD (1x1 kernels without upsample, chosen by the launcher when the input is 16-byte aligned): the patch is staged by quads
// -- a lane loads 4 consecutive positions of one channel with one 16-byte load and stores them with one ds_write_b128 -- and
// the per-channel scale / shift arrive as one per-lane load each.  A 1x1 stage has 9x less MFMA work per staged element than
// a 3x3 one, and global loads are bound by their number (tools/microbench/vmem_rate.hip): per wave and stage 2 + 4 VMEM
// instructions instead of 8 + 8.  The LDS image, and with it the MFMA side of the kernel, is the same.
template <int KH, int KW, int KC, int TZ, int TR, int TW, int TM, int TP, int WGM, int WGP, bool UPS, bool QUAD = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(ConvCfg<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>::OCC,
                                   ConvCfg<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>::OCC)))
void conv_igemm_kernel(const ConvArgs a) {
  using Cfg = ConvCfg<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS>;
  static_assert(!QUAD || (KH == 1 && KW == 1 && !UPS), "quad staging: 1x1 kernels on the source grid");
  constexpr int BM = Cfg::BM, TAPS = Cfg::TAPS, PR = Cfg::PR, PW = Cfg::PW, CHS = Cfg::CHS;
  constexpr int ASZ = Cfg::ASZ, BUF = Cfg::BUF, NPE = Cfg::NPE, NSTEPS = Cfg::NSTEPS;
  constexpr int SW = Cfg::SW, WPC = Cfg::WPC, CPW = Cfg::CPW, EPC = Cfg::EPC;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  if (a.run_if != nullptr && *a.run_if == 0) return;   // guarded exact recomputation of a fp16-split pointwise layer (emo_conv_igemm_f32_guarded)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave id as a scalar (SGPR): everything derived from it (staged channel, base pointers, GN scale/shift) is then
  // wave-uniform and handled by the scalar unit instead of costing VALU issue slots next to the MFMA stream
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int chan0 = KC >= SW ? wave : wave % KC;         // channel (within a chunk) this wave stages; further ones at + g*SW
  const int part = KC >= SW ? 0 : wave / KC;             // which 64-element slices of that channel's patch (WPC waves share it)
  const int wm = wave / WGP, wp = wave % WGP;
  const int m0 = wm * TM * 32, p0 = wp * TP * 32;

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
  const int ptile = bx;
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
  // Branch-free input transform: v = max(v * sc + sh, floor).  Without an affine the loads still happen (from the
  // input itself, any readable address) and are replaced by (1, 0); without ReLU the floor is -inf.
  const float* scale_n = has_affine ? a.scale + (long)n * a.Cin : a.x;
  const float* shift_n = has_affine ? a.shift + (long)n * a.Cin : a.x;
  const float relu_floor = relu_in ? 0.0f : -__builtin_huge_valf();

  // ---- patch staging map: wave w stages input channels {w, w+4, ...} of the chunk; lane l element l + 64*i of the
  //      channel's [TZ][PR][PW] patch.  Per element only a plane offset and a validity bit are kept (constant over
  //      stages); the channel / depth part of the address is a scalar base per stage. ----
  unsigned p_off[EPC]; // BYTE offset (ys*W + xs)*4 inside an input plane (0 when the element is outside the image)
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

  // TZ == 1 (every 2-D layer): an element's validity never changes, so zero padding rides on the clamp of the ReLU --
  // v_med3(v, lo, hi) with [relu floor, +inf] inside the image and [0, 0] outside replaces max + select (and the mask
  // arithmetic behind the select) in the staging of every stage; a missing channel / depth slice (wave-uniform) zeroes the
  // scale and shift instead
  float p_lo[EPC], p_hi[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    p_lo[i] = p_ok[i] ? relu_floor : 0.0f;
    p_hi[i] = p_ok[i] ? __builtin_huge_valf() : 0.0f;
  }

  // ---- quad staging map (QUAD): thread t owns quad t % QPC of channel slot t / QPC; a stage has KC / CPP passes ----
  constexpr int QPC = QUAD ? CHS / 4 : 1;             // quads per channel
  constexpr int CPP = QUAD ? 256 / QPC : 1;           // channels staged per pass of the whole block
  constexpr int NPASS = QUAD ? KC / CPP : 1;
  static_assert(!QUAD || (CHS % 4 == 0 && TW % 4 == 0 && 256 % QPC == 0 && KC % CPP == 0), "whole quads, whole passes");
  const int q_i = tid % QPC, q_cs = tid / QPC;
  unsigned q_off = 0;                                 // byte offset of the quad inside a channel volume
  if (QUAD) {
    const int e0 = 4 * q_i;
    const int pz = e0 / (PR * PW), rem2 = e0 - pz * (PR * PW);
    const int pr = rem2 / PW, pc = rem2 - pr * PW;
    q_off = (unsigned)(((z0 + pz) * a.H + (y0 + pr)) * a.W + (x0 + pc)) * 4u;   // 1x1: no halo, always inside the volume
  }
  floatx4 qv[NPASS];    // the quads of a stage (pinned asm loads)
  float qsc[NPASS], qsh[NPASS];
  bool qcv[NPASS];      // the quad's channel exists

  const int nstages_all = a.n_cchunks * a.KD;
  const int st_begin = ks * a.stages_per_split;                       // this block's share of the K loop
  const int st_end = min(nstages_all, st_begin + a.stages_per_split);
  const floatx4* wsrc = reinterpret_cast<const floatx4*>(a.wpk) + ((long)cotile * nstages_all) * (ASZ / 4);

  // per-lane dump slot behind the two stage buffers (written, never read)
  float* const dump1 = smem + 2 * BUF + lane;

  float pv[NPE];        // staged patch values (raw)
  bool pvz[NPE];        // per-element depth validity (only varies per element when TZ > 1)
  bool sv[CPW];         // wave-uniform: channel exists (and, for TZ == 1, the depth slice is inside the volume)
  float sc[CPW], sh[CPW];
  constexpr int NGL = (ASZ * 4 + 1024 * SW - 1) / (1024 * SW);   // 1-KiB LDS-DMA pieces per staging wave

// The staging steps are macros (not lambdas / conditionals) so that pv[] is an unconditionally defined straight-line value
// and stays in VGPRs (a conditional or lambda-captured definition sent it to scratch).
// PINNED_: loads as asm volatile (EMO_CONV_PIPE 2) or ordinary loads (EMO_CONV_PIPE 1)
#define EMO_ISSUE_PATCH(stage_, PINNED_)                                                              \
  {                                                                                                   \
    const int scc_ = (stage_) / a.KD;                                                                 \
    EMO_ISSUE_PATCH_AT(scc_, (stage_) - scc_ * a.KD, PINNED_)                                         \
  }
/* the same for a stage given as (channel chunk, depth tap): the K loop steps these instead of dividing every stage */
#define EMO_ISSUE_PATCH_AT(cc_, t_, PINNED_)                                                          \
  if (QUAD) {                                                                                         \
    _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                            \
      const int c_ = (cc_) * KC + ps * CPP + q_cs;                                                    \
      qcv[ps] = c_ < a.Cin;                                                                           \
      const unsigned cs_ = qcv[ps] ? (unsigned)c_ : 0u;                                               \
      /* 32-bit byte offsets inside the sample: the launcher checks Cin * D * H * W * 4 < 2^32 */     \
      qv[ps] = emo_gload4_pinned(xn, q_off + cs_ * (unsigned)DHW * 4u);                               \
      if (has_affine) {                                                                               \
        qsc[ps] = emo_gload_pinned(scale_n, cs_ * 4u);                                                \
        qsh[ps] = emo_gload_pinned(shift_n, cs_ * 4u);                                                \
      }                                                                                               \
    }                                                                                                 \
  } else {                                                                                            \
    const int ci0_ = (cc_) * KC;                                                                      \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) {                                                 \
      const int c_ = ci0_ + g * SW + chan0;                                                           \
      const bool cv_ = c_ < a.Cin;                                                                    \
      const int cs_ = cv_ ? c_ : 0;                                                                   \
      const int zu_ = z0 + (t_) - padD;         /* depth of tile slice 0 */                           \
      const bool zv_ = (unsigned)zu_ < (unsigned)a.D;                                                 \
      const float* base_ = xn + (long)cs_ * DHW + (long)((TZ == 1 && zv_) ? zu_ : 0) * HW;            \
      sv[g] = cv_ && (TZ > 1 || zv_);                                                                 \
      if (PINNED_) {                                                                                  \
        sc[g] = emo_gload_pinned(scale_n + cs_, 0u);                                                  \
        sh[g] = emo_gload_pinned(shift_n + cs_, 0u);                                                  \
      } else {                                                                                        \
        sc[g] = scale_n[cs_]; sh[g] = shift_n[cs_];                                                   \
      }                                                                                               \
      _Pragma("unroll") for (int i = 0; i < EPC; ++i) {                                               \
        unsigned off_ = p_off[i];                                                                     \
        if (TZ == 1) {                                                                                \
          pvz[g * EPC + i] = true;                                                                    \
        } else {                                                                                      \
          const int zi = zu_ + p_pz[i];                                                               \
          const bool zok = (unsigned)zi < (unsigned)a.D;                                              \
          pvz[g * EPC + i] = zok;                                                                     \
          off_ += (unsigned)((zok ? zi : 0) * HW) * 4u;                                               \
        }                                                                                             \
        if (PINNED_) pv[g * EPC + i] = emo_gload_pinned(base_, off_);                                 \
        else pv[g * EPC + i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_) + off_); \
      }                                                                                               \
    }                                                                                                 \
  }

/* wait for the pinned loads of EMO_ISSUE_PATCH and pin every consumer behind the wait */
#define EMO_WAIT_PATCH()                                                                              \
  if (QUAD) {                                                                                         \
    emo_wait_vmem0();                                                                                 \
    _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                            \
      emo_touch4(qv[ps]);                                                                             \
      if (has_affine) { emo_touch(qsc[ps]); emo_touch(qsh[ps]); }                                     \
    }                                                                                                 \
  } else {                                                                                            \
    emo_wait_vmem0();                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < NPE; ++q_) emo_touch(pv[q_]);                             \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) { emo_touch(sc[g]); emo_touch(sh[g]); }           \
  }

/* weight tile: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction), no VGPR round trip. */
/* The LDS image is lane-linear = exactly the packed weight order.                               */
#define EMO_ISSUE_WEIGHTS(stage_, dst_)                                                               \
  {                                                                                                   \
    const floatx4* ws_ = wsrc + (long)(stage_) * (ASZ / 4);                                           \
    _Pragma("unroll") for (int i = 0; i < NGL; ++i) {                                                 \
      const int j = wave + SW * i;                                                                    \
      const int boff = j * 1024 + lane * 16;                                                          \
      if (boff < ASZ * 4)                                                                             \
        __builtin_amdgcn_global_load_lds(                                                             \
            (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(ws_) + boff), \
            (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(dst_) + j * 1024), 16, 0, 0); \
    }                                                                                                 \
  }

/* Stores are unconditional: lanes beyond the tile write to a private dump slot (address select, no branch). */ \
#define EMO_STORE_STAGE(buf_)                                                                         \
  if (QUAD) {                                                                                         \
    float* Ps_ = (buf_) + ASZ;                                                                        \
    _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                            \
      float sc_ = has_affine ? qsc[ps] : 1.0f, sh_ = has_affine ? qsh[ps] : 0.0f;                     \
      sc_ = qcv[ps] ? sc_ : 0.0f;                     /* a missing channel stages zeros */            \
      sh_ = qcv[ps] ? sh_ : 0.0f;                                                                     \
      floatx4 v_;                                                                                     \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                   \
        v_[e] = __builtin_amdgcn_fmed3f(__fmaf_rn(qv[ps][e], sc_, sh_), relu_floor, __builtin_huge_valf()); \
      *reinterpret_cast<floatx4*>(Ps_ + (ps * CPP + q_cs) * CHS + 4 * q_i) = v_;                      \
    }                                                                                                 \
  } else {                                                                                            \
    float* Ps_ = (buf_) + ASZ;                                                                        \
    _Pragma("unroll") for (int g = 0; g < CPW; ++g) {                                                 \
      float* Pc_ = Ps_ + (g * SW + chan0) * CHS;                                                      \
      float sc_ = has_affine ? sc[g] : 1.0f, sh_ = has_affine ? sh[g] : 0.0f;                         \
      if (TZ == 1) { sc_ = sv[g] ? sc_ : 0.0f; sh_ = sv[g] ? sh_ : 0.0f; }                            \
      _Pragma("unroll") for (int i = 0; i < EPC; ++i) {                                               \
        const int e = lane + (i * WPC + part) * 64;                                                   \
        float v = __fmaf_rn(pv[g * EPC + i], sc_, sh_);                                               \
        /* zero padding applies to the transformed tensor */                                          \
        if (TZ == 1) {                                                                                \
          v = __builtin_amdgcn_fmed3f(v, p_lo[i], p_hi[i]);                                           \
        } else {                                                                                      \
          v = fmaxf(v, relu_floor);                                                                   \
          v = (p_ok[i] && sv[g] && pvz[g * EPC + i]) ? v : 0.0f;                                      \
        }                                                                                             \
        float* d_ = ((i + 1) * WPC * 64 <= CHS || e < CHS) ? Pc_ + e : dump1;                         \
        *d_ = v;                                                                                      \
      }                                                                                               \
    }                                                                                                 \
  }

  constexpr bool PINNED = EMO_CONV_PIPE == 2;
  static_assert(!QUAD || PINNED, "quad staging exists for the pinned-load schedule only");
  constexpr int STORE_STEP = (NSTEPS * (PINNED ? EMO_CONV_STORE_EIGHTHS : 4)) / 8;

  // ---- prologue: stage st_begin into buffer 0 ----
  EMO_ISSUE_WEIGHTS(st_begin, smem);
  EMO_ISSUE_PATCH(st_begin, PINNED);
  if (PINNED) { EMO_WAIT_PATCH(); }
  EMO_STORE_STAGE(smem);
  if (!PINNED) {   // round-1 schedule: the second stage is in flight across the barrier (registers are loop-carried)
    const int st1_ = (st_begin + 1) < st_end ? (st_begin + 1) : st_begin;
    EMO_ISSUE_PATCH(st1_, false);
  }
  __syncthreads();

  // accumulators in two arrays of at most 64 floats each: one 128-float array (TM*TP = 8) is not scalarised by the
  // compiler (it stays a scratch alloca that the K loop loads and stores around every MFMA)
  constexpr int TPH = TP > 2 ? TP / 2 : TP;
  floatx16 acc_lo[TM][TPH], acc_hi[TM][TPH];
#define acc_at(i_, j_) ((j_) < TPH ? acc_lo[i_][(j_) < TPH ? (j_) : 0] : acc_hi[i_][(j_) >= TPH ? (j_) - TPH : 0])
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TPH; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc_lo[i][j][r] = 0.0f; acc_hi[i][j][r] = 0.0f; }

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

  int nx_cc = st_begin / a.KD, nx_t = st_begin - nx_cc * a.KD;   // (channel chunk, depth tap) of the stage being loaded
  for (int st = st_begin; st < st_end; ++st) {
    float* cur = smem + ((st - st_begin) & 1) * BUF;
    float* nxt = smem + ((st - st_begin + 1) & 1) * BUF;
    // prefetch the next stage while this one computes; on the last stage the (clamped) prefetch re-reads the
    // last stage and its LDS write lands in the idle buffer -- harmless, and it keeps the loop branch-free
    const int stn = (st + 1) < st_end ? (st + 1) : st;
    const float* As = cur;
    const float* Ps = cur + ASZ;
    if (EMO_CONV_ABLATE == 0) {
      EMO_ISSUE_WEIGHTS(stn, nxt);
      if (st + 1 < st_end && ++nx_t == a.KD) { nx_t = 0; ++nx_cc; }   // (chunk, tap) of stage stn
      if (PINNED) { EMO_ISSUE_PATCH_AT(nx_cc, nx_t, true); }
    }
    // one MFMA step = one channel pair x one tap: 1 A read + 1 B read per 32x32 tile, TM*TP MFMAs
#define EMO_READ_OPERANDS(step_, av__, bv__)                                                          \
    {                                                                                                 \
      const int pair_ = (step_) / TAPS, tap_ = (step_) % TAPS;                                        \
      const int r_ = tap_ / KW, s_ = tap_ % KW;                                                       \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) av__[i] = As[a_base + ((pair_ * TAPS + tap_) * 2) * BM + i * 32]; \
      _Pragma("unroll") for (int j = 0; j < TP; ++j) bv__[j] = Ps[b_base[j] + (pair_ * 2) * CHS + r_ * PW + s_];       \
    }
    float av_[2][TM], bv_[2][TP];   // operand registers of the current and (EMO_CONV_LDS_PREFETCH) the next step
    if (EMO_CONV_LDS_PREFETCH) { EMO_READ_OPERANDS(0, av_[0], bv_[0]); }
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int step = 0; step < NSTEPS; ++step) {
      if (step == STORE_STEP && (EMO_CONV_ABLATE == 0 || EMO_CONV_ABLATE == 5)) {
        if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
        if (PINNED && EMO_CONV_ABLATE == 0) { EMO_WAIT_PATCH(); }
        EMO_STORE_STAGE(nxt);
        if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
      }
      const int cur_ = EMO_CONV_LDS_PREFETCH ? (step & 1) : 0;
      if (EMO_CONV_LDS_PREFETCH) {
        if (step + 1 < NSTEPS) { EMO_READ_OPERANDS(step + 1, av_[cur_ ^ 1], bv_[cur_ ^ 1]); }
      } else {
        EMO_READ_OPERANDS(step, av_[0], bv_[0]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
          acc_at(i, j) = __builtin_amdgcn_mfma_f32_32x32x2f32(bv_[cur_][j], av_[cur_][i], acc_at(i, j), 0, 0, 0);   // [position][channel]: conv_epilogue
    }
    if (EMO_CONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
#undef EMO_READ_OPERANDS
    if (!PINNED && EMO_CONV_ABLATE == 0) {
      // round-1 schedule: refill the registers with stage st+2 before the barrier
      const int stn2 = (st + 2) < st_end ? (st + 2) : (st_end - 1);
      EMO_ISSUE_PATCH(stn2, false);
    }
    __syncthreads();
  }

  conv_epilogue<TZ, TR, TW, TM, TP, WGP, BM>(a, acc_lo, acc_hi, smem, n, cotile, ptile, ks, x0, y0, z0, m0, p0, wp, half, l32, tid);
}

#undef acc_at
#undef EMO_ISSUE_PATCH
#undef EMO_ISSUE_PATCH_AT
#undef EMO_WAIT_PATCH
#undef EMO_ISSUE_WEIGHTS
#undef EMO_STORE_STAGE

// opt in to more than 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU) once per (kernel, device).  Every instantiation
// has the same function-pointer TYPE, so the bookkeeping is keyed by the pointer VALUE.
template <typename K>
static int emo_raise_dynamic_lds(K kern) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> raised;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const std::pair<const void*, int> key(reinterpret_cast<const void*>(kern), dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!raised.count(key)) {
    e = hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    raised.insert(key);
  }
  return EMO_OK;
}

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
  const size_t lds = (size_t)Cfg::LDS_BYTES;
  if (lds > 160 * 1024) return EMO_ERR_UNSUPPORTED;
  // 1x1 kernels stage by 16-byte quads when they can: input 16-byte aligned (rows are: W % 4 == 0 follows from the tile
  // shapes) and 32-bit byte offsets inside a sample
  constexpr bool CAN_QUAD = KH == 1 && KW == 1 && !UPS && EMO_CONV_PIPE == 2 && EMO_CONV_QUAD_1X1 &&
                            (TZ * TR * TW) % 4 == 0 && 256 % ((TZ * TR * TW) / 4) == 0 && KC % (256 / ((TZ * TR * TW) / 4)) == 0;
  const bool quad = CAN_QUAD && (reinterpret_cast<unsigned long long>(a.x) & 15ull) == 0 && (a.W & 3) == 0 &&
                    (unsigned long long)a.Cin * a.D * a.H * a.W * 4ull < (1ull << 32);
  void (*kern)(const ConvArgs) = conv_igemm_kernel<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS, false>;
  if (quad) kern = conv_igemm_kernel<KH, KW, KC, TZ, TR, TW, TM, TP, WGM, WGP, UPS, CAN_QUAD>;
  if (lds > 64 * 1024) {
    const int rc = emo_raise_dynamic_lds(kern);
    if (rc != EMO_OK) return rc;
  }
  a.n_cotiles = cot;
  if (a.ksplit < 1 || (a.ksplit > 1 && (!EMO_CONV_XCD_ORDER || !a.partial))) return EMO_ERR_BAD_ARG;
  if (a.ksplit == 1) { a.stages_per_split = a.n_cchunks * a.KD; a.partial = nullptr; }
  if (a.ksplit > 1 && a.gn_stats) return EMO_ERR_BAD_ARG;
  if (nt * cot * a.N * a.ksplit > 0x7fffffffL) return EMO_ERR_UNSUPPORTED;
  dim3 g = EMO_CONV_XCD_ORDER ? dim3((unsigned)(nt * cot * a.N * a.ksplit)) : dim3((unsigned)nt, cot, a.N);
  hipLaunchKernelGGL(kern, g, dim3(Cfg::THREADS), lds, s, a);
  return emo_launch_status();
}
