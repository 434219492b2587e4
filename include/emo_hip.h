/*
 * emo_hip.h -- C ABI of libemoportraits_hip.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * EMOPortraits volumetric-avatar *inference hot path* (SURVEY.md section 8).
 *
 * The reference (neeek2303/EMOPortraits) is pure Python/PyTorch and has NO plugin / FFI interface for this
 * path; every entry point below therefore replaces a stock torch op *call site* of the reference, cited as
 * file:line relative to the reference root.  INTEGRATION.md shows the ctypes stub a reference maintainer
 * would add at each call site.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  All tensor pointers are DEVICE pointers to contiguous fp32
 *     (unless stated), at least 16-byte aligned.  The caller owns every buffer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is asynchronous on it.
 *   - return value: 0 on success; EMO_ERR_* (negative) for argument errors; a positive hipError_t if a
 *     launch failed.  No global mutable state: calls are thread-safe.
 *   - fp32 throughout; index arithmetic of the sampler is bit-identical to ATen's CPU grid_sampler_3d.
 */
#ifndef EMO_HIP_H_
#define EMO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMO_ABI_VERSION 10

#define EMO_OK 0
#define EMO_ERR_BAD_ARG (-1)       /* null pointer / non-positive size / unknown enum          */
#define EMO_ERR_UNSUPPORTED (-2)   /* valid request outside what the kernels implement         */
#define EMO_ERR_ALIGN (-3)         /* pointer not 16-byte aligned                              */

/* padding_mode of torch.nn.functional.grid_sample (args.grid_sample_padding_mode,
 * models/stage_1/volumetric_avatar/va_arguments.py:185; default 'zeros') */
#define EMO_PAD_ZEROS 0
#define EMO_PAD_BORDER 1
#define EMO_PAD_REFLECTION 2

/* memory layout of a 5-D volume */
#define EMO_LAYOUT_NCDHW 0         /* the reference's layout                                    */
#define EMO_LAYOUT_NDHWC 1         /* channels-last, internal fast path (gathers read C contiguous floats) */
/* 2: retired (channel-group-per-XCD layout of round 2: equal time) */
#define EMO_LAYOUT_P4 3            /* packed-4: [N][C/4][D][H][W][4] -- a voxel's channel quad is one 16-byte slot, x-rows are
                                      contiguous: the layout of the LDS-staged sampler (one LDS-DMA lane per box voxel, ds_read_b128
                                      gathers).  Needs C % 4 == 0.  */

/* activation applied in a kernel epilogue */
#define EMO_ACT_NONE 0
#define EMO_ACT_RELU 1
#define EMO_ACT_TANH 2
#define EMO_ACT_SIGMOID 3

int emo_abi_version(void);
/* human-readable build string: arch, compiler, kernel variants */
const char* emo_build_info(void);
/* ABI 9.  Compute units of the current device as the launchers count them (rounded down to a multiple of 8, at least 8): the
 * persistent grids of the split convolutions and the thresholds of their launch forms are sized by it, and so is the Python
 * planner (emoportraits_amd/pack.py) -- no reference counterpart (the reference leaves launch geometry to cuDNN). */
int emo_device_cu_count(void);
/* ABI 9.  Diagnostic (bench.py `roofline.sustained_peak`, not on the hot path): one launch of a bare v_mfma_f32_32x32x16_f16
 * stream -- one wave per SIMD on every CU, `iters` x 96 MFMAs per wave on pseudo-random operands; lds_reads != 0 adds the
 * fragment reads of the split kernels' K loop (8 ds_read_b128 per 12 MFMAs).  *mfma_per_launch = MFMA instructions the launch
 * issues (x 32768 flop each).  sink: >= 256 x CUs floats (never written in practice). */
int emo_mfma_stream_f16(float* sink, int iters, int lds_reads, int64_t* mfma_per_launch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a1 -- 3-D trilinear grid_sample, align_corners=False.
 * Replaces Model.grid_sample = lambda inputs, grid: F.grid_sample(inputs.float(), grid.float(),
 * padding_mode=args.grid_sample_padding_mode)  (models/stage_1/volumetric_avatar/va.py:264-265);
 * call sites notebooks/infer.py:499-500 (source) and :618-619 (driver).
 *
 *   vol   [Nv, C, D, H, W] (NCDHW) or [Nv, D, H, W, C] (NDHWC); Nv = N, or 1 with vol_batch_stride = 0
 *         (one canonical volume shared by N driver frames -- the reference loops batch-1 calls instead).
 *   grid  [N, Do, Ho, Wo, 3] (x, y, z) in [-1, 1] coordinates; or NULL when `theta` is given.
 *   theta [N, 3, 4] row-major affine (rows 0..2 of the 4x4 head-pose matrix).  When non-NULL the grid is
 *         generated in-kernel:  g_j = fma-chain( lin_x[x]*t_j0, lin_y[y]*t_j1, lin_z[z]*t_j2, t_j3 )
 *         which is what identity_grid_3d.bmm(theta[:, :3].transpose(1, 2)) computes
 *         (va.py:101-105 + notebooks/infer.py:441-444,583-588) without materialising the 0.79 MB grid.
 *   lin_x [Wo], lin_y [Ho], lin_z [Do]  the identity lattice (torch.linspace(-1, 1, n) values); required
 *         with theta or grid_kind 1, ignored otherwise.
 *   grid_kind  0: `grid` holds coordinates [N,Do,Ho,Wo,3].  1: `grid` holds planar deltas [N,3,Do,Ho,Wo] and the
 *         coordinate is lattice + delta -- the WarpGenerator output  warp = (identity_grid + deltas).permute(0,2,3,4,1)
 *         (networks/volumetric_avatar/warp_generator_resnet.py:178) consumed without materialising `warp`.
 *   variant    0 = default kernels.  NCDHW -> NCDHW: channels per block of the direct gather (1..C).  NDHWC input, bits:
 *         1 = 4 x 4 x 4 output bricks per block instead of 64-voxel rows (NDHWC output); 2 = non-temporal stores of an
 *         NCDHW output (a result that is read much later: the driver pass's rotation call); 4 = fused multiply-add accumulation
 *         of the eight corners (opt-in: coordinates, floor, corner indices and weights stay bit-identical to ATen's CPU kernel,
 *         the sampled VALUES differ from it by the skipped product roundings, <= 8 * 2^-24 * max |v w| absolute).
 *         For the LDS-staged tile kernels (in_layout EMO_LAYOUT_P4, or NCDHW -> NCDHW with bit 30 set) it is a tuning word:
 *         bits 3..0 / 7..4 / 11..8 log2 of the output tile extents x / y / z (all 0: default), 16..12 channel units per
 *         block, 24..17 LDS per block in KiB, bit 25: 512-thread blocks.  tile voxels = threads or 2 * threads.
 *   out   [N, C, Do, Ho, Wo] or [N, Do, Ho, Wo, C] according to out_layout.
 *   vol_batch_stride  elements between consecutive volumes (0 = shared volume).
 * NDHWC paths require C % 4 == 0 and support out_layout NDHWC and NCDHW.
 * in_layout EMO_LAYOUT_P4 (C % 4 == 0) supports out_layout P4 and NCDHW: the LDS-staged kernels (csrc/gs3d_tile.h) --
 * the source box of an output tile is brought into LDS by LDS-DMA once and the 8-corner gather runs out of LDS.
 */
int emo_grid_sample3d_f32(const float* vol, const float* grid, const float* theta,
                          const float* lin_x, const float* lin_y, const float* lin_z,
                          float* out,
                          int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                          int64_t vol_batch_stride, int padding_mode,
                          int in_layout, int out_layout, int variant, int grid_kind, void* stream);

/* a2 alone -- the rotation warp as a tensor: grid[n,z,y,x,:] = theta[n,:3,:4] . (lin_x[x], lin_y[y], lin_z[z], 1), the same
 * fma chain the theta variant of emo_grid_sample3d_f32 evaluates in-kernel.  Replaces
 * `identity_grid_3d.bmm(theta[:, :3].transpose(1, 2)).view(-1, d, s, s, 3)` (notebooks/infer.py:441-444, :583-588; cached there
 * as self.source_rotation_warp).  theta [N,3,4] row-major, grid [N,Do,Ho,Wo,3]. */
int emo_affine_grid3d_f32(const float* theta, const float* lin_x, const float* lin_y, const float* lin_z,
                          float* grid, int N, int Do, int Ho, int Wo, void* stream);

/* Layout repack of a 5-D volume (used once per identity on the cached canonical volume, notebooks/infer.py:507
 * `self.target_latent_volume`).  to_channels_last: 0 NDHWC -> NCDHW, 1 NCDHW -> NDHWC, 4 NCDHW -> P4, 5 P4 -> NCDHW. */
int emo_volume_repack_f32(const float* in, float* out, int N, int C, int DHW, int to_channels_last, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a10 -- GroupNorm statistics folded to a per-(sample, channel) affine.
 * Replaces the nn.GroupNorm(32, C) / AdaptiveGroupNorm in front of every conv of the reference's ResBlock
 * (networks/volumetric_avatar/utils.py:711-731 block_feats; registries :953-957; AdaptiveGroupNorm :302-325):
 * the normalised tensor is never written, the consumer conv applies x*scale+shift (+ReLU) while staging.
 *   x [N, C, S] (S = product of spatial dims), G groups, eps as nn.GroupNorm (1e-5).
 *   gamma/beta [C] or NULL (=1/0).  ada_gamma/ada_beta [N rows, ada_stride apart] or NULL: the per-sample
 *   weights assigned by assign_adaptive_norm_params (utils.py:983-995); y = (xhat*gamma+beta)*ada_gamma+ada_beta.
 *   scale/shift [N, C] out.  mean_out/rstd_out [N, G] optional (both or neither).
 *   workspace: device buffer of at least emo_groupnorm_workspace_bytes(N, G) bytes.
 */
int64_t emo_groupnorm_workspace_bytes(int N, int G);
int emo_groupnorm_affine_f32(const float* x, int N, int C, int64_t S, int G, float eps,
                             const float* gamma, const float* beta,
                             const float* ada_gamma, const float* ada_beta, int64_t ada_stride,
                             float* scale, float* shift, float* mean_out, float* rstd_out,
                             void* workspace, int64_t workspace_bytes, void* stream);
/* The same affine from the per-tile statistics emo_conv_igemm_f32 leaves in `gn_stats` (no pass over the activation):
 *   stats [N][T][C][2] = (mean, centred sum of squares) of `cnt` values each, T tiles per sample and channel
 *   (emo_conv_igemm_f32: cnt = emo_conv_tile_positions(cfg), T = D*Hl*Wl / cnt).  Combined per (sample, group) in fp64: the tile M2 values add, and the
 *   between-tile term cnt * (sum(mean_i^2) - K * mean^2) is evaluated in fp64 on the fp32 tile means (an fp64 combine of
 *   fp32 tile statistics; within a tile the conv epilogue uses centred sums, so no fp32 E[x^2] - mean^2 is ever formed).  Other arguments as emo_groupnorm_affine_f32. */
int emo_groupnorm_affine_from_tiles_f32(const float* stats, int N, int C, int64_t T, int cnt, int G, float eps,
                                        const float* gamma, const float* beta,
                                        const float* ada_gamma, const float* ada_beta, int64_t ada_stride,
                                        float* scale, float* shift, float* mean_out, float* rstd_out, void* stream);

/* ABI 8.  The second half of emo_groupnorm_affine_f32 alone: `partial` = `split` (sum, sum of squares) fp64 slices per (sample,
 * group), workspace layout [N * G][64][2], left by a producer that had the tensor in registers (emo_upsample_trilinear_gn_sums_f32);
 * S = elements per (sample, channel) of the tensor the sums are of.  Other arguments as emo_groupnorm_affine_f32. */
int emo_groupnorm_affine_from_sums_f32(const void* partial, int split, int N, int C, int64_t S, int G, float eps,
                                       const float* gamma, const float* beta,
                                       const float* ada_gamma, const float* ada_beta, int64_t ada_stride,
                                       float* scale, float* shift, float* mean_out, float* rstd_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a5/a9/a10 -- implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 * Replaces F.conv2d / F.conv3d inside ResBlock / ConvBlock (networks/volumetric_avatar/utils.py:661-788;
 * Conv2d_ws/Conv3d_ws :887-915; spectral-norm hook utils/spectral_norm.py:96-168 -- both folded into the packed
 * weights at load time), stride 1, "same" zero padding, kernel 1x1(x1) or 3x3 (2-D: KD=1) or 3x3x3 (KD=3).
 *   x     [N, Cin, D, H, W]            (D = 1 for 2-D convs)
 *   wpk   weights packed by the host for block config `cfg`: [co_tile][Cin chunk][kd][pair][tap][half][BM]
 *         (emoportraits_amd/pack.py; BM / KC from emo_conv_pack_info)
 *   bias  [Cout] or NULL.
 *   scale/shift [N, Cin] or NULL: input transform x*scale+shift (GroupNorm folded, see above);
 *         relu_in != 0 applies max(.,0) after it (ReLU of block_feats, utils.py:717,731).  Zero padding is applied
 *         to the transformed tensor, as F.conv does.
 *   ups   != 0: the conv input is the nearest-neighbour x2 upsampling of x in H and W (ResBlock
 *         resize_layer_type='nearest', utils.py:684-688,764-781); out is [N, Cout, D, 2H, 2W].  2-D only.
 *   res   residual added before `act` (ResBlock skip, utils.py:783); same shape as out, or the pre-upsample
 *         shape when res_ups != 0.  May alias `out`.
 *   act   EMO_ACT_* applied last (tanh head warp_generator_resnet.py:99-107; sigmoid head decoder.py:347-358).
 *   cfg   0: 128 output channels x 128 positions per block, 1: 64 x 128, 2: 32 x 128, 3: 64 x 256, 4: 64 x 512, 5: 32 x 256 (3-5: 3x3 layers, 3 and 5
 *         also 3x3x3; weights packed as for cfg 1 resp. 2).  emo_conv_tile_positions(cfg) = positions per block.
 *   ksplit / workspace   ksplit > 1 divides the K loop (input-channel chunks x depth taps) of every output tile over
 *         ksplit blocks -- small launches (64x64 maps at batch 1, the 8^3 / 16^3 WarpGenerator layers) otherwise leave most
 *         of the 256 CUs idle; partial sums go to workspace [ksplit][N*Cout*D*Hl*Wl] floats and a second kernel adds them
 *         in fixed order and applies bias / residual / activation (deterministic; `out` may alias `res`).  ksplit = 1,
 *         workspace = NULL: single pass.  emo_conv_igemm_ksplit returns the split count the launch heuristic wants.
 *   gn_stats   NULL, or [N][D*Hl*Wl/P][Cout][2] floats, P = emo_conv_tile_positions(cfg) (ksplit == 1 only): the epilogue
 *         also reduces, per sample, P-position tile and output channel, the mean and the centred sum of squares of the FINAL output values
 *         (after bias / residual / act) -- the statistics of the GroupNorm that follows in the next block
 *         (utils.py:711-731), consumed by emo_groupnorm_affine_from_tiles_f32 instead of a pass over `out`.
 * Supported output widths: multiples of 128, or 64 / 32 / 16 / 8 (with H resp. D divisible by the tile).
 */
int emo_conv_pack_info(int KH, int KW, int cfg, int* BM, int* KC);
int emo_conv_tile_positions(int cfg);
int emo_conv_igemm_f32(const float* x, const float* wpk, const float* bias,
                       const float* scale, const float* shift, const float* res, float* out,
                       int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW,
                       int ups, int relu_in, int act, int res_ups, int cfg, int ksplit, float* workspace,
                       float* gn_stats, void* stream);
int emo_conv_igemm_ksplit(int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW, int ups, int cfg);

/* Reduced-precision mode (BASELINE.json configs[4], "fp16 MFMA convs"; opt-in per layer, never the default): the same
 * convolution with fp16 MFMA operands (v_mfma_f32_32x32x16_f16) and fp32 accumulation.  Tensors in HBM stay fp32; the
 * producer's norm + ReLU is applied in fp32 and rounded to fp16 on the way into LDS.  3x3 (2-D / 3-D) and 1x1 kernels,
 * cfg 3 (64 x 256 tile) or, 3x3 only, cfg 6 (128 x 256 tile, one block per CU); output widths that are multiples of 128, or
 * 64 / 32; Cin % 8 == 0; x 16-byte aligned; Cin <= 1024 when scale / shift are given.  wpk16: fp16 weights packed
 * [co_tile][Cin chunk of KC][kd][q = KC/16][tap][half][BM][8] (channel in chunk = 16*q + 8*half + 0..7), BM / KC from
 * emo_conv_pack_info_f16.  Other arguments, incl. gn_stats, as emo_conv_igemm_f32. */
int emo_conv_pack_info_f16(int KH, int KW, int cfg, int* BM, int* KC);
int emo_conv_igemm_f16acc32(const float* x, const void* wpk16, const float* bias,
                            const float* scale, const float* shift, const float* res, float* out,
                            int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW,
                            int ups, int relu_in, int act, int res_ups, int cfg, int ksplit, float* workspace,
                            float* gn_stats, void* stream);

/* ABI 9.  The reduced-precision mode of BASELINE configs[4] ("fp16 MFMA convs") for the decoders' 3x3 / 3x3x3 layers on the
 * eight-wave two-tile kernel (csrc/conv_igemm_f16x2_w8.h, NPROD = 1): plain fp16 operands -- the LEADING product of the fp16
 * split alone --, fp32 accumulation, fp32 tensors; the operands saturate at +-65504 as in emo_conv_igemm_f16acc32 (no range
 * word, no guarded recomputation).  wpk1 = emoportraits_amd.pack.pack_weight_f16w8(w): the first plane of the split layout,
 * [channel tile][Cin chunk of 16][kd][kernel row][kernel column][half][64][8] fp16 of w * w_scale (a power of two; the result
 * is multiplied by 1 / w_scale).  One launch form only -- the straight-line epilogue's: cfg 3, ksplit 1, no activation, whole
 * 64-channel tiles, 4 x 64 position tiles (Wl % 64 == 0, Hl % 4 == 0), 16-byte aligned tensors, at least two pair items per CU --
 * EMO_ERR_UNSUPPORTED otherwise (run emo_conv_igemm_f16acc32).  Replaces the same reference code as emo_conv_igemm_f32
 * (networks/volumetric_avatar/utils.py:661-788) under notebooks/infer_s2.py:351-387's decoder. */
int emo_conv_igemm_f16w8(const float* x, const void* wpk1, const float* bias, const float* scale, const float* shift,
                         const float* res, float* out, int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW,
                         int ups, int relu_in, int act, int res_ups, int cfg, int ksplit, float* workspace, float* gn_stats,
                         void* stream, float w_scale);

/* ABI 10.  The same mode for a layer with an ODD number (>= 3) of 64-channel tiles (the decoder's 192- and 320-channel layers,
 * networks/volumetric_avatar/decoder.py:277): emo_conv_igemm_f16w8 would run the last tile in a half-empty pair that costs a
 * whole pair's staging; here the whole pairs run that kernel (wpk1, as above) and the last tile runs
 * emo_conv_igemm_f16acc32's kernel on the same arguments (wpk16: THAT entry's layout for cfg 3, the whole layer's weights) --
 * two launches on `stream`, every output element and every gn_stats entry written once.  The launch must be in the form of
 * BOTH kernels (emo_conv_igemm_f16w8's above; emo_conv_igemm_f16acc32's: Wl % 128 == 0 or Wl == 64); EMO_ERR_UNSUPPORTED
 * otherwise and for an even or single tile count. */
int emo_conv_igemm_f16w8_rest(const float* x, const void* wpk1, const void* wpk16, const float* bias, const float* scale,
                              const float* shift, const float* res, float* out, int N, int Cin, int Cout, int D, int H, int W,
                              int KD, int KH, int KW, int ups, int relu_in, int act, int res_ups, int cfg, int ksplit,
                              float* workspace, float* gn_stats, void* stream, float w_scale);

/* fp32 3x3 convolution on the bf16 matrix pipes (NOT a reduced-precision mode): every fp32 operand is the exact sum of three
 * bf16 terms (x = xh + xm + xl, 24 significand bits kept), the six partial products of order <= 2^-16 are accumulated in fp32
 * by v_mfma_f32_32x32x16_bf16; the dropped terms are <= 2^-23 |x w| per product, below the fp32 accumulation error of either
 * kernel.  Same arguments, epilogue, K split and gn_stats as emo_conv_igemm_f32 (reference: torch.nn.Conv2d / Conv3d forward
 * of the decoder and WarpGenerator blocks, networks/volumetric_avatar/utils.py:854-1005, fp32).  3x3 (KD 1 or 3) kernels,
 * cfg 3 (64 x 256 tile = 4 x 64 pixels, or 8 x 32 / 16 x 16 on 32- / 16-wide maps without upsample; one block per CU); output
 * width % 64 == 0 and height % 4 == 0, or width 32 and height % 8 == 0, or width 16 and height % 16 == 0; Cin % 8 == 0; x 16-byte
 * aligned; Cin <= 1024 when scale / shift are given.  wpk3: bf16 weights packed
 * [co_tile][Cin chunk of 16][kd][kernel row][plane h|m|l][kernel column][half][BM = 64][8] (channel in chunk = 8*half + 0..7).
 * Operand range (tests/test_conv_bf16x3_gpu.py::test_conv_bf16x3_operand_contract): the split is exact for every finite staged
 * value up to the largest finite bf16, 3.3895e38 (0x7f7f0000); staged values beyond it, +-inf included, SATURATE there (their
 * first bf16 term would round to infinity and the residual inf - inf to NaN) -- the exact-fp32 kernel passes +-inf on.  A NaN
 * input is staged as the lower clamp bound (0 with relu_in, else -3.39e38) by the v_med3 that also applies ReLU and zero padding,
 * exactly as in emo_conv_igemm_f32.  Signed zeros and fp32 subnormals are split like any other value (a subnormal's terms are
 * bf16 subnormals; the matrix pipe may flush them: absolute error <= 2^-126 per product).
 * run_if   NULL, or a device word: the launch (both halves of a K-split launch) does nothing unless *run_if != 0 when it starts
 *          executing -- the guarded fallback behind emo_conv_igemm_f16x2's overflow_flag (same stream, launched right after it). */
int emo_conv_pack_info_bf16x3(int KH, int KW, int cfg, int* BM, int* KC);
int emo_conv_igemm_bf16x3(const float* x, const void* wpk3, const float* bias,
                          const float* scale, const float* shift, const float* res, float* out,
                          int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW,
                          int ups, int relu_in, int act, int res_ups, int cfg, int ksplit, float* workspace,
                          float* gn_stats, void* stream, const int* run_if);

/* Companion of emo_conv_igemm_bf16x3 with half the matrix work: the SCALED operands (x * in_scale after the producer's
 * norm + ReLU, w * w_scale; powers of two) as the sum of two fp16 terms -- 22+ significand bits, exact to 2^-24 relative for
 * |value| >= 2^-2, to 2^-25 absolute below -- and the three partial products x1 w1 + x1 w2 + x2 w1 (dropped: x2 w2 <= 2^-24),
 * fp32 accumulation, result * 1 / (in_scale * w_scale).  Error against an fp64 convolution: that of a plain fp32 convolution
 * (tools/split_accuracy.py; tests/test_conv_bf16x3_gpu.py).  wpk2: fp16 weights packed like wpk3 with two planes
 * [.. kernel row][plane 1|2][kernel column][half][BM = 64][8].  Otherwise as emo_conv_igemm_bf16x3.
 * CONTRACT, checked on the device: |x * in_scale| saturates at 65504 (in_scale 32: staged inputs beyond +-2047 are clipped).
 *   overflow_flag  NULL (unchecked), or a device word the caller has zeroed: every thread tracks the largest |x * in_scale| it
 *          stages BEFORE the clamp (one v_max3 per two values) and the launch stores 1 there if any exceeded 65504 (+-inf
 *          included; a NaN is staged as the lower clamp bound like everywhere else and does not raise it).  Halo pixels are
 *          checked by every block that stages them; a value that is only ever clamped away by relu_in (below -65504) still
 *          raises the flag -- conservative.
 *   The caller then launches emo_conv_igemm_bf16x3 on the same arguments with run_if = overflow_flag on the same stream: it
 *   recomputes `out` (and gn_stats) only when the flag was raised -- no host synchronisation, hipGraph-capturable; the pair
 *   is what emoportraits_amd.ops.conv_igemm issues for a precision="f16x2" layer
 *   (tests/test_conv_bf16x3_gpu.py::test_conv_f16x2_overflow_is_detected_and_recomputed). */
int emo_conv_igemm_f16x2(const float* x, const void* wpk2, const float* bias,
                         const float* scale, const float* shift, const float* res, float* out,
                         int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW,
                         int ups, int relu_in, int act, int res_ups, int cfg, int ksplit, float* workspace,
                         float* gn_stats, void* stream, float in_scale, float w_scale, int* overflow_flag);

/* ABI 7.  POINTWISE layers on the fp16 split: emo_conv_igemm_f16x2 also takes KH = KW = 1 (KD = 1; 2-D, or 3-D with D as a
 * batch of planes) -- nn.Conv2d(Cin, Cout, 1) of decoder.py:66-70 (1536 -> 512) and the 1x1 skips of utils.py:764-781 -- on a
 * kernel of its own (csrc/conv_igemm_f16x2_p1.h: two 64-channel output tiles per work item on one converted patch).  wpk2 then is
 * [channel tile, padded to an even count][Cin chunk of 32][plane 1|2][k-step of 16][half][BM = 64][8] fp16 of w * w_scale
 * (emoportraits_amd.pack.pack_weight_f16x2_1x1).  Launch form: ksplit 1, ups 0, act EMO_ACT_NONE, Cout % 64 == 0, W % 64 == 0,
 * H % 4 == 0, 16-byte aligned x / out / res; EMO_ERR_UNSUPPORTED otherwise (such a launch runs emo_conv_igemm_f32).  Contract
 * and overflow_flag as above; the guarded exact recomputation behind it is
 * emo_conv_igemm_f32_guarded: emo_conv_igemm_f32 with run_if (NULL, or a device word: the launch does nothing unless
 * *run_if != 0 when it starts executing), same stream, launched right after the fp16-split launch. */
/* ABI 7, also: emo_conv_igemm_f16x2 takes cfg 5 (block config F: 32 channels x 256 positions) for 3x3 / 3x3x3 layers -- a 32-row
 * channel tile for layers with at most 32 output channels (the WarpGenerator's last 3-D block and its 3-channel head,
 * warp_generator_resnet.py:95-123; stage 2's 32-channel ResBlocks), which ran the 64-row tile half empty.  wpk2 is then packed
 * with BM = 32 (emoportraits_amd.pack.pack_weight_f16x2(w, bm=32)); 4 x 64 position tiles (W % 64 == 0, H % 4 == 0), no fused
 * upsample.  The guarded recomputation of such a layer is emo_conv_igemm_bf16x3 with cfg 3 and the BM = 64 weights, as for
 * every 3x3 layer. */
int emo_conv_igemm_f32_guarded(const float* x, const float* wpk, const float* bias,
                               const float* scale, const float* shift, const float* res, float* out,
                               int N, int Cin, int Cout, int D, int H, int W, int KD, int KH, int KW,
                               int ups, int relu_in, int act, int res_ups, int cfg, int ksplit, float* workspace,
                               float* gn_stats, void* stream, const int* run_if);

/* ABI 8.  Pointwise convolution with at most 4 output channels as a stream (the decoder's image head: GroupNorm -> ReLU ->
 * 1x1 conv 128 -> 3 -> sigmoid, decoder.py:381-392): one 16-byte load per (channel, four positions), fp32 FMAs, no LDS.
 *   out[n][o][p] = act(bias[o] + sum_c w[o][c] * in(x[n][c][p])),  in(v) = v * scale[n][c] + shift[n][c], then max(., 0) if relu_in
 *   x [N, Cin, S] (S positions per channel: any spatial rank), w [Cout, Cin] plain row-major fp32 (NOT a packed layout),
 *   bias [Cout] or NULL, scale / shift [N, Cin] both or neither, act EMO_ACT_*.
 * Cout <= 4, S % 4 == 0 (EMO_ERR_UNSUPPORTED otherwise), 16-byte aligned x / out (EMO_ERR_ALIGN): such launches run
 * emo_conv_igemm_f32.  The sum over the channels is sequential in fp32 (the MFMA kernel's is blocked): same bounds against the
 * oracle, not the same bits. */
int emo_conv_head_f32(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                      float* out, int N, int Cin, int Cout, int64_t S, int relu_in, int act, void* stream);

/* ---------------------------------------------------------------------------------------------
 * resampling / pointwise helpers (HBM-bound, one pass)
 *   emo_upsample_trilinear_f32: F.interpolate(x, scale_factor=(fd,fh,fw), mode='trilinear'), factors in {1,2}
 *       (warp_generator_resnet.py:163-166; unet_3d.py:223,269-272).  x [NC, D, H, W] -> [NC, D*fd, H*fh, W*fw]
 *   emo_avgpool_f32: nn.AvgPool2d/3d with kernel == stride in 1..16 per axis (utils.py:962-967; integer-window
 *       AdaptiveAvgPool2d of the embedders)
 *   emo_add_f32: out[i] = (a[i] + b[i % period]) * alpha   (unet_3d.py:281; va.py:857)
 */
int emo_upsample_trilinear_f32(const float* x, float* out, int64_t NC, int D, int H, int W,
                               int fd, int fh, int fw, void* stream);
/* ABI 8.  emo_upsample_trilinear_f32 with the GroupNorm statistics of its OUTPUT reduced on the way (WarpGenerator:
 * F.interpolate -> ResBlock3d, whose first norm would otherwise read the upsampled tensor once more;
 * warp_generator_resnet.py:163-166, utils.py:711-731).  x [N, C, D, H, W]; `partial` (>= emo_groupnorm_workspace_bytes(N, G)
 * bytes) receives *split_out slices per (sample, group) for emo_groupnorm_affine_from_sums_f32.  fw = 2, W even and a 16-byte
 * aligned `out` only: EMO_ERR_UNSUPPORTED otherwise (run the two operations one after the other). */
int emo_upsample_trilinear_gn_sums_f32(const float* x, float* out, int N, int C, int G, int D, int H, int W,
                                       int fd, int fh, int fw, void* partial, int64_t partial_bytes, int* split_out,
                                       void* stream);
int emo_avgpool_f32(const float* x, float* out, int64_t NC, int D, int H, int W, int kd, int kh, int kw, void* stream);
int emo_add_f32(const float* a, const float* b, float* out, int64_t n, int64_t period, float alpha, void* stream);

/* f4 -- F.interpolate(x, size=(Ho, Wo), mode='bilinear' (bicubic = 0) | 'bicubic' (1), align_corners=False) on NC
 * planes of H x W: the wrappers' crop resize (notebooks/infer.py:346,399-401,548-552; notebooks/infer_s2.py:360-362).
 * The input is strided (plane_stride / row_stride in floats; H*W / W when contiguous) so that a crop window of a larger
 * frame is read in place; clamp01 clips the result to [0,1] (crop_image, infer.py:350). */
int emo_resize2d_f32(const float* x, int64_t plane_stride, int64_t row_stride, float* out, int64_t NC, int H, int W,
                     int Ho, int Wo, int bicubic, int clamp01, void* stream);
/* ABI 9.  The same with one crop window PER SAMPLE in one launch (the reference crops every frame around its own face box,
 * notebooks/infer.py:301-352, one frame per call): x = N frames of C planes (plane_stride / row_stride in floats), windows =
 * N x (x0, y0, w, h) int32 in DEVICE memory, out [N, C, Ho, Wo].  Bit-identical to N single-window calls. */
int emo_resize2d_windows_f32(const float* x, int64_t plane_stride, int64_t row_stride, const int* windows, float* out, int N,
                             int C, int Ho, int Wo, int bicubic, int clamp01, void* stream);

/* ---------------------------------------------------------------------------------------------
 * f1 -- embedder ResNets (networks/volumetric_avatar/identity_embedder.py:59-69, expression_embedder.py:424-459,
 * head_pose_regressor.py:21-32; bodies = torchvision.models.resnet*).
 *
 * emo_conv2d_generic_f32: F.conv2d(x', w, bias, stride, padding) with x' = relu?(x * scale[n,c] + shift[n,c]) when
 *   scale/shift are given (the producer's norm + ReLU, applied before zero padding), any KH x KW / stride / pad.
 *   wt is the folded weight transposed to [Cin*KH*KW][CoutP] (k = (ci*KH + ky)*KW + kx, CoutP = Cout rounded up to 64,
 *   zero padded).  x [N,Cin,H,W], out [N,Cout,Ho,Wo].  splits > 1 divides K over gridDim.z through
 *   workspace [splits][N*Cout*Ho*Wo] floats, summed in fixed order (deterministic); emo_conv2d_generic_splits returns
 *   the split count the launch heuristic wants for a shape (>= 1; the kernel accepts any value >= 1).
 * emo_maxpool2d_f32:   nn.MaxPool2d(k, stride, pad) of relu?(x * scale + shift) (scale/shift [NC] or NULL).
 * emo_affine_add_relu_f32: out = relu?((a*sa+ta) + (b*sb+tb)), per-(n,c) affines optional, b optional: the tail of
 *   BasicBlock / Bottleneck.forward.
 * emo_grid_sample2d_f32: F.grid_sample(img, grid) 4-D bilinear / zeros / align_corners=False; either an explicit
 *   grid [N,Ho,Wo,2] or theta [N,2,3] + lin [Ho] (grid = theta @ (lin[xo], lin[yo], 1): expression_embedder.py:221-231);
 *   grid_out (optional) receives the grid that was used. */
int emo_conv2d_generic_f32(const float* x, const float* wt, const float* bias, const float* scale, const float* shift,
                           float* out, int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                           int relu_in, int splits, float* workspace, void* stream);
int emo_conv2d_generic_splits(int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad);
int emo_maxpool2d_f32(const float* x, const float* scale, const float* shift, float* out, int64_t NC, int H, int W,
                      int k, int stride, int pad, int relu, void* stream);
int emo_affine_add_relu_f32(const float* a, const float* sa, const float* ta, const float* b, const float* sb,
                            const float* tb, float* out, int64_t NC, int64_t S, int relu, void* stream);
int emo_grid_sample2d_f32(const float* img, const float* grid, const float* theta, const float* lin, float* out,
                          float* grid_out, int N, int C, int H, int W, int Ho, int Wo, void* stream);

/* stage-2 glue (notebooks/infer_s2.py:365-375):
 *   emo_mul_mask_f32:        out[n,c,p] = img[n,c,p] * mask[n,0,p]            (local_encoder input, :370)
 *   emo_stage2_compose_f32:  out = clamp(img + add * (mask * face_mask), 0, 1)   (:365,373-375); masks are [N,1,H,W] */
int emo_mul_mask_f32(const float* img, const float* mask, float* out, int N, int C, int64_t HW, void* stream);
int emo_stage2_compose_f32(const float* img, const float* add, const float* mask, const float* face_mask, float* out,
                           int N, int C, int64_t HW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * a3/a4/a5 -- small wave-reduced kernels
 *   emo_small_gemm_f32: C[b][m][0..NN) = sum_k A[m][k] * B[b][k][0..NN), NN in {1,2,4,16}: pose_unsqueeze_nw Linear
 *       (va.py:172-173,820-823), warp_embed_head_orig_nw 1x1 conv on 4x4 (va.py:177-181,857), WarpGenerator.first_conv
 *       (warp_generator_resnet.py:70,141), ProjectorNorm u @ embed (utils.py:1137-1151).
 *   emo_projector_finalize_f32: (u@embed) @ v -> (d_gamma, d_beta), then ada_gamma = gamma + d_gamma, ada_beta = beta +
 *       d_beta (utils.py:1146-1149 + assign_adaptive_norm_params :983-995) for all adaptive norms of a net at once:
 *       T [B, R, E], V [n_norms, E, 2], norm_of_row [R] int32, gamma/beta [R] -> ada_gamma/ada_beta [B, R].
 *   emo_pose_theta_f32: utils/point_transforms.py:188-242 get_transform_matrix -> theta [B,4,4]; scale [B,scale_cols].
 *   emo_pack_rgb8: notebooks/infer.py:641-644 clamp(0,1) + ToPILImage: [N,3,H,W] fp32 -> [N,H,W,3] uint8.
 *   emo_unpack_rgb8: notebooks/infer.py:211-223 convert_to_tensor (ToTensor) of decoded video frames:
 *       [N,H,W,3] uint8 -> [N,3,H,W] fp32 = byte / 255, on the device (frames are uploaded as bytes: 4x less PCIe).
 */
int emo_small_gemm_f32(const float* A, const float* B, float* C, int M, int K, int NN, int batch,
                       int64_t b_stride, int64_t c_stride, void* stream);
int emo_projector_finalize_f32(const float* T, const float* V, const int* norm_of_row, const float* gamma,
                               const float* beta, float* ada_gamma, float* ada_beta, int B, int R, int E, void* stream);
int emo_pose_theta_f32(const float* scale, int scale_cols, const float* rotation, const float* translation,
                       float* theta, int B, void* stream);
/* inverse of B row-major 4x4 matrices (`theta.float().inverse()`: notebooks/infer.py:443, expression_embedder.py:185-188) */
int emo_mat4_inverse_f32(const float* in, float* out, int B, void* stream);
int emo_pack_rgb8(const float* img, uint8_t* out, int N, int H, int W, void* stream);
int emo_unpack_rgb8(const uint8_t* in, float* out, int N, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMO_HIP_H_ */
