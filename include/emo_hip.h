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

#define EMO_ABI_VERSION 1

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

/* activation applied in a kernel epilogue */
#define EMO_ACT_NONE 0
#define EMO_ACT_RELU 1
#define EMO_ACT_TANH 2
#define EMO_ACT_SIGMOID 3

int emo_abi_version(void);
/* human-readable build string: arch, compiler, kernel variants */
const char* emo_build_info(void);

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
 *         with theta, ignored otherwise.
 *   out   [N, C, Do, Ho, Wo] or [N, Do, Ho, Wo, C] according to out_layout.
 *   vol_batch_stride  elements between consecutive volumes (0 = shared volume).
 * NDHWC paths require C % 4 == 0.
 */
int emo_grid_sample3d_f32(const float* vol, const float* grid, const float* theta,
                          const float* lin_x, const float* lin_y, const float* lin_z,
                          float* out,
                          int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                          int64_t vol_batch_stride, int padding_mode,
                          int in_layout, int out_layout, int variant, void* stream);

/* NCDHW <-> NDHWC repack of a 5-D volume (used once per identity on the cached canonical volume,
 * notebooks/infer.py:507 `self.target_latent_volume`).  to_channels_last != 0: NCDHW -> NDHWC. */
int emo_volume_repack_f32(const float* in, float* out, int N, int C, int DHW, int to_channels_last, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EMO_HIP_H_ */
