"""Host-side wrappers of the HIP kernels: torch tensors in, torch tensors out, raw pointers underneath.

Names and argument meaning follow the torch ops the reference calls on the hot path, so that
the parity tests read like `ours(x) == torch_cpu(x)`.
"""
import ctypes
import functools
import os

import torch

from . import hip
from . import pack as pack_mod


@functools.lru_cache(maxsize=None)
def _lattice(n, device_index):
    """torch.linspace(-1, 1, n): the identity lattice of models/stage_1/volumetric_avatar/va.py:101-105.
    Computed by torch on the CPU so that the values are the reference's, then kept resident on the device."""
    return torch.linspace(-1, 1, n).to(torch.device("cuda", device_index))


def grid_sample3d(vol, grid=None, theta=None, padding_mode="zeros", in_layout="ncdhw", out_layout="ncdhw",
                  variant=0, out=None, delta=None):
    """5-D trilinear grid_sample, align_corners=False  (== F.grid_sample(vol, grid, padding_mode=...)).

    vol    [Nv,C,D,H,W] ('ncdhw'), [Nv,C/4,D,H,W,4] ('p4': packed channel quads, the layout of the LDS-staged tile kernels,
           out_layout 'p4' or 'ncdhw') or [Nv,D,H,W,C] ('ndhwc': channels-last, the driver pass's layout; out_layout 'ndhwc'
           or 'ncdhw'); Nv == N or 1 (volume shared by all N samples).
    variant  'p4' input (always the LDS-staged tile kernels): tile_variant(...) tuning word, 0 = defaults;
             'ncdhw' -> 'ncdhw': TILE | tile_variant(...) selects the LDS-staged planar kernel instead of the direct gather.
    grid   [N,Do,Ho,Wo,3]; or None with theta [N,3,4] / [N,4,4]: the sampling grid is then the head-pose affine of
           the identity lattice (notebooks/infer.py:583-588), generated inside the kernel, output size = D,H,W.
    delta  [N,3,Do,Ho,Wo] planar deltas: grid = identity lattice + delta (WarpGenerator output,
           warp_generator_resnet.py:178) without materialising the grid.
    """
    lib = hip.load()
    hip.require_cuda_f32(vol, grid, delta)
    if theta is not None:
        theta = theta.float().contiguous()      # e.g. torch.linalg.inv returns a column-major result
        hip.require_cuda_f32(theta)
    layouts = {"ncdhw": hip.LAYOUT_NCDHW, "ndhwc": hip.LAYOUT_NDHWC, "p4": hip.LAYOUT_P4}
    if in_layout not in layouts or out_layout not in layouts:
        raise ValueError("layouts are 'ncdhw', 'ndhwc' or 'p4'")
    if in_layout == "p4":
        Nv, Q4, D, H, W, four = vol.shape
        if four != 4:
            raise ValueError("a 'p4' volume is [N, C/4, D, H, W, 4]")
        C = 4 * Q4
    elif in_layout == "ndhwc":
        Nv, D, H, W, C = vol.shape
    else:
        Nv, C, D, H, W = vol.shape
    lx = ly = lz = None
    grid_kind = 0
    idx = vol.device.index if vol.device.index is not None else torch.cuda.current_device()
    if delta is not None:
        if grid is not None or theta is not None:
            raise ValueError("pass exactly one of grid / theta / delta")
        if delta.dim() != 5 or delta.shape[1] != 3:
            raise ValueError("delta must be [N,3,Do,Ho,Wo]")
        N, _, Do, Ho, Wo = delta.shape
        lx, ly, lz = _lattice(Wo, idx), _lattice(Ho, idx), _lattice(Do, idx)
        grid, grid_kind = delta, 1
    elif theta is not None:
        if grid is not None:
            raise ValueError("pass exactly one of grid / theta / delta")
        if theta.dim() != 3 or theta.shape[1] not in (3, 4) or theta.shape[2] != 4:
            raise ValueError("theta must be [N,3,4] or [N,4,4]")
        theta = theta[:, :3].contiguous()
        N = theta.shape[0]
        Do, Ho, Wo = D, H, W
        lx, ly, lz = _lattice(Wo, idx), _lattice(Ho, idx), _lattice(Do, idx)
    else:
        if grid is None or grid.dim() != 5 or grid.shape[-1] != 3:
            raise ValueError("grid must be [N,Do,Ho,Wo,3]")
        N, Do, Ho, Wo, _ = grid.shape
    if Nv not in (1, N):
        raise ValueError(f"volume batch {Nv} does not match grid batch {N}")
    stride = 0 if (Nv == 1 and N > 1) else C * D * H * W
    if (in_layout == "ncdhw" and out_layout == "ncdhw" and variant == 0 and C % 4 == 0 and C >= 16
            and Nv * C * D * H * W >= (1 << 20)
            # limits of the channels-last kernels the redirect lands on (csrc/grid_sample3d.hip): 64 tap records + C rows of 65
            # floats in <= 64 KiB of LDS, 32-bit byte offsets inside one volume.  Beyond them the direct NCDHW gather runs
            and C * 260 + 5120 <= 65536 and C * D * H * W * 4 < 2 ** 32):
        # the reference's call shape (model.grid_sample(NCDHW, grid) -> NCDHW, va.py:264-265) on a large volume: one repack
        # to channels-last + the channels-last gather (NCDHW out) moves 51 MB in 23 us, the NCDHW gather needs 35 us
        # (its 4-byte corner loads are bound by the vector-memory instruction rate; archive/profiles/r3_sampler_seam.jsonl)
        return grid_sample3d(volume_to_channels_last(vol), grid if delta is None else None, theta, padding_mode, "ndhwc",
                             "ncdhw", 0, out, delta)
    shape = {"ndhwc": (N, Do, Ho, Wo, C), "ncdhw": (N, C, Do, Ho, Wo), "p4": (N, C // 4, Do, Ho, Wo, 4)}[out_layout]
    if out is None:
        out = torch.empty(shape, device=vol.device, dtype=torch.float32)
    else:
        hip.require_cuda_f32(out)
        if tuple(out.shape) != shape:
            raise ValueError("bad out shape")
    rc = lib.emo_grid_sample3d_f32(hip.ptr(vol), hip.ptr(grid), hip.ptr(theta), hip.ptr(lx), hip.ptr(ly), hip.ptr(lz),
                                   hip.ptr(out), N, C, D, H, W, Do, Ho, Wo, stride, hip.PAD_MODES[padding_mode],
                                   layouts[in_layout], layouts[out_layout], int(variant), grid_kind, hip.current_stream())
    hip.check(rc, "emo_grid_sample3d_f32")
    return out


def affine_grid3d(theta, size):
    """identity_grid_3d.bmm(theta[:, :3].transpose(1, 2)).view(N, d, s, s, 3) (notebooks/infer.py:441-444, :583-588): the
    rotation warp the theta variant of grid_sample3d generates in-kernel, as a tensor.  size = (d, h, w)."""
    lib = hip.load()
    theta = theta.float()[:, :3].contiguous()
    hip.require_cuda_f32(theta)
    if theta.dim() != 3 or theta.shape[1:] != (3, 4):
        raise ValueError("theta must be [N,3,4] or [N,4,4]")
    N = theta.shape[0]
    D, H, W = size
    idx = theta.device.index if theta.device.index is not None else torch.cuda.current_device()
    grid = torch.empty((N, D, H, W, 3), device=theta.device, dtype=torch.float32)
    hip.check(lib.emo_affine_grid3d_f32(hip.ptr(theta), hip.ptr(_lattice(W, idx)), hip.ptr(_lattice(H, idx)),
                                        hip.ptr(_lattice(D, idx)), hip.ptr(grid), N, D, H, W, hip.current_stream()),
              "emo_affine_grid3d_f32")
    return grid


def volume_to_channels_last(vol):
    """[N,C,D,H,W] -> [N,D,H,W,C] (one pass through a 64x64 LDS tile)."""
    lib = hip.load()
    hip.require_cuda_f32(vol)
    N, C, D, H, W = vol.shape
    out = torch.empty((N, D, H, W, C), device=vol.device, dtype=torch.float32)
    hip.check(lib.emo_volume_repack_f32(hip.ptr(vol), hip.ptr(out), N, C, D * H * W, 1, hip.current_stream()),
              "emo_volume_repack_f32")
    return out


TILE = 1 << 30      # grid_sample3d(variant=TILE | tuning): NCDHW -> NCDHW through the LDS-staged planar kernel


def tile_variant(tile=None, units_per_block=0, lds_kib=0, threads=256):
    """tuning word of the LDS-staged sampler (include/emo_hip.h): tile = (tx, ty, tz) output voxels per block, powers of two
    with tx * ty * tz in {threads, 2 * threads}; 0 / None = the kernel's default"""
    v = 0
    if tile is not None:
        tx, ty, tz = (int(t).bit_length() - 1 for t in tile)
        v |= tx | (ty << 4) | (tz << 8)
    v |= (int(units_per_block) & 31) << 12
    v |= (int(lds_kib) & 255) << 17
    if threads == 512:
        v |= 1 << 25
    return v


def volume_to_p4(vol):
    """[N,C,D,H,W] -> [N,C/4,D,H,W,4] (EMO_LAYOUT_P4: a voxel's channel quad is one 16-byte slot)"""
    lib = hip.load()
    hip.require_cuda_f32(vol)
    N, C, D, H, W = vol.shape
    out = torch.empty((N, C // 4, D, H, W, 4), device=vol.device, dtype=torch.float32)
    hip.check(lib.emo_volume_repack_f32(hip.ptr(vol), hip.ptr(out), N, C, D * H * W, 4, hip.current_stream()),
              "emo_volume_repack_f32")
    return out


def volume_from_p4(vol):
    """[N,C/4,D,H,W,4] -> [N,C,D,H,W]"""
    lib = hip.load()
    hip.require_cuda_f32(vol)
    N, Q, D, H, W, _ = vol.shape
    out = torch.empty((N, 4 * Q, D, H, W), device=vol.device, dtype=torch.float32)
    hip.check(lib.emo_volume_repack_f32(hip.ptr(vol), hip.ptr(out), N, 4 * Q, D * H * W, 5, hip.current_stream()),
              "emo_volume_repack_f32")
    return out


def volume_to_channels_first(vol):
    """[N,D,H,W,C] -> [N,C,D,H,W]"""
    lib = hip.load()
    hip.require_cuda_f32(vol)
    N, D, H, W, C = vol.shape
    out = torch.empty((N, C, D, H, W), device=vol.device, dtype=torch.float32)
    hip.check(lib.emo_volume_repack_f32(hip.ptr(vol), hip.ptr(out), N, C, D * H * W, 0, hip.current_stream()),
              "emo_volume_repack_f32")
    return out


# ----------------------------------------------------------------------------------------------------------------
# GroupNorm -> per-(sample, channel) affine
# ----------------------------------------------------------------------------------------------------------------
def _gn_workspace(N, G, device):
    """per-call scratch for the split partial sums (stream-ordered through torch's caching allocator, so concurrent
    streams never share it)"""
    lib = hip.load()
    need = lib.emo_groupnorm_workspace_bytes(N, G)
    return torch.empty(need, dtype=torch.uint8, device=device), need


def _ada_views(ada_gamma, ada_beta, N, C):
    ada_stride = 0
    if ada_gamma is not None:
        for t in (ada_gamma, ada_beta):
            if not t.is_cuda or t.dtype != torch.float32 or t.stride(-1) != 1 or t.shape != (N, C):
                raise RuntimeError("ada_gamma/ada_beta must be float32 cuda [N,C] with unit inner stride")
        if ada_gamma.stride(0) != ada_beta.stride(0):
            raise RuntimeError("ada_gamma/ada_beta must share the row stride")
        ada_stride = ada_gamma.stride(0)
    return ada_stride


def groupnorm_affine(x, gamma=None, beta=None, ada_gamma=None, ada_beta=None, groups=32, eps=1e-5, want_stats=False,
                     stats=None):
    """scale, shift [N,C] such that GroupNorm(x) == x*scale + shift (per sample and channel).
    ada_gamma / ada_beta: [N, C] views (row stride arbitrary, unit column stride) of the adaptive weights.
    stats: the TileStats the producing conv_igemm(..., want_stats=True) returned for x -- the statistics are then
    combined from the tiles and x itself is not read; or the RunSums of upsample_trilinear(..., gn_groups=)."""
    lib = hip.load()
    hip.require_cuda_f32(x, gamma, beta)
    N, C = x.shape[0], x.shape[1]
    S = x.numel() // (N * C)
    ada_stride = _ada_views(ada_gamma, ada_beta, N, C)
    scale = torch.empty((N, C), device=x.device, dtype=torch.float32)
    shift = torch.empty((N, C), device=x.device, dtype=torch.float32)
    mean = rstd = None
    if want_stats:
        mean = torch.empty((N, groups), device=x.device, dtype=torch.float32)
        rstd = torch.empty((N, groups), device=x.device, dtype=torch.float32)
    if isinstance(stats, RunSums):
        if stats.shape != tuple(x.shape) or stats.groups != groups:
            raise ValueError("run sums do not belong to this tensor")
        rc = lib.emo_groupnorm_affine_from_sums_f32(hip.ptr(stats.partial), stats.split, N, C, S, groups, eps, hip.ptr(gamma),
                                                    hip.ptr(beta), hip.ptr(ada_gamma), hip.ptr(ada_beta), ada_stride,
                                                    hip.ptr(scale), hip.ptr(shift), hip.ptr(mean), hip.ptr(rstd),
                                                    hip.current_stream())
        hip.check(rc, "emo_groupnorm_affine_from_sums_f32")
    elif stats is not None:
        T = stats.stats.shape[1]
        if tuple(stats.stats.shape) != (N, T, C, 2) or T * stats.cnt != S:
            raise ValueError("tile statistics do not belong to this tensor")
        rc = lib.emo_groupnorm_affine_from_tiles_f32(hip.ptr(stats.stats), N, C, T, stats.cnt, groups, eps, hip.ptr(gamma),
                                                     hip.ptr(beta), hip.ptr(ada_gamma), hip.ptr(ada_beta), ada_stride,
                                                     hip.ptr(scale), hip.ptr(shift), hip.ptr(mean), hip.ptr(rstd),
                                                     hip.current_stream())
        hip.check(rc, "emo_groupnorm_affine_from_tiles_f32")
    else:
        ws, need = _gn_workspace(N, groups, x.device)
        rc = lib.emo_groupnorm_affine_f32(hip.ptr(x), N, C, S, groups, eps, hip.ptr(gamma), hip.ptr(beta),
                                          hip.ptr(ada_gamma), hip.ptr(ada_beta), ada_stride, hip.ptr(scale),
                                          hip.ptr(shift), hip.ptr(mean), hip.ptr(rstd), hip.ptr(ws), ws.numel(),
                                          hip.current_stream())
        hip.check(rc, "emo_groupnorm_affine_f32")
    if want_stats:
        return scale, shift, mean, rstd
    return scale, shift


# ----------------------------------------------------------------------------------------------------------------
# implicit-GEMM convolution
# ----------------------------------------------------------------------------------------------------------------
class RunSums:
    """fp64 (sum, sum of squares) slices per (sample, group) of a tensor, in the GroupNorm workspace layout
    ([N * G][64][2] doubles), written by the kernel that produced the tensor (upsample_trilinear(..., gn_groups=))."""
    __slots__ = ("partial", "split", "shape", "groups")

    def __init__(self, partial, split, shape, groups):
        self.partial, self.split, self.shape, self.groups = partial, split, tuple(shape), groups


class TileStats:
    """Per-tile GroupNorm statistics written by the conv epilogue: stats [N, T, C, 2] = (mean, centred sum of squares)
    of `cnt` output values per (sample, tile, channel).  Handed to groupnorm_affine(..., stats=) instead of the tensor."""
    __slots__ = ("stats", "cnt")

    def __init__(self, stats, cnt):
        self.stats, self.cnt = stats, cnt


def _shares_storage(a, b):
    """True when the two tensors overlap in memory (same allocation and intersecting byte ranges)"""
    if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return False
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a0 < b0 + b.numel() * b.element_size() and b0 < a0 + a.numel() * a.element_size()


def conv_igemm(x, layer, scale=None, shift=None, relu_in=False, ups=False, res=None, res_ups=False, act="none",
               out=None, ksplit=None, want_stats=False):
    """layer: emoportraits_amd.pack.PackedConv.  x [N,Cin,H,W] or [N,Cin,D,H,W].
    ksplit: K-loop split of the launch (None: pack.plan_launch decides together with the block config).
    want_stats: also return the TileStats of the output (None when this launch cannot produce them: K-split launches)
    -> (out, stats)."""
    lib = hip.load()
    hip.require_cuda_f32(x, scale, shift, res)
    three_d = x.dim() == 5
    if three_d:
        N, Cin, D, H, W = x.shape
    else:
        N, Cin, H, W = x.shape
        D = 1
    if Cin != layer.cin:
        raise ValueError(f"conv expects {layer.cin} input channels, got {Cin}")
    if (layer.kd == 3) != three_d and layer.kd == 3:
        raise ValueError("3x3x3 conv needs a 5-D input")
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    shape = (N, layer.cout, D, Hl, Wl) if three_d else (N, layer.cout, Hl, Wl)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != shape:
        raise ValueError("bad out shape")
    elif _shares_storage(out, x):
        raise ValueError("out overlaps x: tiles read their neighbours' input halo while others write")
    if res is not None:
        want = (N, layer.cout, D, Hl // 2, Wl // 2) if res_ups else (N, layer.cout, D, Hl, Wl)
        if res.numel() != want[0] * want[1] * want[2] * want[3] * want[4]:
            raise ValueError("bad residual shape")
    positions = N * D * Hl * Wl
    cfg, ks, prec = layer.plan_for(max(1, -(-positions // 128)), Hl, Wl, ups, affine=scale is not None,
                                   # (the pointwise split kernel has the straight-line epilogue only: 16-byte aligned out / res)
                                   aligned16=x.data_ptr() % 16 == 0 and (not getattr(layer, "pointwise_split", False) or (
                                       (res is None or res.data_ptr() % 16 == 0) and (out is None or out.data_ptr() % 16 == 0))),
                                   in_elems_per_sample=x.numel() // max(1, N), act=act,
                                   io_aligned16=(res is None or res.data_ptr() % 16 == 0) and out.data_ptr() % 16 == 0)
    pointwise_split = prec == "f16x2" and getattr(layer, "pointwise_split", False)
    if ksplit is not None:
        ks = int(ksplit)
    ws = torch.empty((ks, out.numel()), device=x.device, dtype=torch.float32) if ks > 1 else None
    stats = None
    # (a pointwise layer on the fp16 split keeps its tile statistics in the 128-position layout of the fp32 MFMA kernel that
    # recomputes it behind a raised overflow word: csrc/conv_igemm_f16x2_p1.h writes two half entries per 256-position tile)
    bp = 128 if pointwise_split else pack_mod._BP[cfg]
    if want_stats and ks == 1 and (D * Hl * Wl) % bp == 0:
        stats = TileStats(torch.empty((N, D * Hl * Wl // bp, layer.cout, 2), device=x.device, dtype=torch.float32), bp)
    layer.last_plan = (cfg, ks, prec)        # which kernel ran (bench.py meters the kernels separately)
    entry = {"f32": lib.emo_conv_igemm_f32, "f16": lib.emo_conv_igemm_f16acc32, "bf16x3": lib.emo_conv_igemm_bf16x3,
             "f16x2": lib.emo_conv_igemm_f16x2, "f16w8": lib.emo_conv_igemm_f16w8}[prec]
    wpk = layer.packed(cfg, prec)
    common = (hip.ptr(layer.bias), hip.ptr(scale), hip.ptr(shift), hip.ptr(res), hip.ptr(out), N, Cin, layer.cout, D, H, W,
              layer.kd, layer.kh, layer.kw, int(ups), int(relu_in), hip.ACT[act], int(res_ups), cfg, ks, hip.ptr(ws),
              hip.ptr(stats.stats) if stats is not None else None, hip.current_stream())
    if prec == "f16x2":
        # the fp16 split checks its operand range on the device (overflow word of the layer); the guarded bf16x3 launch behind
        # it recomputes the layer with exact operands when the word is raised -- no host synchronisation, graph-capturable
        flag = pack_mod.overflow_flag_ptr(x.device, layer.flag_slot) if F16X2_GUARD else None
        caller_out = None
        if F16X2_GUARD and res is not None and _shares_storage(out, res):
            # the guarded launch re-reads res AFTER the first launch has written out: an output that aliases it (a residual
            # updated in place) would feed the first launch's result into the recomputation -- conv + clipped conv + res.
            # Both launches write a private buffer; the caller's `out` receives the result behind them, so that `out=` means
            # the same in every precision mode and guard state
            caller_out = out
            out = torch.empty(shape, device=x.device, dtype=torch.float32)
            common = common[:4] + (hip.ptr(out),) + common[5:]
        gplan = None
        if F16X2_GUARD and pointwise_split:
            # (pointwise layer: the exact recomputation is the fp32 MFMA kernel at its own launch plan -- made BEFORE the fp16-split
            # launch is enqueued, and without a K split when tile statistics travel with the output: a split launch cannot
            # produce them, and raising behind the first launch would leave `out` / `stats` un-recomputed with the word raised)
            gcfg, gks = pack_mod.plan_launch(layer.cout, layer.cin, layer.kd, layer.kh, layer.kw, max(1, -(-positions // 128)),
                                             layer.allowed, "f32")
            if stats is not None:
                gks = 1
            gplan = (gcfg, gks, layer.packed(gcfg))
        rc = entry(hip.ptr(x), hip.ptr(wpk), *common, pack_mod.F16X2_IN_SCALE, layer.w_scale, flag)
        hip.check(rc, f"emo_conv_igemm_f16x2[{layer.name}]")
        if gplan is not None:
            gcfg, gks, gw = gplan
            gws = torch.empty((gks, out.numel()), device=x.device, dtype=torch.float32) if gks > 1 else None
            gcommon = common[:18] + (gcfg, gks, hip.ptr(gws), None if gks > 1 else common[21], common[22])
            rc = lib.emo_conv_igemm_f32_guarded(hip.ptr(x), hip.ptr(gw), *gcommon, flag)
            hip.check(rc, f"emo_conv_igemm_f32_guarded[{layer.name}]")
        elif F16X2_GUARD:
            # (the exact recomputation runs the 64-row tile of the bf16 split whatever tile the fp16-split launch used)
            gcommon = common[:18] + (pack_mod.CFG_D,) + common[19:]
            rc = lib.emo_conv_igemm_bf16x3(hip.ptr(x), hip.ptr(layer.packed(pack_mod.CFG_D, "bf16x3")), *gcommon, flag)
            hip.check(rc, f"emo_conv_igemm_bf16x3[{layer.name}, guarded]")
        if caller_out is not None:
            caller_out.copy_(out)
            out = caller_out
    elif prec == "f16w8":
        # plain fp16 operands on the eight-wave two-tile kernel (opt-in precision 'f16'; csrc/conv_igemm_f16x2_w8.h, NPROD = 1);
        # the straight-line epilogue reads 16-byte aligned out / res: anything else was planned onto the older fp16 kernel
        if pack_mod.f16w8_rest_fits(layer.cout, Hl, Wl):
            # an odd tile count: the pairs on that kernel, the last tile on the older fp16-operand kernel (ABI 10) -- no half-empty pair
            rc = lib.emo_conv_igemm_f16w8_rest(hip.ptr(x), hip.ptr(wpk), hip.ptr(layer.packed(pack_mod.CFG_D, "f16")), *common,
                                               layer.w_scale16)
            hip.check(rc, f"emo_conv_igemm_f16w8_rest[{layer.name}]")
        else:
            rc = entry(hip.ptr(x), hip.ptr(wpk), *common, layer.w_scale16)
            hip.check(rc, f"emo_conv_igemm_f16w8[{layer.name}]")
    else:
        extra = (None,) if prec == "bf16x3" else ()
        rc = entry(hip.ptr(x), hip.ptr(wpk), *common, *extra)
        hip.check(rc, f"emo_conv_igemm_{prec}[{layer.name}]")
    return (out, stats) if want_stats else out


def conv_head(x, layer, scale=None, shift=None, relu_in=False, act="none"):
    """A 1x1(x1) convolution with at most 4 output channels as a stream (csrc/conv_head.hip; the decoder's image head):
    same arguments and result as conv_igemm(x, layer, scale, shift, relu_in=, act=).  Launch forms the stream kernel does not
    take (positions per channel not a multiple of 4, unaligned views, EMO_CONV_HEAD=0) run conv_igemm."""
    lib = hip.load()
    hip.require_cuda_f32(x, scale, shift)
    N, Cin = x.shape[0], x.shape[1]
    S = x.numel() // max(1, N * Cin)
    if (not CONV_HEAD_STREAM or (layer.kd, layer.kh, layer.kw) != (1, 1, 1) or layer.cout > 4 or S % 4 or N > 65535
            or x.data_ptr() % 16 or not x.is_contiguous()):
        return conv_igemm(x, layer, scale, shift, relu_in=relu_in, act=act)
    if Cin != layer.cin:
        raise ValueError(f"conv expects {layer.cin} input channels, got {Cin}")
    out = torch.empty((N, layer.cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
    layer.last_plan = ("head", 1, "stream")
    hip.check(lib.emo_conv_head_f32(hip.ptr(x), hip.ptr(layer.plain_weight()), hip.ptr(layer.bias), hip.ptr(scale), hip.ptr(shift),
                                    hip.ptr(out), N, Cin, layer.cout, S, int(relu_in), hip.ACT[act], hip.current_stream()),
              f"emo_conv_head_f32[{layer.name}]")
    return out


# A/B switch (measurements only): 0 runs the image head on the fp32 MFMA implicit-GEMM kernel, as round 4 did
CONV_HEAD_STREAM = os.environ.get("EMO_CONV_HEAD", "1") != "0"


# A/B switch (measurements only): 0 launches the fp16 split without its device-side range check and guarded recomputation
F16X2_GUARD = pack_mod.F16X2_GUARD_DEFAULT
clear_overflow_flags = pack_mod.clear_overflow_flags
overflow_events = pack_mod.overflow_events


# ----------------------------------------------------------------------------------------------------------------
# resampling / pointwise
# ----------------------------------------------------------------------------------------------------------------
def upsample_trilinear(x, factors, gn_groups=None):
    """F.interpolate(x, scale_factor=factors, mode='trilinear') for 5-D x, factors in {1,2}^3.
    gn_groups: also return the RunSums of the OUTPUT for a GroupNorm of that many groups (reduced by the upsampling kernel
    from the values it writes) -> (out, sums); sums is None where the fused kernel does not apply (width factor 1, odd width)."""
    lib = hip.load()
    hip.require_cuda_f32(x)
    N, C, D, H, W = x.shape
    fd, fh, fw = factors
    out = torch.empty((N, C, D * fd, H * fh, W * fw), device=x.device, dtype=torch.float32)
    if gn_groups is not None and FUSE_UPSAMPLE_STATS and fw == 2 and W % 2 == 0 and C % gn_groups == 0:
        ws, need = _gn_workspace(N, gn_groups, x.device)
        split = ctypes.c_int(0)
        hip.check(lib.emo_upsample_trilinear_gn_sums_f32(hip.ptr(x), hip.ptr(out), N, C, gn_groups, D, H, W, fd, fh, fw,
                                                         hip.ptr(ws), need, ctypes.byref(split), hip.current_stream()),
                  "emo_upsample_trilinear_gn_sums_f32")
        return out, RunSums(ws, split.value, out.shape, gn_groups)
    hip.check(lib.emo_upsample_trilinear_f32(hip.ptr(x), hip.ptr(out), N * C, D, H, W, fd, fh, fw, hip.current_stream()),
              "emo_upsample_trilinear_f32")
    return (out, None) if gn_groups is not None else out


# A/B switch (measurements only): 0 reduces the statistics of an upsampled tensor with a pass of its own, as round 4 did
FUSE_UPSAMPLE_STATS = os.environ.get("EMO_FUSE_UPSAMPLE_STATS", "1") != "0"


def avgpool(x, kernel):
    """nn.AvgPool3d(kernel, stride=kernel) (5-D, kernel=(kd,kh,kw)) or nn.AvgPool2d (4-D, kernel=(kh,kw))"""
    lib = hip.load()
    hip.require_cuda_f32(x)
    if x.dim() == 5:
        N, C, D, H, W = x.shape
        kd, kh, kw = kernel
        oshape = (N, C, D // kd, H // kh, W // kw)
    else:
        N, C, H, W = x.shape
        D, kd = 1, 1
        kh, kw = kernel
        oshape = (N, C, H // kh, W // kw)
    out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    hip.check(lib.emo_avgpool_f32(hip.ptr(x), hip.ptr(out), N * C, D, H, W, kd, kh, kw, hip.current_stream()),
              "emo_avgpool_f32")
    return out


def add(a, b, alpha=1.0, out=None):
    """(a + b) * alpha; b is broadcast over the leading dimension when it is smaller (period = b.numel())"""
    lib = hip.load()
    hip.require_cuda_f32(a, b)
    if a.numel() % b.numel():
        raise ValueError("b does not tile a")
    if out is None:
        out = torch.empty_like(a)
    hip.check(lib.emo_add_f32(hip.ptr(a), hip.ptr(b), hip.ptr(out), a.numel(), b.numel(), float(alpha), hip.current_stream()),
              "emo_add_f32")
    return out


def small_gemm(A, B, NN):
    """C[b][m][:NN] = sum_k A[m][k] * B[b][k][:NN];  A [M,K], B [batch,K,NN] -> [batch,M,NN]"""
    lib = hip.load()
    hip.require_cuda_f32(A, B)
    M, K = A.shape
    batch = B.shape[0]
    if B.numel() != batch * K * NN:
        raise ValueError("bad B shape")
    C = torch.empty((batch, M, NN), device=A.device, dtype=torch.float32)
    hip.check(lib.emo_small_gemm_f32(hip.ptr(A), hip.ptr(B), hip.ptr(C), M, K, NN, batch, K * NN, M * NN, hip.current_stream()),
              "emo_small_gemm_f32")
    return C


def projector_finalize(T, V, norm_of_row, gamma, beta):
    """T [B,R,E], V [n,E,2], norm_of_row [R] int32, gamma/beta [R] -> ada_gamma, ada_beta [B,R]"""
    lib = hip.load()
    hip.require_cuda_f32(T, V, gamma, beta)
    B, R, E = T.shape
    ag = torch.empty((B, R), device=T.device, dtype=torch.float32)
    ab = torch.empty((B, R), device=T.device, dtype=torch.float32)
    hip.check(lib.emo_projector_finalize_f32(hip.ptr(T), hip.ptr(V), hip.ptr(norm_of_row), hip.ptr(gamma), hip.ptr(beta),
                                             hip.ptr(ag), hip.ptr(ab), B, R, E, hip.current_stream()),
              "emo_projector_finalize_f32")
    return ag, ab


def pose_theta(scale, rotation, translation):
    """utils/point_transforms.py:188-242 get_transform_matrix on the device -> [B,4,4]"""
    lib = hip.load()
    hip.require_cuda_f32(scale, rotation, translation)
    B = scale.shape[0]
    theta = torch.empty((B, 4, 4), device=scale.device, dtype=torch.float32)
    hip.check(lib.emo_pose_theta_f32(hip.ptr(scale), scale.shape[1], hip.ptr(rotation), hip.ptr(translation),
                                     hip.ptr(theta), B, hip.current_stream()), "emo_pose_theta_f32")
    return theta


def pack_rgb8(img):
    """[N,3,H,W] fp32 -> [N,H,W,3] uint8 = clamp(0,1)*255 truncated (notebooks/infer.py:641-644 + ToPILImage)"""
    lib = hip.load()
    hip.require_cuda_f32(img)
    N, C, H, W = img.shape
    if C != 3:
        raise ValueError("expected 3 channels")
    out = torch.empty((N, H, W, 3), device=img.device, dtype=torch.uint8)
    hip.check(lib.emo_pack_rgb8(hip.ptr(img), hip.ptr(out), N, H, W, hip.current_stream()), "emo_pack_rgb8")
    return out


def unpack_rgb8(frames_u8):
    """[N,H,W,3] uint8 (decoded video frames) -> [N,3,H,W] fp32 in [0,1] = byte / 255 (notebooks/infer.py:211-223)"""
    lib = hip.load()
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or not frames_u8.is_contiguous():
        raise RuntimeError("unpack_rgb8 expects a contiguous uint8 cuda tensor [N,H,W,3]")
    N, H, W, C = frames_u8.shape
    if C != 3:
        raise ValueError("expected 3 channels")
    out = torch.empty((N, 3, H, W), device=frames_u8.device, dtype=torch.float32)
    hip.check(lib.emo_unpack_rgb8(hip.ptr(frames_u8), hip.ptr(out), N, H, W, hip.current_stream()), "emo_unpack_rgb8")
    return out


def mul_mask(img, mask):
    """img [N,C,H,W] * mask [N,1,H,W]  (notebooks/infer_s2.py:370)"""
    lib = hip.load()
    hip.require_cuda_f32(img, mask)
    N, C, H, W = img.shape
    if mask.numel() != N * H * W:
        raise ValueError("mask must be [N,1,H,W]")
    out = torch.empty_like(img)
    hip.check(lib.emo_mul_mask_f32(hip.ptr(img), hip.ptr(mask), hip.ptr(out), N, C, H * W, hip.current_stream()),
              "emo_mul_mask_f32")
    return out


def stage2_compose(img, add_img, mask, face_mask):
    """clamp(img + add * (mask * face_mask), 0, 1)  (notebooks/infer_s2.py:365,373-375)"""
    lib = hip.load()
    hip.require_cuda_f32(img, add_img, mask, face_mask)
    N, C, H, W = img.shape
    if mask.numel() != N * H * W or face_mask.numel() != N * H * W or add_img.shape != img.shape:
        raise ValueError("bad shapes")
    out = torch.empty_like(img)
    hip.check(lib.emo_stage2_compose_f32(hip.ptr(img), hip.ptr(add_img), hip.ptr(mask), hip.ptr(face_mask), hip.ptr(out),
                                         N, C, H * W, hip.current_stream()), "emo_stage2_compose_f32")
    return out


def resize2d(x, size, mode="bilinear", window=None, clamp01=False):
    """F.interpolate(x[..., y0:y0+h, x0:x0+w], size=size, mode=mode, align_corners=False) for 4-D x; mode 'bilinear' or
    'bicubic'; window = (x0, y0, w, h) reads a crop of the frame in place (default: the whole frame)"""
    lib = hip.load()
    hip.require_cuda_f32(x)
    N, C, H, W = x.shape
    x0, y0, w, h = window if window is not None else (0, 0, W, H)
    if not (0 <= x0 and 0 <= y0 and w > 0 and h > 0 and x0 + w <= W and y0 + h <= H):
        raise ValueError(f"resize window {(x0, y0, w, h)} is not inside the {W}x{H} frame")
    Ho, Wo = size
    out = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
    first = ctypes.c_void_p(x.data_ptr() + 4 * (y0 * W + x0))
    hip.check(lib.emo_resize2d_f32(first, H * W, W, hip.ptr(out), N * C, h, w, Ho, Wo,
                                   {"bilinear": 0, "bicubic": 1}[mode], int(clamp01), hip.current_stream()),
              "emo_resize2d_f32")
    return out


def resize2d_windows(x, size, windows, mode="bicubic", clamp01=False):
    """torch.cat([F.interpolate(x[i:i+1, :, y0:y0+h, x0:x0+w], size=size, mode=mode, align_corners=False) for i ...]) in ONE
    launch (notebooks/infer.py:301-352 crops every frame around its own face box): windows = one (x0, y0, w, h) per frame --
    a host sequence (uploaded here: 16 bytes per frame) or an int32 [N,4] device tensor.  Bit-identical to resize2d per frame."""
    lib = hip.load()
    hip.require_cuda_f32(x)
    N, C, H, W = x.shape
    if isinstance(windows, torch.Tensor):
        win = windows
        if not win.is_cuda or win.dtype != torch.int32 or tuple(win.shape) != (N, 4) or not win.is_contiguous():
            raise RuntimeError("windows must be a contiguous int32 cuda tensor [N,4]")
    else:
        host = torch.tensor([[int(v) for v in w] for w in windows], dtype=torch.int32).reshape(-1, 4)
        if host.shape[0] != N:
            raise ValueError(f"{host.shape[0]} windows for {N} frames")
        lo, hi = host[:, :2], host[:, :2] + host[:, 2:]
        if not (bool((lo >= 0).all()) and bool((host[:, 2:] > 0).all()) and bool((hi[:, 0] <= W).all()) and bool((hi[:, 1] <= H).all())):
            raise ValueError(f"a resize window is not inside the {W}x{H} frame")
        win = host.to(x.device, non_blocking=True)
    Ho, Wo = size
    out = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
    hip.check(lib.emo_resize2d_windows_f32(hip.ptr(x), H * W, W, hip.ptr(win), hip.ptr(out), N, C, Ho, Wo,
                                           {"bilinear": 0, "bicubic": 1}[mode], int(clamp01), hip.current_stream()),
              "emo_resize2d_windows_f32")
    return out


def device_cu_count():
    """compute units of the current device as the C launchers count them (include/emo_hip.h, ABI 9)"""
    return hip.load().emo_device_cu_count()


def mfma_stream(iters=2000, lds_reads=False, sink=None):
    """one launch of the bare fp16 MFMA stream (diagnostic: bench.py `roofline.sustained_peak`) -> MFMA instructions issued"""
    lib = hip.load()
    if sink is None:
        sink = torch.empty(256 * device_cu_count(), device="cuda", dtype=torch.float32)
    n = ctypes.c_int64(0)
    hip.check(lib.emo_mfma_stream_f16(hip.ptr(sink), int(iters), int(bool(lds_reads)), ctypes.byref(n), hip.current_stream()),
              "emo_mfma_stream_f16")
    return n.value


# ---- embedder ResNets -----------------------------------------------------------------------
def conv2d_generic(x, wt, cout, kh, kw, stride, pad, bias=None, scale=None, shift=None, relu_in=False, splits=None):
    """F.conv2d(relu?(x*scale+shift), w, bias, stride, pad); wt = pack.pack_generic(w) [Cin*kh*kw, CoutP].
    splits: K split count (None = the library's launch heuristic)"""
    lib = hip.load()
    hip.require_cuda_f32(x)
    N, Cin, H, W = x.shape
    if wt.shape[0] != Cin * kh * kw:
        raise ValueError(f"packed weight has K={wt.shape[0]}, input needs {Cin * kh * kw}")
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.empty((N, cout, Ho, Wo), device=x.device, dtype=torch.float32)
    if splits is None:
        splits = lib.emo_conv2d_generic_splits(N, Cin, H, W, cout, kh, kw, stride, pad)
        if splits < 1:
            hip.check(splits, "emo_conv2d_generic_splits")
    ws = torch.empty((splits, out.numel()), device=x.device, dtype=torch.float32) if splits > 1 else None
    hip.check(lib.emo_conv2d_generic_f32(hip.ptr(x), hip.ptr(wt), hip.ptr(bias), hip.ptr(scale), hip.ptr(shift),
                                         hip.ptr(out), N, Cin, H, W, cout, kh, kw, stride, pad, int(relu_in), splits,
                                         hip.ptr(ws), hip.current_stream()), "emo_conv2d_generic_f32")
    return out


def maxpool2d(x, k, stride, pad, scale=None, shift=None, relu=False):
    lib = hip.load()
    hip.require_cuda_f32(x)
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
    hip.check(lib.emo_maxpool2d_f32(hip.ptr(x), hip.ptr(scale), hip.ptr(shift), hip.ptr(out), N * C, H, W, k, stride,
                                    pad, int(relu), hip.current_stream()), "emo_maxpool2d_f32")
    return out


def affine_add_relu(a, sa=None, ta=None, b=None, sb=None, tb=None, relu=True):
    """relu?((a*sa+ta) + (b*sb+tb)) with per-(n,c) affines"""
    lib = hip.load()
    hip.require_cuda_f32(a)
    N, C = a.shape[:2]
    S = a.numel() // (N * C)
    if b is not None and b.shape != a.shape:
        raise ValueError("affine_add_relu: shape mismatch")
    out = torch.empty_like(a)
    hip.check(lib.emo_affine_add_relu_f32(hip.ptr(a), hip.ptr(sa), hip.ptr(ta), hip.ptr(b), hip.ptr(sb), hip.ptr(tb),
                                          hip.ptr(out), N * C, S, int(relu), hip.current_stream()),
              "emo_affine_add_relu_f32")
    return out


def grid_sample2d(img, grid=None, theta=None, size=None, want_grid=False):
    """F.grid_sample(img, grid) (bilinear, zeros, align_corners=False); or theta [N,2,3] on the square
    linspace(-1,1,size) lattice of ExpressionEmbed"""
    lib = hip.load()
    hip.require_cuda_f32(img)
    N, C, H, W = img.shape
    lin = None
    if grid is not None:
        Ho, Wo = grid.shape[1:3]
    else:
        Ho = Wo = int(size)
        lin = _lattice(Ho, img.device.index)
        theta = theta.float().contiguous()
    out = torch.empty((N, C, Ho, Wo), device=img.device, dtype=torch.float32)
    gout = torch.empty((N, Ho, Wo, 2), device=img.device, dtype=torch.float32) if want_grid else None
    hip.check(lib.emo_grid_sample2d_f32(hip.ptr(img), hip.ptr(grid), hip.ptr(theta), hip.ptr(lin), hip.ptr(out),
                                        hip.ptr(gout), N, C, H, W, Ho, Wo, hip.current_stream()), "emo_grid_sample2d_f32")
    return (out, gout) if want_grid else out


def mat4_inverse(m):
    """[B,4,4] -> inverse, on the device (no host LAPACK round trip)"""
    lib = hip.load()
    hip.require_cuda_f32(m)
    if m.shape[1:] != (4, 4):
        raise ValueError("mat4_inverse expects [B,4,4]")
    out = torch.empty_like(m)
    hip.check(lib.emo_mat4_inverse_f32(hip.ptr(m), hip.ptr(out), m.shape[0], hip.current_stream()), "emo_mat4_inverse_f32")
    return out
