"""Host-side wrappers of the HIP kernels: torch tensors in, torch tensors out, raw pointers underneath.

Names and argument meaning follow the torch ops the reference calls on the hot path, so that
the parity tests read like `ours(x) == torch_cpu(x)`.
"""
import functools

import torch

from . import hip


@functools.lru_cache(maxsize=None)
def _lattice(n, device_index):
    """torch.linspace(-1, 1, n): the identity lattice of models/stage_1/volumetric_avatar/va.py:101-105.
    Computed by torch on the CPU so that the values are the reference's, then kept resident on the device."""
    return torch.linspace(-1, 1, n).to(torch.device("cuda", device_index))


def grid_sample3d(vol, grid=None, theta=None, padding_mode="zeros", in_layout="ncdhw", out_layout="ncdhw",
                  batch=None, variant=0, out=None):
    """5-D trilinear grid_sample, align_corners=False  (== F.grid_sample(vol, grid, padding_mode=...)).

    vol    [Nv,C,D,H,W] ('ncdhw') or [Nv,D,H,W,C] ('ndhwc'); Nv == N or 1 (volume shared by all N samples).
    grid   [N,Do,Ho,Wo,3]; or None with theta [N,3,4] / [N,4,4]: the sampling grid is then the head-pose affine of
           the identity lattice (notebooks/infer.py:583-588), generated inside the kernel, output size = D,H,W.
    """
    lib = hip.load()
    hip.require_cuda_f32(vol, grid, theta)
    cl_in = in_layout == "ndhwc"
    cl_out = out_layout == "ndhwc"
    if cl_in:
        Nv, D, H, W, C = vol.shape
    else:
        Nv, C, D, H, W = vol.shape
    lx = ly = lz = None
    if theta is not None:
        if grid is not None:
            raise ValueError("pass either grid or theta")
        if theta.dim() != 3 or theta.shape[1] not in (3, 4) or theta.shape[2] != 4:
            raise ValueError("theta must be [N,3,4] or [N,4,4]")
        theta = theta[:, :3].contiguous()
        N = theta.shape[0]
        Do, Ho, Wo = D, H, W
        idx = vol.device.index if vol.device.index is not None else torch.cuda.current_device()
        lx, ly, lz = _lattice(Wo, idx), _lattice(Ho, idx), _lattice(Do, idx)
    else:
        if grid is None or grid.dim() != 5 or grid.shape[-1] != 3:
            raise ValueError("grid must be [N,Do,Ho,Wo,3]")
        N, Do, Ho, Wo, _ = grid.shape
    if Nv not in (1, N):
        raise ValueError(f"volume batch {Nv} does not match grid batch {N}")
    stride = 0 if (Nv == 1 and N > 1) else C * D * H * W
    shape = (N, Do, Ho, Wo, C) if cl_out else (N, C, Do, Ho, Wo)
    if out is None:
        out = torch.empty(shape, device=vol.device, dtype=torch.float32)
    else:
        hip.require_cuda_f32(out)
        if tuple(out.shape) != shape:
            raise ValueError("bad out shape")
    rc = lib.emo_grid_sample3d_f32(hip.ptr(vol), hip.ptr(grid), hip.ptr(theta), hip.ptr(lx), hip.ptr(ly), hip.ptr(lz),
                                   hip.ptr(out), N, C, D, H, W, Do, Ho, Wo, stride, hip.PAD_MODES[padding_mode],
                                   int(cl_in), int(cl_out), int(variant), hip.current_stream())
    hip.check(rc, "emo_grid_sample3d_f32")
    return out


def volume_to_channels_last(vol):
    """[N,C,D,H,W] -> [N,D,H,W,C] (one pass through a 64x64 LDS tile)."""
    lib = hip.load()
    hip.require_cuda_f32(vol)
    N, C, D, H, W = vol.shape
    out = torch.empty((N, D, H, W, C), device=vol.device, dtype=torch.float32)
    hip.check(lib.emo_volume_repack_f32(hip.ptr(vol), hip.ptr(out), N, C, D * H * W, 1, hip.current_stream()),
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
