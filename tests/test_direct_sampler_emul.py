"""The production 3-D sampler of the driver pass (csrc/grid_sample3d.hip: the channels-last direct-gather kernels, the planar
direct gather, the rotation-warp kernel), run on the CPU from the product's own source and compared BIT FOR BIT with ATen's CPU
grid_sample -- SURVEY.md section 8 rows a1 / a2 without a GPU.

The source is compiled as host C++ by ROCm's clang++ (it knows the vector extensions the kernels use) against the stand-in
<hip/hip_runtime.h> of tests/emul/hipshim in its threaded mode: the threads of a block are OS threads, __syncthreads is a barrier.
One line is rewritten on the way -- `extern __shared__ ... float smem[];` becomes a pointer to the stand-in's per-block buffer --
and the two dispatchers of the LDS-staged tile kernels (another translation unit; tests/test_sampler_emul.py runs those) are
stubs that refuse.  Wave votes are evaluated per lane (see hipshim): the branches they select in gather_quad are
result-equivalent per lane.  The product never loads this library.
"""
import ctypes
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "emoportraits_amd", "csrc")
SHIM = os.path.join(HERE, "emul", "hipshim")
BUILD = os.path.join(HERE, "emul", "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
PAD = {"zeros": 0, "border": 1, "reflection": 2}
NCDHW, NDHWC = 0, 1

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not installed")


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, os.path.join(HERE, "emul"))
    import emulibs
    return emulibs.sampler()


def _buf(t):
    """16-byte aligned float32 copy of a tensor / array"""
    a = np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)
    raw = np.empty(a.size + 8, np.float32)
    off = (-(raw.ctypes.data // 4)) % 4
    out = raw[off:off + a.size].reshape(a.shape)
    out[...] = a
    assert out.ctypes.data % 16 == 0
    return out


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def sample(lib, vol, *, grid=None, theta=None, delta=None, pad="zeros", in_layout=NCDHW, out_layout=NCDHW, variant=0, out_size=None):
    """vol: NCDHW tensor [Nv,C,D,H,W] (repacked here for NDHWC input); returns an NCDHW tensor"""
    Nv, C, D, H, W = vol.shape
    v = _buf(vol.permute(0, 2, 3, 4, 1) if in_layout == NDHWC else vol)
    lx = ly = lz = g = th = None
    kind = 0
    if grid is not None:
        N, Do, Ho, Wo, _ = grid.shape
        g = _buf(grid)
    elif delta is not None:
        N, _, Do, Ho, Wo = delta.shape
        g, kind = _buf(delta), 1
    else:
        N, (Do, Ho, Wo) = theta.shape[0], out_size or (D, H, W)
        th = _buf(theta[:, :3, :4])
    if g is None or kind == 1:
        lx, ly, lz = (_buf(torch.linspace(-1, 1, n)) for n in (Wo, Ho, Do))
    out = _buf(np.full((N, Do, Ho, Wo, C) if out_layout == NDHWC else (N, C, Do, Ho, Wo), np.nan, np.float32))
    rc = lib.emo_grid_sample3d_f32(_p(v), _p(g), _p(th), _p(lx), _p(ly), _p(lz), _p(out), N, C, D, H, W, Do, Ho, Wo,
                                   ctypes.c_int64(0 if Nv == 1 and N > 1 else C * D * H * W), PAD[pad], in_layout, out_layout, variant, kind, None)
    assert rc == 0, rc
    o = torch.from_numpy(out.copy())
    return o.permute(0, 4, 1, 2, 3).contiguous() if out_layout == NDHWC else o


def _same_bits(a, b):
    return np.array_equal(a.numpy().view(np.uint32), b.numpy().view(np.uint32))


def _grid(gen, N, Do, Ho, Wo, amp):
    lin = [torch.linspace(-1, 1, n) for n in (Do, Ho, Wo)]
    zz, yy, xx = torch.meshgrid(*lin, indexing="ij")
    base = torch.stack([xx, yy, zz], -1)[None]
    return (base + amp * torch.tanh(torch.randn(N, Do, Ho, Wo, 3, generator=gen))).contiguous()


@pytest.mark.parametrize("pad", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("C,vshape,oshape,N,shared", [(8, (3, 5, 6), (3, 4, 7), 2, False),       # ragged last block (84 voxels)
                                                      (96, (2, 4, 4), (4, 4, 4), 3, True),        # the released channel count, one volume for all
                                                      (4, (4, 4, 8), (4, 8, 8), 1, False)])
def test_channels_last_kernels_bit_exact_with_aten(lib, pad, C, vshape, oshape, N, shared):
    """gs3d_cl_v2_kernel (NDHWC -> NDHWC), gs3d_cl2ncdhw_v2_kernel (NDHWC -> NCDHW through the LDS transpose tile, with and
    without non-temporal stores), gs3d_cl_brick_kernel (4 x 4 x 4 output bricks) on grids that reach 1.6 x outside the volume"""
    gen = torch.Generator().manual_seed(C + N)
    vol = torch.randn(1 if shared else N, C, *vshape, generator=gen)
    grid = _grid(gen, N, *oshape, amp=0.6)
    ref = F.grid_sample(vol.expand(N, -1, -1, -1, -1) if shared else vol, grid, mode="bilinear", padding_mode=pad, align_corners=False)
    assert _same_bits(sample(lib, vol, grid=grid, pad=pad, in_layout=NDHWC, out_layout=NDHWC), ref)
    assert _same_bits(sample(lib, vol, grid=grid, pad=pad, in_layout=NDHWC, out_layout=NCDHW), ref)
    assert _same_bits(sample(lib, vol, grid=grid, pad=pad, in_layout=NDHWC, out_layout=NCDHW, variant=2), ref)
    if all(o % 4 == 0 for o in oshape):
        assert _same_bits(sample(lib, vol, grid=grid, pad=pad, in_layout=NDHWC, out_layout=NDHWC, variant=1), ref)


@pytest.mark.parametrize("pad", ["zeros", "border", "reflection"])
def test_planar_direct_gather_bit_exact_with_aten(lib, pad):
    """gs3d_ncdhw_kernel (the reference's own layout on both sides; an odd channel count, channels per block 1 .. C)"""
    gen = torch.Generator().manual_seed(11)
    vol = torch.randn(2, 5, 3, 6, 5, generator=gen)
    grid = _grid(gen, 2, 2, 5, 9, amp=0.5)
    ref = F.grid_sample(vol, grid, mode="bilinear", padding_mode=pad, align_corners=False)
    for cpb in (0, 1, 2, 5):
        assert _same_bits(sample(lib, vol, grid=grid, pad=pad, variant=cpb), ref), cpb


@pytest.mark.parametrize("pad", ["zeros", "reflection"])
def test_theta_and_delta_modes_equal_the_materialised_grids(lib, pad):
    """a2: the rotation warp generated in-kernel samples what the materialised warp (emo_affine_grid3d_f32, the same arithmetic)
    samples; the WarpGenerator's planar deltas (lattice + delta) sample what the materialised `warp` samples
    (warp_generator_resnet.py:178)"""
    gen = torch.Generator().manual_seed(5)
    N, C, D, H, W = 2, 8, 3, 4, 6
    vol = torch.randn(N, C, D, H, W, generator=gen)
    a = torch.tensor([0.3, -0.2])
    theta = torch.eye(4)[None].repeat(N, 1, 1)
    theta[:, 0, 0], theta[:, 0, 1], theta[:, 1, 0], theta[:, 1, 1] = torch.cos(a), -torch.sin(a), torch.sin(a), torch.cos(a)
    theta[:, :3, 3] = torch.tensor([[0.05, -0.03, 0.02], [-0.04, 0.01, 0.0]])
    th, lx, ly, lz = _buf(theta[:, :3, :4]), _buf(torch.linspace(-1, 1, W)), _buf(torch.linspace(-1, 1, H)), _buf(torch.linspace(-1, 1, D))
    warp = np.empty((N, D, H, W, 3), np.float32)
    assert lib.emo_affine_grid3d_f32(_p(th), _p(lx), _p(ly), _p(lz), _p(warp), N, D, H, W, None) == 0
    bmm = torch.stack(torch.meshgrid(torch.linspace(-1, 1, D), torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")[::-1], -1)
    bmm = torch.cat([bmm, torch.ones(D, H, W, 1)], -1).view(1, -1, 4).expand(N, -1, -1).bmm(theta[:, :3].transpose(1, 2)).view(N, D, H, W, 3)
    assert np.abs(warp - bmm.numpy()).max() <= 2.5 * 2.0 ** -23 * 2.0          # the reference's bmm grid, to its rounding
    ref = F.grid_sample(vol, torch.from_numpy(warp), mode="bilinear", padding_mode=pad, align_corners=False)
    for il, ol in ((NDHWC, NCDHW), (NDHWC, NDHWC), (NCDHW, NCDHW)):
        assert _same_bits(sample(lib, vol, theta=theta, pad=pad, in_layout=il, out_layout=ol), ref)
    delta = 0.1 * torch.tanh(torch.randn(N, 3, D, H, W, generator=gen))
    lin = [torch.linspace(-1, 1, n) for n in (D, H, W)]
    zz, yy, xx = torch.meshgrid(*lin, indexing="ij")
    wgrid = (torch.stack([xx, yy, zz], 0)[None] + delta).permute(0, 2, 3, 4, 1).contiguous()
    ref = F.grid_sample(vol, wgrid, mode="bilinear", padding_mode=pad, align_corners=False)
    for il, ol in ((NDHWC, NCDHW), (NDHWC, NDHWC), (NCDHW, NCDHW)):
        assert _same_bits(sample(lib, vol, delta=delta, pad=pad, in_layout=il, out_layout=ol), ref)


def test_non_finite_coordinates_and_refusals(lib):
    gen = torch.Generator().manual_seed(2)
    vol = torch.randn(1, 4, 2, 3, 4, generator=gen)
    grid = _grid(gen, 1, 2, 3, 4, amp=0.2)
    grid[0, 0, 0, 0, 0], grid[0, 1, 2, 3, 1], grid[0, 0, 1, 1, 2] = float("nan"), float("inf"), float("-inf")
    for pad in ("zeros", "border", "reflection"):
        ref = F.grid_sample(vol, grid, mode="bilinear", padding_mode=pad, align_corners=False)
        got = sample(lib, vol, grid=grid, pad=pad, in_layout=NDHWC, out_layout=NCDHW)
        assert np.array_equal(np.isnan(got.numpy()), np.isnan(ref.numpy()))
        ok = ~np.isnan(ref.numpy())
        assert np.array_equal(got.numpy()[ok].view(np.uint32), ref.numpy()[ok].view(np.uint32))
    v, g, o = _buf(vol), _buf(grid), _buf(np.zeros((1, 4, 2, 3, 4), np.float32))
    call = lambda *a: lib.emo_grid_sample3d_f32(*a)
    assert call(_p(v), None, None, None, None, None, _p(o), 1, 4, 2, 3, 4, 2, 3, 4, ctypes.c_int64(96), 0, 0, 0, 0, 0, None) == -1   # no grid, no theta
    assert call(_p(v), _p(g), None, None, None, None, _p(o), 0, 4, 2, 3, 4, 2, 3, 4, ctypes.c_int64(96), 0, 0, 0, 0, 0, None) == -1  # empty batch
    assert call(_p(v[0, 0, 0, 0, 1:]), _p(g), None, None, None, None, _p(o), 1, 4, 2, 3, 4, 2, 3, 4, ctypes.c_int64(96), 0, 0, 0, 0, 0, None) == -3
