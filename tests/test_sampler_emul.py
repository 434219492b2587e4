"""The LDS-staged tile sampler (emoportraits_amd/csrc/gs3d_tile.h) executed on the CPU, phase by phase and thread by thread
(tests/emul), against the plain-C oracle (oracle/grid_sample3d.c, itself pinned to torch's CPU F.grid_sample by
tests/test_oracle.py): bit-exact for every coordinate source, padding mode, layout and tile configuration, including the
paths a friendly input never takes (brick-by-brick staging, direct fallback, zero border, dead voxels, ragged lattices)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
import c_oracle  # noqa: E402
import emul  # noqa: E402


def ident(D, H, W):
    lz, ly, lx = emul.lattice(D), emul.lattice(H), emul.lattice(W)
    w, v, u = np.meshgrid(lz, ly, lx, indexing="ij")
    return np.stack([u, v, w], -1)[None].astype(np.float32)


def theta_of(rng, n, angle=0.3, scale=0.1, shift=0.05):
    out = []
    for _ in range(n):
        a, b, c = rng.uniform(-angle, angle, 3)
        rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        m = np.diag(rng.uniform(1 - scale, 1 + scale, 3)) @ rz @ ry @ rx
        out.append(np.concatenate([m, rng.uniform(-shift, shift, (3, 1))], 1))
    return np.stack(out).astype(np.float32)


def same_bits(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


LAYOUTS = [(True, True), (True, False), (False, False)]


@pytest.mark.parametrize("in_p4,out_p4", LAYOUTS)
@pytest.mark.parametrize("pad", ["zeros", "border", "reflection"])
def test_near_identity_warp_all_sources(in_p4, out_p4, pad):
    rng = np.random.default_rng(1)
    C, D, H, W = 8, 8, 24, 32
    vol = rng.standard_normal((2, C, D, H, W)).astype(np.float32)
    warp = (ident(D, H, W) + 0.08 * np.tanh(rng.standard_normal((2, D, H, W, 3)))).astype(np.float32)
    ref = c_oracle.grid_sample3d(vol, warp, pad)
    out, st = emul.run(vol, grid=warp, pad=pad, in_p4=in_p4, out_p4=out_p4, tile=(3, 3, 2), upb=2)
    assert same_bits(out, ref) and st["direct_passes"] == 0
    delta = np.ascontiguousarray((warp - ident(D, H, W)).transpose(0, 4, 1, 2, 3))
    ref_d = c_oracle.grid_sample3d(vol, (ident(D, H, W) + delta.transpose(0, 2, 3, 4, 1)).astype(np.float32), pad)
    out, st = emul.run(vol, delta=delta, pad=pad, in_p4=in_p4, out_p4=out_p4, tile=(4, 3, 2), upb=1)
    assert same_bits(out, ref_d)
    th = theta_of(rng, 2)
    gt = c_oracle.affine_grid3d(th, emul.lattice(W), emul.lattice(H), emul.lattice(D))
    out, st = emul.run(vol, theta=th, pad=pad, in_p4=in_p4, out_p4=out_p4, tile=(2, 3, 3), upb=2)
    assert same_bits(out, c_oracle.grid_sample3d(vol, gt, pad))


@pytest.mark.parametrize("in_p4,out_p4", LAYOUTS)
def test_ragged_lattice_other_output_size_and_shared_volume(in_p4, out_p4):
    """output lattice not a multiple of the tile (partial tiles), output size != volume size, one volume for 3 samples"""
    rng = np.random.default_rng(2)
    C, D, H, W = 4, 5, 13, 20
    vol = rng.standard_normal((1, C, D, H, W)).astype(np.float32)
    Do, Ho, Wo = 7, 11, 19
    grid = rng.uniform(-1.2, 1.2, (3, Do, Ho, Wo, 3)).astype(np.float32)      # wild: direct / brick passes
    ref = c_oracle.grid_sample3d(vol, grid)
    for tile, threads in (((3, 3, 2), 256), ((3, 3, 3), 256), ((4, 3, 3), 512)):
        out, st = emul.run(vol, grid=grid, in_p4=in_p4, out_p4=out_p4, tile=tile, threads=threads, upb=1, cap_slots=700)
        assert same_bits(out, ref), (tile, st)
    smooth = (ident(Do, Ho, Wo) * 0.9 + 0.03 * rng.standard_normal((3, Do, Ho, Wo, 3))).astype(np.float32)
    out, st = emul.run(vol, grid=smooth, in_p4=in_p4, out_p4=out_p4, tile=(3, 3, 3), upb=1)
    assert same_bits(out, c_oracle.grid_sample3d(vol, smooth)) and st["staged_passes"] > 0


@pytest.mark.parametrize("in_p4,out_p4", LAYOUTS)
def test_small_stage_forces_brick_passes_and_direct(in_p4, out_p4):
    """the same input with stages too small for the union box, then too small for a brick: all three paths agree"""
    rng = np.random.default_rng(3)
    C, D, H, W = 8, 8, 32, 32
    vol = rng.standard_normal((1, C, D, H, W)).astype(np.float32)
    th = theta_of(rng, 2, angle=0.5)
    ref = c_oracle.grid_sample3d(vol, c_oracle.affine_grid3d(th, emul.lattice(W), emul.lattice(H), emul.lattice(D)))
    seen = set()
    for cap in (4000, 1100, 500, 40):
        out, st = emul.run(vol, theta=th, in_p4=in_p4, out_p4=out_p4, tile=(3, 3, 3), upb=2, cap_slots=cap)
        assert same_bits(out, ref), (cap, st)
        seen.add((st["union_blocks"] == st["blocks"], st["direct_passes"] > 0, st["staged_passes"] > 0))
    assert len(seen) >= 3, seen        # all-union, mixed, all-direct were exercised


def test_zero_border_dead_voxels_and_non_finite_coordinates():
    rng = np.random.default_rng(4)
    C, D, H, W = 4, 4, 16, 16
    vol = rng.standard_normal((1, C, D, H, W)).astype(np.float32)
    vol[0, :, 0, 0, 0] = np.inf               # a dead voxel must give exactly 0 even next to non-finite data
    grid = (ident(D, H, W) * 1.3).astype(np.float32)                          # a third of the lattice falls outside the volume
    grid[0, 1, 2, 3] = np.nan
    grid[0, 2, 5, 7] = (np.inf, 0.0, 0.0)
    grid[0, 3, 9, 1] = (3e38, -3e38, 0.5)
    grid[0, 0, 0, 0] = (-1.0 - 2.0 / W, -1.0, -1.0)                           # floor corner exactly -1 / 0 boundaries
    grid[0, 0, 0, 1] = (1.0, 1.0, 1.0)
    ref = c_oracle.grid_sample3d(vol, grid)
    for in_p4, out_p4 in LAYOUTS:
        out, st = emul.run(vol, grid=grid, in_p4=in_p4, out_p4=out_p4, tile=(3, 3, 2), upb=1)
        assert same_bits(np.nan_to_num(out, nan=7.0, posinf=8.0, neginf=9.0), np.nan_to_num(ref, nan=7.0, posinf=8.0, neginf=9.0))
        assert np.all(out[0, :, 1, 2, 3] == 0) and np.all(out[0, :, 2, 5, 7] == 0)


def test_bench_shape_staging_statistics():
    """the driver-pass shapes [96,16,64,64] (2 channel quads here): the rotation call of SURVEY 8(d) stages every tile; the
    staged bytes stay below 4x the output (the L1 sees that instead of the 8x of a direct gather)"""
    rng = np.random.default_rng(5)
    C, D, H, W = 8, 16, 64, 64
    vol = rng.standard_normal((1, C, D, H, W)).astype(np.float32)
    th = theta_of(rng, 2)
    ref = c_oracle.grid_sample3d(vol, c_oracle.affine_grid3d(th, emul.lattice(W), emul.lattice(H), emul.lattice(D)))
    out, st = emul.run(vol, theta=th, in_p4=True, out_p4=False, tile=(3, 3, 2), upb=2, cap_slots=2552)
    assert same_bits(out, ref)
    assert st["direct_passes"] == 0
    assert st["slots_filled"] / (2 * (C // 4) * D * H * W) < 4.0
