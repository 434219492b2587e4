"""GPU parity tests of the HIP 3-D grid_sample through the C ABI.
Bar: bit-exact against torch's CPU F.grid_sample / the C oracle (integer+fp32 index arithmetic and the
accumulation order are restated exactly)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import c_oracle  # noqa: E402
import restate as O  # noqa: E402

from emoportraits_amd import ops  # noqa: E402

pytestmark = pytest.mark.gpu
PADS = ["zeros", "border", "reflection"]
DEV = "cuda:0"


# (tag, tuning word) of the LDS-staged kernels: default; one voxel per thread; two bricks per thread; 512-thread blocks;
# a 2 KiB stage (union box never fits: brick-by-brick and direct passes); 1 channel unit per block
TILE_TUNINGS = [("", 0), ("_8x8x4", ops.tile_variant((8, 8, 4))), ("_16x8x4", ops.tile_variant((16, 8, 4), units_per_block=5)),
                ("_8x8x16_512", ops.tile_variant((8, 8, 16), threads=512)),
                ("_tiny_stage", ops.tile_variant((8, 8, 8), lds_kib=2)), ("_upb1", ops.tile_variant((4, 8, 8), units_per_block=1, lds_kib=16))]


def _all_layouts(vol_cpu, grid_cpu=None, theta_cpu=None, pm="zeros", shared=False):
    """run every (in_layout, out_layout) combination; return dict name -> NCDHW cpu tensor"""
    vol = vol_cpu.to(DEV)
    grid = None if grid_cpu is None else grid_cpu.to(DEV)
    theta = None if theta_cpu is None else theta_cpu.to(DEV)
    res = {}
    res["ncdhw"] = ops.grid_sample3d(vol, grid, theta, pm).cpu()
    for cpb in (1, 5, 96):
        res[f"ncdhw_cpb{cpb}"] = ops.grid_sample3d(vol, grid, theta, pm, variant=cpb).cpu()
    if vol.shape[1] % 4 == 0:
        vcl = ops.volume_to_channels_last(vol)
        assert torch.equal(vcl.cpu(), vol_cpu.permute(0, 2, 3, 4, 1).contiguous())
        for var in (0, 1):       # default (64-voxel rows) and 4x4x4 output bricks where the lattice allows
            ocl = ops.grid_sample3d(vcl, grid, theta, pm, in_layout="ndhwc", out_layout="ndhwc", variant=var)
            res[f"cl_var{var}"] = ops.volume_to_channels_first(ocl).cpu()
            assert torch.equal(res[f"cl_var{var}"], ocl.cpu().permute(0, 4, 1, 2, 3).contiguous())
        for var in (0, 2):       # NCDHW output: plain and non-temporal stores
            res[f"cl2ncdhw_var{var}"] = ops.grid_sample3d(vcl, grid, theta, pm, in_layout="ndhwc", out_layout="ncdhw", variant=var).cpu()
    if vol.shape[1] % 4 == 0:
        # LDS-staged tile kernels: packed-4 layout [N, C/4, D, H, W, 4] (csrc/gs3d_tile.h), default + forced tunings
        vp4 = ops.volume_to_p4(vol)
        N_, C_, D_, H_, W_ = vol_cpu.shape
        assert torch.equal(vp4.cpu(), vol_cpu.view(N_, C_ // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
        assert torch.equal(ops.volume_from_p4(vp4).cpu(), vol_cpu)
        for tag, var in TILE_TUNINGS:
            o = ops.grid_sample3d(vp4, grid, theta, pm, in_layout="p4", out_layout="p4", variant=ops.TILE | var)
            res["p4_tile" + tag] = ops.volume_from_p4(o).cpu()
            # (P4 -> NCDHW keeps 4 KiB of transposition scratch next to the stage: the 2 KiB tuning is rejected there,
            # test_tile_tuning_word_with_too_little_lds_is_rejected; the tiny stage is 6 KiB for this pair)
            var_n = ops.tile_variant((8, 8, 8), lds_kib=6) if tag == "_tiny_stage" else var
            res["p4_tile_to_ncdhw" + tag] = ops.grid_sample3d(vp4, grid, theta, pm, in_layout="p4", out_layout="ncdhw", variant=ops.TILE | var_n).cpu()
    if vol.shape[4] % 4 == 0:
        for tag, var in TILE_TUNINGS:
            res["ncdhw_tile" + tag] = ops.grid_sample3d(vol, grid, theta, pm, variant=ops.TILE | var).cpu()
    return res


@pytest.mark.parametrize("pm", PADS)
def test_kat_golden_bit_exact(golden_dir, pm):
    k = dict(np.load(os.path.join(golden_dir, "sampler_kat.npz")))
    vol, grid = torch.from_numpy(k["vol"]), torch.from_numpy(k["grid"])
    for name, got in _all_layouts(vol, grid, pm=pm).items():
        assert torch.equal(got, torch.from_numpy(k["out_" + pm])), f"{name} {pm}"
    for name, got in _all_layouts(vol[:1].contiguous(), grid, pm=pm).items():
        assert torch.equal(got, torch.from_numpy(k["out_shared_" + pm])), f"shared {name} {pm}"


@pytest.mark.parametrize("pm", PADS)
@pytest.mark.parametrize("shape", [((1, 8, 3, 5, 7), (2, 4, 6)), ((3, 4, 16, 9, 33), (5, 17, 70)), ((2, 12, 1, 1, 1), (1, 1, 1))])
def test_random_shapes_vs_torch_cpu(pm, shape):
    (N, C, D, H, W), (Do, Ho, Wo) = shape
    g = torch.Generator().manual_seed(N * 100 + C)
    vol = torch.randn(N, C, D, H, W, generator=g)
    grid = torch.rand(N, Do, Ho, Wo, 3, generator=g) * 3 - 1.5
    ref = F.grid_sample(vol, grid, padding_mode=pm, align_corners=False)
    for name, got in _all_layouts(vol, grid, pm=pm).items():
        assert torch.equal(got, ref), f"{name} {pm} {shape}"


@pytest.mark.parametrize("pm", PADS)
def test_fma_accumulation_mode_keeps_the_taps_and_stays_within_the_product_roundings(pm):
    """variant bit 4 of the channels-last kernels (include/emo_hip.h): the index arithmetic is untouched -- a lattice-aligned grid
    (every weight 0 or 1) therefore reproduces ATen bit for bit, which pins the corner selection -- and on a generic grid the
    values differ from ATen's separate multiply / add by at most the eight skipped product roundings"""
    g = torch.Generator().manual_seed(21)
    N, C, D, H, W = 3, 24, 6, 10, 12
    vol = torch.randn(N, C, D, H, W, generator=g)
    vcl = ops.volume_to_channels_last(vol.to(DEV))
    # (1) grid points exactly on voxel centres (unnormalised coordinate integral): one corner has weight 1
    zi, yi, xi = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    exact = torch.stack([(2 * xi + 1) / W - 1, (2 * yi + 1) / H - 1, (2 * zi + 1) / D - 1], -1).float()[None].expand(N, -1, -1, -1, -1).contiguous()
    for out_layout in ("ndhwc", "ncdhw"):
        ref = F.grid_sample(vol, exact, padding_mode=pm, align_corners=False)
        got = ops.grid_sample3d(vcl, exact.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout=out_layout, variant=4)
        got = ops.volume_to_channels_first(got).cpu() if out_layout == "ndhwc" else got.cpu()
        plain = ops.grid_sample3d(vcl, exact.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout="ncdhw").cpu()
        assert torch.equal(plain, ref)
        assert (got - ref).abs().max().item() <= 2.0 ** -22 * vol.abs().max().item()      # weights 1 - eps roundings only
    # (2) generic grid with out-of-range points
    grid = torch.rand(N, 5, 7, 9, 3, generator=g) * 2.6 - 1.3
    ref = F.grid_sample(vol, grid, padding_mode=pm, align_corners=False)
    for out_layout in ("ndhwc", "ncdhw"):
        got = ops.grid_sample3d(vcl, grid.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout=out_layout, variant=4)
        got = ops.volume_to_channels_first(got).cpu() if out_layout == "ndhwc" else got.cpu()
        err = (got - ref).abs().max().item()
        assert err <= 8 * 2.0 ** -24 * vol.abs().max().item(), err
        assert not torch.equal(got, ref) or pm == "zeros"        # (it IS a different rounding sequence)
    # shared volume + analytic head-pose warp (the driver pass's second call)
    th = torch.eye(4)[None, :3].repeat(8, 1, 1) + 0.05 * torch.randn(8, 3, 4, generator=g)
    v1 = ops.volume_to_channels_last(vol[:1].to(DEV))
    a = ops.grid_sample3d(v1, theta=th.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout="ncdhw").cpu()
    b = ops.grid_sample3d(v1, theta=th.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout="ncdhw", variant=4).cpu()
    assert (a - b).abs().max().item() <= 8 * 2.0 ** -24 * vol.abs().max().item()


def test_tile_tuning_word_with_too_little_lds_is_rejected():
    """lds_kib below header + scratch + a minimal stage: EMO_ERR_BAD_ARG instead of an unsigned underflow of the stage size
    (gs3d_tile_launch.h); the same word is fine where no scratch is needed"""
    g = torch.Generator().manual_seed(11)
    vol = torch.randn(1, 8, 4, 8, 8, generator=g).to(DEV)
    grid = (torch.rand(1, 4, 8, 8, 3, generator=g) * 2 - 1).to(DEV)
    vp4 = ops.volume_to_p4(vol)
    small = ops.TILE | ops.tile_variant((8, 8, 4), lds_kib=2)
    with pytest.raises(RuntimeError, match="BAD_ARG"):
        ops.grid_sample3d(vp4, grid, in_layout="p4", out_layout="ncdhw", variant=small)
    ref = F.grid_sample(vol.cpu(), grid.cpu(), align_corners=False)
    assert torch.equal(ops.volume_from_p4(ops.grid_sample3d(vp4, grid, in_layout="p4", out_layout="p4", variant=small)).cpu(), ref)


def test_reference_call_shape_beyond_the_channels_last_limits_takes_the_direct_gather():
    """grid_sample3d(NCDHW, grid) -> NCDHW redirects large volumes through the channels-last kernels; C = 256 exceeds their
    LDS rows (C * 260 + 5120 <= 64 KiB) and must run the direct NCDHW gather instead of raising (ops.py)"""
    g = torch.Generator().manual_seed(12)
    vol = torch.randn(1, 256, 4, 32, 32, generator=g)
    grid = torch.rand(1, 3, 9, 11, 3, generator=g) * 2.2 - 1.1
    got = ops.grid_sample3d(vol.to(DEV), grid.to(DEV)).cpu()
    assert torch.equal(got, F.grid_sample(vol, grid, align_corners=False))


def test_odd_channel_count_ncdhw_only():
    g = torch.Generator().manual_seed(3)
    vol = torch.randn(2, 7, 3, 4, 5, generator=g)
    grid = torch.rand(2, 3, 4, 5, 3, generator=g) * 2.4 - 1.2
    got = ops.grid_sample3d(vol.to(DEV), grid.to(DEV)).cpu()
    assert torch.equal(got, F.grid_sample(vol, grid, align_corners=False))
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.grid_sample3d(vol.permute(0, 2, 3, 4, 1).contiguous().to(DEV), grid.to(DEV), in_layout="ndhwc", out_layout="ndhwc")


def test_non_finite_coordinates_sample_nothing():
    vol = torch.randn(1, 4, 2, 3, 4)
    grid = torch.tensor([float("nan"), 0, 0, float("inf"), 0, 0, 0, -float("inf"), 0, 1e30, 0, 0]).view(1, 1, 1, 4, 3)
    for name, got in _all_layouts(vol, grid).items():
        assert torch.all(got == 0), name


@pytest.mark.parametrize("pm", ["zeros", "reflection"])
def test_full_size_released_shape_bit_exact(pm):
    """[N,96,16,64,64] (the released latent volume), realistic warp = identity + 0.05*tanh(randn) and a hard one."""
    g = torch.Generator().manual_seed(1)
    C, D, S = 96, 16, 64
    vol = torch.randn(1, C, D, S, S, generator=g)
    ident = O.identity_grid_3d(D, S)[..., :3].view(1, D, S, S, 3)
    warps = torch.cat([ident + 0.05 * torch.tanh(torch.randn(1, D, S, S, 3, generator=g)),
                       ident * 1.3 + 0.3 * torch.randn(1, D, S, S, 3, generator=g)])
    ref = torch.from_numpy(c_oracle.grid_sample3d(vol.numpy(), warps.numpy(), pm))   # shared volume, N=2
    torch_ref = F.grid_sample(vol.expand(2, -1, -1, -1, -1), warps, padding_mode=pm, align_corners=False)
    assert torch.equal(ref, torch_ref)
    for name, got in _all_layouts(vol, warps, pm=pm).items():
        assert torch.equal(got, ref), f"{name} {pm}"


@pytest.mark.parametrize("pm", PADS)
def test_analytic_affine_grid_equals_explicit_rotation_warp(golden_dir, pm):
    """a2: theta applied in-kernel to the identity lattice == sampling with the materialised rotation warp."""
    p = dict(np.load(os.path.join(golden_dir, "pose_theta.npz")))
    theta = torch.from_numpy(p["theta"][:4])
    g = torch.Generator().manual_seed(8)
    C, D, S = 8, 16, 64
    vol = torch.randn(1, C, D, S, S, generator=g)
    grid = torch.from_numpy(c_oracle.affine_grid3d(theta[:, :3].numpy(), p["lin_s"], p["lin_s"], p["lin_z"]))
    # the C fma-chain grid equals the reference's bmm grid on the committed sub-lattice up to 1 ulp (tests/test_oracle.py);
    # with the SAME grid values the analytic kernel must be bit-exact
    ref = F.grid_sample(vol.expand(4, -1, -1, -1, -1), grid, padding_mode=pm, align_corners=False)
    for name, got in _all_layouts(vol, theta_cpu=theta, pm=pm).items():
        assert torch.equal(got, ref), f"{name} {pm}"
    # and against the reference's own construction (torch CPU bmm) within fp32 coordinate rounding
    ref_bmm = F.grid_sample(vol.expand(4, -1, -1, -1, -1), O.rotation_warp(theta, D, S), padding_mode=pm, align_corners=False)
    got = ops.grid_sample3d(vol.to(DEV), theta=theta.to(DEV), padding_mode=pm).cpu()
    assert (got - ref_bmm).abs().max().item() <= 1e-4 * ref_bmm.abs().max().item()


def test_size_independent_properties_at_batch_64():
    """BASELINE config 2 size: 64 drivers sharing one canonical volume.  Properties that need no CPU reference:
    (i) shared-volume call == per-sample call, (ii) channel permutation commutes with sampling bit-exactly,
    (iii) the pixel-centre identity grid reproduces the volume exactly, (iv) all layouts agree bit-exactly."""
    g = torch.Generator().manual_seed(64)
    C, D, S, N = 96, 16, 64, 64
    vol = torch.randn(1, C, D, S, S, generator=g).to(DEV)
    ident = O.identity_grid_3d(D, S)[..., :3].view(1, D, S, S, 3)
    grid = (ident + 0.05 * torch.tanh(torch.randn(N, D, S, S, 3, generator=g))).to(DEV)
    out = ops.grid_sample3d(vol, grid)
    one = ops.grid_sample3d(vol, grid[17:18].contiguous())
    assert torch.equal(out[17:18], one)
    perm = torch.randperm(C, generator=g).to(DEV)
    outp = ops.grid_sample3d(vol[:, perm].contiguous(), grid[:4].contiguous())
    assert torch.equal(outp, out[:4][:, perm])
    vcl = ops.volume_to_channels_last(vol)
    ocl = ops.grid_sample3d(vcl, grid, in_layout="ndhwc", out_layout="ndhwc")
    assert torch.equal(ops.volume_to_channels_first(ocl), out)
    assert torch.equal(ops.grid_sample3d(vcl, grid, in_layout="ndhwc", out_layout="ncdhw"), out)
    # with a shared volume and N % 8 == 0 the row-shaped blocks run row-group-major over each XCD's samples; 5 samples take the
    # plain XCD-contiguous order; variant 2: non-temporal output stores; variant 1: brick-shaped blocks
    assert torch.equal(ops.grid_sample3d(vcl, grid, in_layout="ndhwc", out_layout="ncdhw", variant=2), out)
    assert torch.equal(ops.grid_sample3d(vcl, grid[:5].contiguous(), in_layout="ndhwc", out_layout="ncdhw"), out[:5])
    assert torch.equal(ops.volume_to_channels_first(ops.grid_sample3d(vcl, grid[:16].contiguous(), in_layout="ndhwc", out_layout="ndhwc", variant=1)), out[:16])
    # the LDS-staged tile kernels on the same 64-sample batch
    vp4 = ops.volume_to_p4(vol)
    assert torch.equal(ops.grid_sample3d(vp4, grid, in_layout="p4", out_layout="ncdhw"), out)
    zs, ys, xs = [((2 * torch.arange(n) + 1) / n - 1) for n in (D, S, S)]
    zz, yy, xx = torch.meshgrid(zs, ys, xs, indexing="ij")
    centre = torch.stack([xx, yy, zz], -1)[None].to(DEV)
    rec = ops.grid_sample3d(vol, centre)
    assert (rec - vol).abs().max().item() <= 2e-6 * vol.abs().max().item()


@pytest.mark.parametrize("pm", PADS)
def test_delta_grid_mode_equals_materialised_warp(pm):
    """WarpGenerator output consumed as planar deltas: lattice + delta inside the kernel == the reference's
    warp = (identity_grid + deltas).permute(0,2,3,4,1) fed to F.grid_sample"""
    g = torch.Generator().manual_seed(12)
    C, D, S, N = 8, 16, 64, 3
    vol = torch.randn(1, C, D, S, S, generator=g)
    delta = torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.4
    ident = O.identity_grid_3d(D, S)[..., :3].view(1, D, S, S, 3).permute(0, 4, 1, 2, 3)
    warp = (ident + delta).permute(0, 2, 3, 4, 1)
    ref = F.grid_sample(vol.expand(N, -1, -1, -1, -1), warp, padding_mode=pm, align_corners=False)
    v = vol.to(DEV)
    assert torch.equal(ops.grid_sample3d(v, delta=delta.to(DEV), padding_mode=pm).cpu(), ref)
    vp4 = ops.volume_to_p4(v)
    for tag, var in TILE_TUNINGS:
        assert torch.equal(ops.grid_sample3d(v, delta=delta.to(DEV), padding_mode=pm, variant=ops.TILE | var).cpu(), ref), tag
        var_n = ops.tile_variant((8, 8, 8), lds_kib=6) if tag == "_tiny_stage" else var   # (P4 -> NCDHW needs its 4 KiB scratch)
        assert torch.equal(ops.grid_sample3d(vp4, delta=delta.to(DEV), padding_mode=pm, in_layout="p4", out_layout="ncdhw", variant=ops.TILE | var_n).cpu(), ref), tag
        o = ops.grid_sample3d(vp4, delta=delta.to(DEV), padding_mode=pm, in_layout="p4", out_layout="p4", variant=ops.TILE | var)
        assert torch.equal(ops.volume_from_p4(o).cpu(), ref), tag
    vcl = ops.volume_to_channels_last(v)
    assert torch.equal(ops.grid_sample3d(vcl, delta=delta.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout="ncdhw").cpu(), ref)
    o = ops.grid_sample3d(vcl, delta=delta.to(DEV), padding_mode=pm, in_layout="ndhwc", out_layout="ndhwc")
    assert torch.equal(o.cpu().permute(0, 4, 1, 2, 3), ref)


def test_empty_batch_is_rejected():
    vol = torch.randn(1, 4, 2, 2, 2, device=DEV)
    with pytest.raises(RuntimeError, match="BAD_ARG"):
        ops.grid_sample3d(vol, torch.zeros(1, 0, 1, 1, 3, device=DEV))
