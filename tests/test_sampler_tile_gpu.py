"""GPU parity of the LDS-staged tile sampler (csrc/gs3d_tile.h; SURVEY.md section 8 rows a1 + a2) through the C ABI, on the
driver pass's own call pair at the released shape: shared canonical volume + planar deltas -> packed-4 intermediate ->
analytic theta -> NCDHW.  Bar: bit-exact against torch's CPU F.grid_sample on the materialised grids.  (Every other sampler
test -- KATs, ragged shapes, non-finite coordinates, all paddings -- also runs the tile kernels through
tests/test_grid_sample_gpu.py::_all_layouts.)"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import restate as O  # noqa: E402

from emoportraits_amd import ops  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair_inputs(C, D, S, N, seed, delta_amp):
    g = torch.Generator().manual_seed(seed)
    vol = torch.randn(1, C, D, S, S, generator=g)
    delta = torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * delta_amp
    theta = O.get_transform_matrix(1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g),
                                   0.05 * torch.randn(N, 3, generator=g))
    ident = O.identity_grid_3d(D, S)[..., :3].view(1, D, S, S, 3).permute(0, 4, 1, 2, 3)
    warp = (ident + delta).permute(0, 2, 3, 4, 1)
    return vol, delta, theta, warp


@pytest.mark.parametrize("pm", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("case", [(96, 16, 64, 3, 0.03), (96, 16, 64, 2, 0.6), (32, 3, 12, 5, 0.1)])
def test_driver_pair_packed4_bit_exact(pm, case):
    """near-identity warp (|delta| < 1 voxel: every tile staged), wild warp (brick / direct passes), small ragged volume"""
    C, D, S, N, amp = case
    vol, delta, theta, warp = _pair_inputs(C, D, S, N, C + N, amp)
    ref1 = F.grid_sample(vol.expand(N, -1, -1, -1, -1), warp, padding_mode=pm, align_corners=False)
    vp4 = ops.volume_to_p4(vol.to(DEV))
    grid2 = ops.affine_grid3d(theta.to(DEV), (D, S, S)).cpu()
    ref2 = F.grid_sample(ref1, grid2, padding_mode=pm, align_corners=False)
    for variant in (0, ops.tile_variant((16, 8, 4), 6), ops.tile_variant((16, 16, 4), 12, threads=512)):
        mid = ops.grid_sample3d(vp4, delta=delta.to(DEV), padding_mode=pm, in_layout="p4", out_layout="p4", variant=variant)
        assert mid.shape == (N, C // 4, D, S, S, 4)
        assert torch.equal(ops.volume_from_p4(mid).cpu(), ref1)
        out = ops.grid_sample3d(mid, theta=theta.to(DEV), padding_mode=pm, in_layout="p4", out_layout="ncdhw", variant=variant)
        assert torch.equal(out.cpu(), ref2)
    # the planar (reference layout) LDS-staged kernel on the same two calls
    a = ops.grid_sample3d(vol.to(DEV), delta=delta.to(DEV), padding_mode=pm, variant=ops.TILE)
    assert torch.equal(a.cpu(), ref1)
    b = ops.grid_sample3d(a, theta=theta.to(DEV), padding_mode=pm, variant=ops.TILE)
    assert torch.equal(b, out)


def test_tile_kernel_equals_direct_gather_kernels_at_batch_16():
    """size-independent property at the bench batch: the LDS-staged path and the round-1 direct-gather path agree bit for
    bit on 16 frames (too large for the CPU oracle to finish in seconds)"""
    C, D, S, N = 96, 16, 64, 16
    vol, delta, theta, _ = _pair_inputs(C, D, S, N, 7, 0.03)
    v = vol.to(DEV)
    mid_ref = ops.grid_sample3d(v, delta=delta.to(DEV))
    out_ref = ops.grid_sample3d(mid_ref, theta=theta.to(DEV))
    vp4 = ops.volume_to_p4(v)
    for chunk, variant in ((16, 0), (4, ops.tile_variant((16, 8, 4), 24))):
        out = torch.empty_like(out_ref)
        for a in range(0, N, chunk):
            mid = ops.grid_sample3d(vp4, delta=delta[a:a + chunk].to(DEV), in_layout="p4", out_layout="p4", variant=variant)
            assert torch.equal(ops.volume_from_p4(mid), mid_ref[a:a + chunk])
            ops.grid_sample3d(mid, theta=theta[a:a + chunk].to(DEV), in_layout="p4", out_layout="ncdhw", out=out[a:a + chunk],
                              variant=variant)
        assert torch.equal(out, out_ref)


def test_tile_kernel_rejects_what_it_cannot_index():
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):   # packed-4 input only feeds packed-4 or NCDHW output
        ops.grid_sample3d(torch.randn(1, 1, 2, 4, 4, 4, device=DEV), torch.zeros(1, 2, 2, 2, 3, device=DEV), in_layout="p4",
                          out_layout="ndhwc")
    with pytest.raises(RuntimeError, match="BAD_ARG"):       # tile of 128 voxels: fewer than one per thread
        ops.grid_sample3d(torch.randn(1, 1, 2, 4, 4, 4, device=DEV), torch.zeros(1, 2, 2, 2, 3, device=DEV), in_layout="p4",
                          out_layout="p4", variant=ops.TILE | ops.tile_variant((4, 4, 8)))
    # a planar volume whose rows are not 16-byte multiples silently takes the direct-gather kernel (same result)
    v5 = torch.randn(1, 4, 3, 5, 5, device=DEV)
    g5 = torch.rand(1, 3, 5, 5, 3, device=DEV) * 2 - 1
    assert torch.equal(ops.grid_sample3d(v5, g5, variant=ops.TILE), ops.grid_sample3d(v5, g5))
