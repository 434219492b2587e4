"""GPU parity tests of the embedder path: the generic conv / pooling / alignment kernels against
torch CPU fp32, and IdtEmbed / HeadPoseRegressor / ExpressionEmbed against the oracle (oracle/restate.py) and the golden
outputs of the reference's own classes (tests/golden/embedders.pt, made by oracle/make_golden.py).

Tolerances, relative to max|reference| of each tensor:
  conv kernel        2e-5   exact-fp32 MFMA, different summation order than the CPU library (K up to 4608)
  alignment sampler  2e-6   same formulas as ATen's vectorised CPU kernel, FMA contraction of the CPU build unknown
  embedder outputs   2e-4   20 (resnet18) / 53 (resnet50) conv + GroupNorm layers in sequence
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import restate as O  # noqa: E402

from emoportraits_amd import embedders as E  # noqa: E402
from emoportraits_amd import ops, pack  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, ref):
    return (got.cpu().double() - ref.double()).abs().max().item() / (ref.double().abs().max().item() + 1e-30)


@pytest.mark.parametrize("case", [
    # N, Cin, H, W, Cout, k, stride, pad, affine, bias
    (2, 3, 128, 128, 64, 7, 2, 3, True, False),     # ResNet stem (ImageNet normalisation as the input affine)
    (3, 64, 32, 32, 128, 3, 2, 1, True, False),     # BasicBlock.conv1 of a down-sampling stage
    (3, 64, 32, 32, 128, 1, 2, 0, False, False),    # its downsample conv
    (5, 256, 8, 8, 256, 3, 1, 1, True, True),       # weight-standardised conv2 (has a bias) on an 8x8 map
    (5, 512, 4, 4, 512, 3, 1, 1, True, False),      # 4x4 map: batch folded into the GEMM columns
    (1, 512, 4, 4, 128, 1, 1, 0, False, False),     # the 1x1 `fc` conv
    (2, 2048, 8, 8, 512, 1, 1, 0, False, False),    # IdtEmbed fc
    (2, 40, 17, 23, 9, 3, 2, 1, True, True),        # ragged everything: odd sizes, Cout < one tile, K not a multiple of 32
    (1, 16, 5, 5, 70, 5, 1, 2, False, True),
])
def test_conv2d_generic_matches_torch(case):
    N, Cin, H, W, Cout, k, stride, pad, affine, bias = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) if bias else None
    xin, sc, sh = x, None, None
    if affine:
        sc, sh = 1 + 0.3 * torch.randn(N, Cin, generator=g), 0.3 * torch.randn(N, Cin, generator=g)
        xin = F.relu(x * sc[:, :, None, None] + sh[:, :, None, None])
    ref = F.conv2d(xin, w, b, stride=stride, padding=pad)
    got = ops.conv2d_generic(x.to(DEV), pack.pack_generic(w).to(DEV), Cout, k, k, stride, pad,
                             None if b is None else b.to(DEV), None if sc is None else sc.to(DEV).contiguous(),
                             None if sh is None else sh.to(DEV).contiguous(), relu_in=affine)
    assert got.shape == ref.shape
    assert rel_err(got, ref) <= 2e-5
    # explicit K splits (workspace + fixed-order reduction), including more splits than K chunks
    for splits in (1, 3, 64):
        again = ops.conv2d_generic(x.to(DEV), pack.pack_generic(w).to(DEV), Cout, k, k, stride, pad,
                                   None if b is None else b.to(DEV), None if sc is None else sc.to(DEV).contiguous(),
                                   None if sh is None else sh.to(DEV).contiguous(), relu_in=affine, splits=splits)
        assert rel_err(again, ref) <= 2e-5


def test_maxpool_with_folded_norm_and_relu():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 64, 64, 64, generator=g)
    sc, sh = torch.randn(3, 64, generator=g), torch.randn(3, 64, generator=g)       # negative scales included
    ref = F.max_pool2d(F.relu(x * sc[:, :, None, None] + sh[:, :, None, None]), 3, 2, 1)
    got = ops.maxpool2d(x.to(DEV), 3, 2, 1, sc.to(DEV), sh.to(DEV), relu=True)
    assert rel_err(got, ref) <= 1e-6
    ref = F.max_pool2d(x[:, :, :33, :21], 3, 2, 1)                                   # odd sizes, -inf padding
    got = ops.maxpool2d(x[:, :, :33, :21].contiguous().to(DEV), 3, 2, 1)
    assert torch.equal(got.cpu(), ref)


def test_affine_add_relu_block_tail():
    g = torch.Generator().manual_seed(2)
    a, b = torch.randn(2, 96, 9, 7, generator=g), torch.randn(2, 96, 9, 7, generator=g)
    s = [torch.randn(2, 96, generator=g) for _ in range(4)]
    e = lambda t: t[:, :, None, None]
    ref = F.relu(a * e(s[0]) + e(s[1]) + b * e(s[2]) + e(s[3]))
    got = ops.affine_add_relu(a.to(DEV), s[0].to(DEV), s[1].to(DEV), b.to(DEV), s[2].to(DEV), s[3].to(DEV))
    assert rel_err(got, ref) <= 1e-6
    ref = F.relu(a * e(s[0]) + e(s[1]) + b)
    got = ops.affine_add_relu(a.to(DEV), s[0].to(DEV), s[1].to(DEV), b.to(DEV))
    assert rel_err(got, ref) <= 1e-6


def test_grid_sample2d_matches_torch_cpu():
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 37, 53, generator=g)
    grid = torch.rand(2, 20, 31, 2, generator=g) * 2.8 - 1.4                        # out-of-range corners -> zeros
    ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    got = ops.grid_sample2d(img.to(DEV), grid=grid.to(DEV))
    assert rel_err(got, ref) <= 2e-6
    # affine form of ExpressionEmbed: identity_grid.bmm(A^T) (expression_embedder.py:221-231)
    A = torch.tensor([[[0.6, 0.1, 0.05], [-0.1, 0.55, -0.02]], [[0.4, -0.2, 0.3], [0.15, 0.7, 0.1]]])
    n = 64
    lin = torch.linspace(-1, 1, n)
    v, u = torch.meshgrid(lin, lin, indexing="ij")
    ident = torch.stack([u, v, torch.ones_like(u)], dim=2).view(1, -1, 3)
    warp = ident.repeat_interleave(2, dim=0).bmm(A.transpose(1, 2)).view(2, n, n, 2)
    ref = F.grid_sample(img, warp, align_corners=False)
    got, gout = ops.grid_sample2d(img.to(DEV), theta=A.to(DEV), size=n, want_grid=True)
    assert (gout.cpu() - warp).abs().max().item() <= 2.4e-7                          # <= 1 ulp of coordinates ~1
    assert rel_err(got, ref) <= 2e-5                                                 # 1-ulp coordinates x image gradient


def test_mat4_inverse():
    theta = O.get_transform_matrix(1 + 0.1 * torch.randn(6, 3), 0.5 * torch.randn(6, 3), 0.1 * torch.randn(6, 3))
    got = ops.mat4_inverse(theta.to(DEV).contiguous()).cpu()
    assert (got - theta.inverse()).abs().max().item() <= 1e-6
    assert (got @ theta - torch.eye(4)).abs().max().item() <= 1e-6


# ---- whole embedders -------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emb(golden_dir):
    blob = torch.load(os.path.join(golden_dir, "embedders.pt"), weights_only=False)
    cfg, seeds = blob["cfg"], blob["seeds"]
    sds = dict(idt=E.random_state_dict(E.idt_schema(cfg), seeds["idt"]),
               expression=E.random_state_dict(E.expression_schema(cfg), seeds["expression"]),
               head_pose=E.random_state_dict(E.head_pose_schema(), seeds["head_pose"]))
    crops = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(seeds["inputs"]))
    return blob, sds, crops


def test_idt_embed_matches_reference_golden_and_oracle(emb):
    blob, sds, crops = emb
    net = E.IdtEmbed(sds["idt"], blob["cfg"], torch.device(DEV))
    got = net(crops[:1].to(DEV))
    assert got.shape == (1, 512, 4, 4)
    assert rel_err(got, blob["idt_embed"]) <= 2e-4
    with torch.no_grad():
        ref = O.idt_embed(sds["idt"], "idt_embedder_nw", crops[:1])
    assert rel_err(got, ref) <= 2e-4


def test_head_pose_regressor_matches_reference_golden(emb):
    blob, sds, crops = emb
    net = E.HeadPoseRegressor(sds["head_pose"], torch.device(DEV))
    theta, scale, rotation, translation = net.forward(crops.to(DEV), True)
    for name, got in (("theta", theta), ("scale", scale), ("rotation", rotation), ("translation", translation)):
        assert rel_err(got, blob["head_pose"][name]) <= 2e-4, name
    assert torch.equal(net(crops.to(DEV)), theta)
    # a 128x128 input skips the resize branch (head_pose_regressor.py:23-24)
    small = F.interpolate(crops, size=(128, 128), mode="bilinear")
    with torch.no_grad():
        ref = O.head_pose(sds["head_pose"], small)
    assert rel_err(net(small.to(DEV)), ref["theta"]) <= 2e-4


def test_expression_embed_matches_reference_golden(emb):
    blob, sds, crops = emb
    net = E.ExpressionEmbed(sds["expression"], blob["cfg"], torch.device(DEV))
    pose, aligned, warp = net.forward(crops.to(DEV), blob["theta"].to(DEV), want_aligned=True)
    assert pose.shape == (2, 128) and aligned.shape == (2, 3, 128, 128)
    assert (warp.cpu()[:, ::8, ::8] - blob["align_warp_sub"]).abs().max().item() <= 1e-6
    assert (aligned.cpu()[:, :, ::8, ::8] - blob["img_align_sub"]).abs().max().item() <= 2e-4   # random-noise image: |grad| ~ 1/px
    assert rel_err(pose, blob["pose_embed"]) <= 2e-4
    # the wrapper's chain: theta from the pose net feeds the alignment (notebooks/infer.py:562,596-601)
    hp = E.HeadPoseRegressor(sds["head_pose"], torch.device(DEV))
    chain = net(crops.to(DEV), hp(crops.to(DEV)))
    assert rel_err(chain, blob["pose_embed_chain"]) <= 1e-3


def test_embedders_reject_mismatched_checkpoints(emb):
    blob, sds, _ = emb
    bad = dict(sds["expression"])
    bad.pop("expression_embedder_nw.net_face.net.layer3.1.conv2.bias")
    with pytest.raises(KeyError):
        E.ExpressionEmbed(bad, blob["cfg"], torch.device(DEV))
    with pytest.raises(KeyError):
        E.IdtEmbed(sds["idt"], E.embedder_config(overrides=dict(idt_backbone="resnet18")), torch.device(DEV))
