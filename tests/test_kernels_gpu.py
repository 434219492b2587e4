"""GPU parity tests of the conv / GroupNorm / resampling / small kernels against torch CPU fp32 (same op, same
inputs).  Tolerances are stated relative to max|reference| of each tensor: the kernels accumulate in fp32 (exact-fp32
MFMA = fmaf chain), in a different summation order than the CPU library, so agreement is to fp32 rounding:
  conv:  2e-5 * max|ref|   (K up to 4608 products per output)
  norm:  1e-5 * max|ref|   (statistics reduced in fp64 on the GPU)
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

from emoportraits_amd import ops, pack  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, ref):
    return (got.cpu().double() - ref.double()).abs().max().item() / (ref.double().abs().max().item() + 1e-30)


def run_conv(N, Cin, Cout, dims, k, cfg, affine=False, relu_in=False, ups=False, res=False, res_ups=False,
             bias=True, act="none", seed=0, ksplit=None, inplace=False, precision="f32"):
    g = torch.Generator().manual_seed(seed)
    three_d = len(dims) == 3
    x = torch.randn(N, Cin, *dims, generator=g)
    kd = k if three_d else 1
    wshape = (Cout, Cin, k, k, k) if three_d else (Cout, Cin, k, k)
    w = torch.randn(*wshape, generator=g) / math.sqrt(Cin * k * k * kd)
    b = torch.randn(Cout, generator=g) if bias else None
    scale = shift = None
    xin = x
    if affine:
        scale = torch.rand(N, Cin, generator=g) + 0.5
        shift = torch.randn(N, Cin, generator=g) * 0.3
        bshape = (N, Cin) + (1,) * len(dims)
        xin = x * scale.view(bshape) + shift.view(bshape)
    if relu_in:
        xin = F.relu(xin)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = (F.conv3d if three_d else F.conv2d)(xin, w, b, padding=k // 2)
    r = None
    if res:
        rshape = list(ref.shape)
        if res_ups:
            rshape[-1] //= 2
            rshape[-2] //= 2
        r = torch.randn(*rshape, generator=g)
        ref = ref + (F.interpolate(r, scale_factor=2, mode="nearest") if res_ups else r)
    if act == "tanh":
        ref = torch.tanh(ref)
    elif act == "sigmoid":
        ref = torch.sigmoid(ref)
    elif act == "relu":
        ref = F.relu(ref)
    layer = pack.PackedConv("test", w, b, DEV, cfg=cfg, precision=precision)
    rd = None if r is None else r.to(DEV)
    got = ops.conv_igemm(x.to(DEV), layer, None if scale is None else scale.to(DEV),
                         None if shift is None else shift.to(DEV), relu_in=relu_in, ups=ups,
                         res=rd, res_ups=res_ups, act=act, ksplit=ksplit, out=rd if inplace else None)
    return rel_err(got, ref), got, ref


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("hw", [16, 32, 64, 128, 256])
def test_conv2d_3x3_all_tile_shapes(cfg, hw):
    e, got, ref = run_conv(2, 8, 40, (hw, hw), 3, cfg, seed=hw + cfg)
    assert got.shape == ref.shape
    assert e < 2e-5, e


@pytest.mark.parametrize("hw", [32, 64, 128, 256])
@pytest.mark.parametrize("mode", ["plain", "ups", "full"])
def test_conv2d_3x3_config_D_64x256_tiles(hw, mode):
    """block config 3: 64 output channels x 256 positions (2x128 / 4x64 / 8x32 position tiles)"""
    if mode == "plain":
        e, got, ref = run_conv(2, 12, 72, (hw, hw), 3, 3, seed=hw)
    elif mode == "ups":
        e, got, ref = run_conv(2, 12, 72, (hw // 2, hw // 2), 3, 3, ups=True, affine=True, relu_in=True, seed=hw + 1)
    else:
        e, got, ref = run_conv(1, 24, 130, (hw, hw), 3, 3, affine=True, relu_in=True, res=True, act="tanh", seed=hw + 2)
    assert got.shape == ref.shape and e < 2e-5, e


@pytest.mark.parametrize("hw", [64, 128, 256])
@pytest.mark.parametrize("mode", ["plain", "ups", "full"])
def test_conv2d_3x3_config_E_64x512_tiles(hw, mode):
    """block config 4: 64 output channels x 512 positions (4x128 / 8x64 position tiles, 8 accumulator tiles per wave)"""
    if mode == "plain":
        e, got, ref = run_conv(2, 12, 72, (hw, hw), 3, 4, seed=hw)
    elif mode == "ups":
        e, got, ref = run_conv(2, 12, 72, (hw // 2, hw // 2), 3, 4, ups=True, affine=True, relu_in=True, seed=hw + 1)
    else:
        e, got, ref = run_conv(1, 24, 130, (hw, hw), 3, 4, affine=True, relu_in=True, res=True, act="tanh", seed=hw + 2)
    assert got.shape == ref.shape and e < 2e-5, e


def test_conv_config_D_tile_statistics_and_unsupported_shapes():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 64, 64, generator=g)
    w = torch.randn(64, 16, 3, 3, generator=g) / 12
    layer = pack.PackedConv("d", w, None, DEV, cfg=3)
    out, st = ops.conv_igemm(x.to(DEV), layer, want_stats=True, ksplit=1)
    assert st.cnt == 256 and st.stats.shape == (2, 64 * 64 // 256, 64, 2)
    s1, h1 = ops.groupnorm_affine(out, stats=st)
    s0, h0 = ops.groupnorm_affine(out)
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item() and (h1 - h0).abs().max().item() <= 2e-6
    ref = F.conv2d(x, w, padding=1)
    assert rel_err(out, ref) < 2e-5
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):       # 16-wide outputs have no 64 x 256 tile
        ops.conv_igemm(torch.randn(1, 16, 16, 16, device=DEV), pack.PackedConv("d3", torch.randn(64, 16, 3, 3), None, DEV, cfg=3))


def test_conv_fp16_config_G_tile_statistics():
    """128 x 256 tile of the fp16-operand kernel (config 6): GroupNorm tile statistics of its epilogue (2 x 2 waves) describe
    the tensor it wrote"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 64, 64, generator=g)
    w = torch.randn(160, 32, 3, 3, generator=g) / 17
    layer = pack.PackedConv("g", w, None, DEV, cfg=6, precision="f16")
    assert layer.plan_for(64, 64, 64)[0] == 6
    out, st = ops.conv_igemm(x.to(DEV), layer, want_stats=True, ksplit=1)
    assert st.cnt == 256 and st.stats.shape == (2, 64 * 64 // 256, 160, 2)
    s1, h1 = ops.groupnorm_affine(out, stats=st)
    s0, h0 = ops.groupnorm_affine(out)
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item() and (h1 - h0).abs().max().item() <= 2e-6
    assert 1e-5 < rel_err(out, F.conv2d(x, w, padding=1)) < 3e-3


@pytest.mark.parametrize("cfg", [3, 5])
@pytest.mark.parametrize("dims", [(8, 32, 32), (4, 64, 64), (2, 128, 128), (64, 64), (128, 128)])
def test_conv_256_position_tiles_2d_and_3d(cfg, dims):
    """block configs 3 (64 x 256) and 5 (32 x 256) on 2-D and 3-D layers (depth taps as K stages, one depth slice per tile)"""
    e, got, ref = run_conv(1, 12, 40 if cfg == 3 else 24, dims, 3, cfg, affine=True, relu_in=True, res=True, seed=sum(dims) + cfg)
    assert got.shape == ref.shape and e < 2e-5, e


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("hw", [16, 64, 128])
def test_conv2d_1x1(cfg, hw):
    e, _, _ = run_conv(2, 40, 70, (hw, hw), 1, cfg, seed=3 * hw + cfg)
    assert e < 2e-5, e


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("dims", [(8, 8, 8), (16, 16, 16), (8, 32, 32), (4, 64, 64), (2, 128, 128)])
def test_conv3d_3x3x3(cfg, dims):
    e, _, _ = run_conv(1, 6, 33, dims, 3, cfg, seed=sum(dims) + cfg)
    assert e < 2e-5, e


@pytest.mark.parametrize("ksplit", [2, 5, 64])
@pytest.mark.parametrize("case", [
    dict(N=1, Cin=96, Cout=72, dims=(64, 64), k=3, cfg=0, affine=True, relu_in=True, res=True, inplace=True),
    dict(N=2, Cin=64, Cout=40, dims=(16, 16), k=3, cfg=1, ups=True, res=True, res_ups=True, act="tanh"),
    dict(N=1, Cin=70, Cout=33, dims=(8, 8, 8), k=3, cfg=1, affine=True, relu_in=True, bias=False),
    dict(N=1, Cin=200, Cout=48, dims=(32, 32), k=1, cfg=2, act="sigmoid"),
])
def test_conv_split_k_equals_single_pass(case, ksplit):
    """K split over gridDim (workspace + fixed-order epilogue): same function as the single pass, incl. the fused
    bias / residual (plain, nearest-x2, in place) / activation, ragged channel chunks and more splits than stages"""
    e, got, ref = run_conv(seed=7, ksplit=ksplit, **case)
    assert got.shape == ref.shape
    assert e < 2e-5, e
    e1, got1, _ = run_conv(seed=7, ksplit=1, **case)
    assert (got - got1).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("case", [
    dict(N=2, Cin=40, Cout=72, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True, res=True),
    dict(N=2, Cin=64, Cout=40, dims=(16, 16), k=3, cfg=3, ups=True, res=True, res_ups=True, act="tanh"),
    dict(N=1, Cin=128, Cout=128, dims=(128, 128), k=3, cfg=3, affine=True, relu_in=True),
    dict(N=1, Cin=72, Cout=33, dims=(8, 32, 32), k=3, cfg=3, affine=True, relu_in=True, bias=False),
    dict(N=1, Cin=200, Cout=48, dims=(32, 32), k=1, cfg=3, act="sigmoid"),
    dict(N=2, Cin=96, Cout=130, dims=(64, 64), k=1, cfg=3, ups=True),
    dict(N=1, Cin=96, Cout=72, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True, ksplit=3),
    dict(N=1, Cin=24, Cout=64, dims=(256, 256), k=3, cfg=3, affine=True, relu_in=True, res=True),
    dict(N=1, Cin=32, Cout=64, dims=(64, 64), k=3, cfg=3, ups=True, affine=True, relu_in=True),    # 2 x 128 tile on source pixels
    dict(N=2, Cin=16, Cout=96, dims=(32, 32), k=3, cfg=3, ups=True, affine=True),                  # 4 x 64 tile, no ReLU
    # the 128 x 256 tile (config 6, one block per CU): channel tail, upsample + residual, 3-D taps, statistics-free K split
    dict(N=2, Cin=40, Cout=200, dims=(64, 64), k=3, cfg=6, affine=True, relu_in=True, res=True),
    dict(N=2, Cin=64, Cout=128, dims=(16, 16), k=3, cfg=6, ups=True, res=True, res_ups=True, act="tanh"),
    dict(N=1, Cin=128, Cout=128, dims=(128, 128), k=3, cfg=6, affine=True, relu_in=True),
    dict(N=1, Cin=72, Cout=130, dims=(8, 32, 32), k=3, cfg=6, affine=True, relu_in=True, bias=False),
    dict(N=1, Cin=96, Cout=72, dims=(64, 64), k=3, cfg=6, affine=True, relu_in=True, ksplit=3),
    dict(N=1, Cin=32, Cout=256, dims=(64, 64), k=3, cfg=6, ups=True, affine=True, relu_in=True),
])
def test_conv_fp16_operands(case):
    """opt-in reduced-precision mode (BASELINE configs[4]): fp16 MFMA operands (32x32x16), fp32 accumulation, 64 x 256 tile.
    Operand rounding is 2^-11 relative, so the output agrees with the fp32 reference to ~1e-3 of max|out| (bound 3e-3); same
    fused prologue / epilogue, ragged channel chunks (Cin a multiple of 8 but not of 16 / 32), up-sampling gather, 3-D taps
    and K split."""
    e, got, ref = run_conv(seed=11, precision="f16", **case)
    assert got.shape == ref.shape
    print("PARITY conv fp16 operands:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
    assert e < 3e-3, e
    with pytest.raises(ValueError):
        pack.PackedConv("bad", torch.zeros(8, 8, 7, 7), None, DEV, precision="f16")
    with pytest.raises(ValueError):
        pack.PackedConv("bad", torch.zeros(64, 12, 3, 3), None, DEV, precision="f16")     # Cin not a multiple of 8


F16W8_CASES = [
    dict(N=2, Cin=128, Cout=128, dims=(128, 128), k=3, cfg=None, affine=True, relu_in=True, res=True),
    dict(N=2, Cin=48, Cout=192, dims=(64, 64), k=3, cfg=None, affine=True, relu_in=True),                   # odd tile count
    dict(N=3, Cin=64, Cout=320, dims=(32, 64), k=3, cfg=None, affine=True, relu_in=True, res=True),         # five tiles, chains
    dict(N=2, Cin=40, Cout=64, dims=(64, 128), k=3, cfg=None, affine=True, relu_in=True, res=True),         # ONE tile: half-empty pairs
    dict(N=2, Cin=64, Cout=192, dims=(32, 32), k=3, cfg=None, ups=True, affine=True, relu_in=True, res=True, res_ups=True),
    dict(N=1, Cin=24, Cout=128, dims=(6, 64, 64), k=3, cfg=None, affine=True, relu_in=True, res=True),      # depth taps
    dict(N=4, Cin=512, Cout=512, dims=(64, 64), k=3, cfg=None, affine=True, relu_in=True, res=True),        # the decoder trunk's layer
]


@pytest.mark.parametrize("rest", [True, False])
@pytest.mark.parametrize("case", F16W8_CASES)
def test_conv_fp16_operands_on_the_eight_wave_two_tile_kernel(case, rest, monkeypatch):
    """precision='f16' in the decoders' launch form runs emo_conv_igemm_f16w8 (round 6: conv_igemm_f16x2_w8.h with the leading
    product alone -- plain fp16 operands, two waves per SIMD): the fp16-operand bound against torch CPU fp32, deterministic, and
    within accumulation-order noise of the older fp16-operand kernel (EMO_F16_W8=0), which rounds the same operands.  An odd
    channel-tile count runs its last tile on that older kernel (rest: emo_conv_igemm_f16w8_rest, ABI 10 -- the planner's choice)
    or in a half-empty pair of the eight-wave kernel (EMO_F16_W8_REST=0)"""
    if not rest and (case["Cout"] // 64) % 2 == 0:
        pytest.skip("even tile count: one launch form")
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    monkeypatch.setattr(pack, "F16_W8_ODD", 1)           # (every odd tile count: the planner takes them from five tiles on only)
    monkeypatch.setattr(pack, "F16_W8_REST", rest)
    e, got, ref = run_conv(seed=12, precision="f16", **case)
    print("PARITY conv fp16 operands, eight-wave kernel:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
    assert e < 3e-3, e
    layer = pack.PackedConv("t", torch.randn(case["Cout"], case["Cin"], *([3] * len(case["dims"]))), None, DEV, precision="f16")
    Hl, Wl = [d * (2 if case.get("ups") else 1) for d in case["dims"][-2:]]
    assert layer.plan_for(1 << 12, Hl, Wl, case.get("ups", False), in_elems_per_sample=case["Cin"] * 4096)[2] == "f16w8"
    e2, got2, _ = run_conv(seed=12, precision="f16", **case)
    assert torch.equal(got, got2), "two launches on the same input differ: a race in the pipeline"
    monkeypatch.setattr(pack, "F16_W8", False)
    e3, old, _ = run_conv(seed=12, precision="f16", **case)
    assert (got - old).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_conv_fp16_odd_tile_count_tile_statistics_come_from_both_kernels(monkeypatch):
    """a 192-channel layer in the plain-fp16 mode: two channel tiles from the eight-wave kernel, the third from the older kernel
    (emo_conv_igemm_f16w8_rest) -- the GroupNorm scale / shift from the fused tile statistics equal the ones from a pass over
    the output"""
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    g = torch.Generator().manual_seed(9)
    w = torch.randn(192, 64, 3, 3, generator=g) / 24
    layer = pack.PackedConv("l3", w, None, DEV, precision="f16")
    x = torch.randn(2, 64, 64, 128, generator=g)
    assert pack.f16w8_rest_fits(192, 64, 128)
    out, st = ops.conv_igemm(x.to(DEV), layer, want_stats=True, ksplit=1)
    assert layer.last_plan[2] == "f16w8" and st is not None and st.cnt == 256
    assert 1e-5 < rel_err(out, F.conv2d(x, w, padding=1)) < 3e-3
    s1, h1 = ops.groupnorm_affine(out, stats=st)
    s0, h0 = ops.groupnorm_affine(out)
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item()
    assert (h1 - h0).abs().max().item() <= 2e-6 * max(1.0, h0.abs().max().item())


def test_conv_fp16_layer_falls_back_to_fp32_on_narrow_maps_and_writes_tile_statistics():
    """a layer built with precision='f16' runs the fp16-operand kernel where its 64 x 256 tile fits and the exact-fp32 kernel on
    the 16- / 8-wide maps (WarpGenerator); both produce the GroupNorm tile statistics"""
    g = torch.Generator().manual_seed(4)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    layer = pack.PackedConv("l", w, None, DEV, precision="f16")
    assert layer.plan_for(64, 64, 64)[2] == "f16" and layer.plan_for(2, 16, 16)[2] == "f32"
    x16 = torch.randn(1, 32, 16, 16, generator=g)
    assert rel_err(ops.conv_igemm(x16.to(DEV), layer), F.conv2d(x16, w, padding=1)) < 2e-5          # exact fp32 path
    x64 = torch.randn(2, 32, 64, 64, generator=g)
    out, st = ops.conv_igemm(x64.to(DEV), layer, want_stats=True, ksplit=1)
    assert st is not None and st.cnt == 256
    assert 1e-5 < rel_err(out, F.conv2d(x64, w, padding=1)) < 3e-3                                   # fp16 operands
    s1, h1 = ops.groupnorm_affine(out, stats=st)
    s0, h0 = ops.groupnorm_affine(out)
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item()


def test_conv_on_tensors_that_are_not_16_byte_aligned():
    """the fp16-operand kernel loads 16-byte quads and the epilogue stores 16 bytes per lane: a 4-byte-aligned input falls back
    to the exact-fp32 kernel, a 4-byte-aligned output (or residual) to scalar stores -- same results"""
    g = torch.Generator().manual_seed(23)
    N, C, S = 1, 32, 64
    w = torch.randn(64, C, 3, 3, generator=g) / math.sqrt(C * 9)
    xa = torch.randn(N, C, S, S, generator=g).to(DEV)
    xm = torch.empty(xa.numel() + 1, device=DEV)[1:].view_as(xa)
    xm.copy_(xa)
    assert xm.data_ptr() % 16 == 4
    ref = F.conv2d(xa.cpu(), w, padding=1)
    lh = pack.PackedConv("h", w, None, DEV, precision="f16")
    got_h = ops.conv_igemm(xm, lh)
    assert rel_err(got_h, ref) < 2e-5                       # fp32 accuracy: the fp32 kernel ran
    assert rel_err(ops.conv_igemm(xa, lh), ref) > 1e-5      # (the aligned tensor does take the fp16-operand kernel)
    lf = pack.PackedConv("f", w, None, DEV)
    om = torch.empty(ref.numel() + 1, device=DEV)[1:].view_as(ref)
    rm = torch.empty(ref.numel() + 1, device=DEV)[1:].view_as(ref)
    r = torch.randn(ref.shape, generator=g)
    rm.copy_(r.to(DEV))
    got = ops.conv_igemm(xa, lf, res=rm, out=om)
    assert got.data_ptr() == om.data_ptr() and om.data_ptr() % 16 == 4
    assert rel_err(got, ref + r) < 2e-5


def test_conv_fp16_operands_saturate():
    """activations beyond the fp16 range are clamped to +-65504 on the way into LDS, not turned into inf / NaN"""
    x = torch.full((1, 8, 32, 32), 1.0e6)
    x[0, :, 16:] = -3.0e5
    w = torch.zeros(32, 8, 3, 3)
    w[:, 0, 1, 1] = 1.0
    layer = pack.PackedConv("sat", w, None, DEV, cfg=3, precision="f16")
    out = ops.conv_igemm(x.to(DEV), layer).cpu()
    assert torch.isfinite(out).all()
    assert out[0, 0, 2, 2].item() == 65504.0 and out[0, 0, 24, 2].item() == -65504.0


def test_conv3d_1x1x1():
    e, _, _ = run_conv(2, 20, 12, (8, 16, 16), 1, 2, seed=5)
    assert e < 2e-5, e


@pytest.mark.parametrize("cin,cout", [(3, 3), (5, 1), (17, 129), (4, 320), (96, 96)])
def test_conv_ragged_channel_counts(cin, cout):
    for cfg in (0, 1, 2):
        e, _, _ = run_conv(1, cin, cout, (64, 64), 3, cfg, seed=cin * 7 + cout)
        assert e < 2e-5, (cfg, e)


def test_conv_fused_groupnorm_affine_relu_and_padding_semantics():
    # padding must be zero AFTER the affine+relu: with shift > 0 a wrong order shows up at the border
    for k in (1, 3):
        e, _, _ = run_conv(2, 12, 24, (64, 64), k, 1, affine=True, relu_in=True, seed=11 + k)
        assert e < 2e-5, e
    e, _, _ = run_conv(2, 12, 24, (8, 16, 16), 3, 2, affine=True, relu_in=True, seed=13)
    assert e < 2e-5, e
    e, _, _ = run_conv(1, 12, 24, (64, 64), 3, 0, affine=True, relu_in=False, seed=14)
    assert e < 2e-5, e
    e, _, _ = run_conv(1, 12, 24, (64, 64), 3, 0, affine=False, relu_in=True, seed=15)
    assert e < 2e-5, e


@pytest.mark.parametrize("k", [1, 3])
def test_conv_fused_nearest_upsample(k):
    for cfg in (0, 1, 2):
        e, got, ref = run_conv(2, 10, 36, (64, 64), k, cfg, affine=True, relu_in=True, ups=True, seed=21 + k + cfg)
        assert got.shape == ref.shape == (2, 36, 128, 128)
        assert e < 2e-5, (cfg, e)
    e, _, _ = run_conv(1, 6, 8, (128, 128), 3, 2, ups=True, seed=29)
    assert e < 2e-5, e


def test_conv_epilogue_residual_bias_activations():
    e, _, _ = run_conv(2, 8, 20, (64, 64), 3, 1, res=True, seed=31)
    assert e < 2e-5, e
    e, _, _ = run_conv(2, 8, 20, (64, 64), 3, 1, ups=True, res=True, res_ups=True, seed=32)
    assert e < 2e-5, e
    e, _, _ = run_conv(1, 8, 20, (4, 32, 32), 3, 2, res=True, bias=False, seed=33)
    assert e < 2e-5, e
    for act in ("tanh", "sigmoid", "relu"):
        e, _, _ = run_conv(1, 16, 3, (4, 64, 64), 3, 2, act=act, seed=34)
        assert e < 2e-5, (act, e)


def test_conv_residual_may_alias_output():
    g = torch.Generator().manual_seed(41)
    x = torch.randn(1, 8, 64, 64, generator=g)
    w = torch.randn(16, 8, 3, 3, generator=g) * 0.1
    r = torch.randn(1, 16, 64, 64, generator=g)
    ref = F.conv2d(x, w, padding=1) + r
    layer = pack.PackedConv("alias", w, None, DEV)
    buf = r.to(DEV).clone()
    ops.conv_igemm(x.to(DEV), layer, res=buf, out=buf)
    assert rel_err(buf, ref) < 2e-5


def test_conv_released_decoder_layer_shapes():
    """the heaviest released shapes, batch 1: K=4608 trunk conv and the 512^2 conv"""
    e, _, _ = run_conv(1, 512, 512, (64, 64), 3, 0, affine=True, relu_in=True, res=True, seed=51)
    assert e < 2e-5, e
    e, _, _ = run_conv(1, 192, 128, (256, 256), 3, 0, affine=True, relu_in=True, ups=True, seed=52)
    assert e < 2e-5, e
    e, _, _ = run_conv(1, 1536, 512, (64, 64), 1, 0, seed=53)
    assert e < 2e-5, e


def test_conv_unsupported_width_is_reported():
    layer = pack.PackedConv("bad", torch.randn(4, 4, 3, 3), None, DEV)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.conv_igemm(torch.randn(1, 4, 12, 12, device=DEV), layer)


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 64, 8, 8), (3, 96, 16, 64, 64), (1, 128, 512, 512), (2, 32, 5, 7, 3)])
def test_groupnorm_affine(shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g) * 3 + 1.5
    C = shape[1]
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, 1e-5)
    scale, shift, mean, rstd = ops.groupnorm_affine(x.to(DEV), gamma.to(DEV), beta.to(DEV), want_stats=True)
    bshape = (shape[0], C) + (1,) * (len(shape) - 2)
    got = x * scale.cpu().view(bshape) + shift.cpu().view(bshape)
    assert rel_err(got, ref) < 1e-5
    xg = x.view(shape[0], 32, -1).double()
    assert torch.allclose(mean.cpu().double(), xg.mean(-1), atol=1e-5)
    assert torch.allclose(rstd.cpu().double(), 1 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5), rtol=1e-5)
    # no affine
    s2, h2 = ops.groupnorm_affine(x.to(DEV))
    got2 = x * s2.cpu().view(bshape) + h2.cpu().view(bshape)
    assert rel_err(got2, F.group_norm(x, 32, None, None, 1e-5)) < 1e-5


def test_groupnorm_large_mean_is_stable():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 64, 64, generator=g) * 0.01 + 1000.0
    s, h = ops.groupnorm_affine(x.to(DEV))
    got = x * s.cpu().view(1, 32, 1, 1) + h.cpu().view(1, 32, 1, 1)
    ref = F.group_norm(x.double(), 32).float()
    assert (got - ref).abs().max().item() < 0.05   # fp32 x*scale+shift cancellation bound at |x|=1e3, rstd=1e2


# ---- GroupNorm statistics reduced in the conv epilogue (conv_igemm.h gn_stats -> emo_groupnorm_affine_from_tiles_f32) ----
@pytest.mark.parametrize("case", [
    # (N, Cin, Cout, dims, k, cfg, res, bias)
    (2, 64, 128, (64, 64), 3, 0, True, True),       # 128-row config, waves 2x2
    (2, 64, 320, (32, 32), 3, 1, True, False),      # 64-row config, waves 1x4, 10 channels per group
    (1, 32, 96, (8, 32, 32), 3, 2, False, True),    # 32-row config, 3-D, 3 channels per group
    (3, 48, 192, (128, 128), 1, 1, False, False),   # 1x1, 6 channels per group
    (2, 16, 32, (16, 16), 3, 2, True, True),        # one channel per group
])
def test_conv_epilogue_groupnorm_statistics(case):
    """scale / shift from the tile statistics the conv epilogue writes == GroupNorm of the conv output (torch CPU, fp64
    statistics), and identical (to fp32 rounding of the final affine) to the separate-pass kernel on the same tensor"""
    N, Cin, Cout, dims, k, cfg, res, bias = case
    g = torch.Generator().manual_seed(11)
    three_d = len(dims) == 3
    x = torch.randn(N, Cin, *dims, generator=g)
    wshape = (Cout, Cin) + (k,) * len(dims)
    w = torch.randn(*wshape, generator=g) / math.sqrt(Cin * k ** len(dims))
    b = torch.randn(Cout, generator=g) if bias else None
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    layer = pack.PackedConv("test", w, b, DEV, cfg=cfg)
    r = torch.randn(N, Cout, *dims, generator=g).to(DEV) if res else None
    out, st = ops.conv_igemm(x.to(DEV), layer, res=r, want_stats=True, ksplit=1)   # (small test launches would be K-split)
    assert st is not None and st.stats.shape == (N, out[0, 0].numel() // 128, Cout, 2)
    s1, h1 = ops.groupnorm_affine(out, gamma.to(DEV), beta.to(DEV), stats=st)
    s0, h0 = ops.groupnorm_affine(out, gamma.to(DEV), beta.to(DEV))
    o = out.cpu()
    ref = F.group_norm(o.double(), 32, gamma.double(), beta.double()).float()
    bshape = (N, Cout) + (1,) * len(dims)
    got = o * s1.cpu().view(bshape) + h1.cpu().view(bshape)
    assert rel_err(got, ref) < 1e-5
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item()
    assert (h1 - h0).abs().max().item() <= 2e-6 * max(h0.abs().max().item(), 1.0)
    # the conv output itself is unchanged by asking for statistics
    assert torch.equal(out, ops.conv_igemm(x.to(DEV), layer, res=r, ksplit=1))


def test_conv_epilogue_groupnorm_statistics_large_mean_is_stable():
    """same stress as test_groupnorm_large_mean_is_stable, through the conv epilogue: outputs 1000 +- 0.01 (bias 1000).
    The tile statistics are (mean, centred sum of squares), so nothing of the form E[x^2] - mean^2 is ever evaluated in fp32."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 64, 64, generator=g)
    w = torch.randn(32, 16, 3, 3, generator=g) * (0.01 / 12.0)
    b = torch.full((32,), 1000.0)
    layer = pack.PackedConv("test", w, b, DEV, cfg=2)
    out, st = ops.conv_igemm(x.to(DEV), layer, want_stats=True, ksplit=1)
    s, h = ops.groupnorm_affine(out, stats=st)
    o = out.cpu()
    got = o * s.cpu().view(1, 32, 1, 1) + h.cpu().view(1, 32, 1, 1)
    ref = F.group_norm(o.double(), 32).float()
    assert o.std().item() < 0.05 and abs(o.mean().item() - 1000.0) < 0.1
    assert (got - ref).abs().max().item() < 0.05   # fp32 x*scale+shift cancellation bound at |x|=1e3, rstd~1e2


def test_conv_statistics_are_refused_where_they_cannot_be_produced():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 16, 16, generator=g).to(DEV)
    layer = pack.PackedConv("test", torch.randn(64, 64, 3, 3, generator=g), None, DEV, cfg=0)
    out, st = ops.conv_igemm(x, layer, ksplit=4, want_stats=True)      # K-split launch: no tile statistics, caller falls back
    assert st is None
    ref = ops.conv_igemm(x, layer, ksplit=1)
    assert rel_err(out, ref.cpu()) < 1e-5


def test_upsampling_resblock_skip_commutes_with_nearest_upsampling():
    """nets.ResBlock runs the 1x1 skip of an up-block on the pre-upsample tensor (conv1x1(up(x)) == up(conv1x1(x))
    element for element) and lets conv2's epilogue read it at (y>>1, x>>1): bit-identical to the direct form"""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 32, 32, generator=g).to(DEV)
    h = torch.randn(2, 96, 64, 64, generator=g).to(DEV)
    skip = pack.PackedConv("skip", torch.randn(96, 64, 1, 1, generator=g) / 8, None, DEV)
    conv2 = pack.PackedConv("conv2", torch.randn(96, 96, 3, 3, generator=g) / 30, None, DEV)
    r_big = ops.conv_igemm(x, skip, ups=True, ksplit=1)
    direct = ops.conv_igemm(h, conv2, res=r_big)
    r_small = ops.conv_igemm(x, skip, ksplit=1)
    commuted = ops.conv_igemm(h, conv2, res=r_small, res_ups=True)
    assert torch.equal(F.interpolate(r_small, scale_factor=2, mode="nearest"), r_big)
    assert torch.equal(direct, commuted)


def test_adaptive_groupnorm_matches_reference_quirk():
    g = torch.Generator().manual_seed(6)
    N, C = 3, 64
    x = torch.randn(N, C, 4, 8, 8, generator=g)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    dg, db = torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)
    sd = {"p.weight": gamma, "p.bias": beta}
    ref = O.ada_group_norm(x, sd, "p", (dg, db))
    big_g = torch.zeros(N, 200)
    big_b = torch.zeros(N, 200)
    big_g[:, 100:100 + C] = gamma[None] + dg
    big_b[:, 100:100 + C] = beta[None] + db
    big_g, big_b = big_g.to(DEV), big_b.to(DEV)
    s, h = ops.groupnorm_affine(x.to(DEV), gamma.to(DEV), beta.to(DEV), big_g[:, 100:100 + C], big_b[:, 100:100 + C])
    got = x * s.cpu().view(N, C, 1, 1, 1) + h.cpu().view(N, C, 1, 1, 1)
    assert rel_err(got, ref) < 1e-5


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act", ["none", "tanh", "sigmoid"])
@pytest.mark.parametrize("N,cin,cout,dims,affine,relu_in", [
    (2, 128, 3, (16, 32), True, True),         # the image head's form (decoder.py:381-392) on a small map
    (3, 40, 4, (2, 6, 32), True, False),       # 3-D positions, a channel count off the unroll, no ReLU
    (1, 7, 1, (4, 32), False, True),           # fewer channels than one unrolled group, no affine
    (2, 64, 2, (8, 128), False, False),
])
def test_conv_head_stream(N, cin, cout, dims, affine, relu_in, act):
    """ops.conv_head (csrc/conv_head.hip): a 1x1 convolution with at most 4 output channels as a stream -- against torch's CPU
    convolution of the same operands (2e-5 of the output's magnitude: the bound of the implicit-GEMM kernel it replaces for the
    image head) and against that kernel on the same tensors"""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(N, cin, *dims, generator=g)
    w = torch.randn(cout, cin, *([1] * len(dims)), generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    scale = shift = None
    xin = x
    if affine:
        scale, shift = torch.rand(N, cin, generator=g) + 0.5, torch.randn(N, cin, generator=g) * 0.3
        bs = (N, cin) + (1,) * len(dims)
        xin = x * scale.view(bs) + shift.view(bs)
    if relu_in:
        xin = F.relu(xin)
    ref = (F.conv3d if len(dims) == 3 else F.conv2d)(xin, w, b)
    ref = {"none": lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](ref)
    layer = pack.PackedConv("head", w, b, DEV)
    args = (x.to(DEV), layer, None if scale is None else scale.to(DEV), None if shift is None else shift.to(DEV))
    got = ops.conv_head(*args, relu_in=relu_in, act=act)
    assert layer.last_plan == ("head", 1, "stream") and got.shape == ref.shape
    assert rel_err(got, ref) < 2e-5
    if cin % 16 == 0:                                   # (whole channel stages of the fp32 MFMA kernel it replaces)
        mfma = ops.conv_igemm(*args, relu_in=relu_in, act=act)
        assert layer.last_plan[2] == "f32" and rel_err(got, mfma.cpu()) < 2e-5


def test_conv_head_launch_forms_the_stream_does_not_take_run_the_mfma_kernel():
    g = torch.Generator().manual_seed(3)
    w, b = torch.randn(3, 16, 1, 1, generator=g) / 4, torch.randn(3, generator=g)
    layer = pack.PackedConv("head", w, b, DEV)
    x = torch.randn(1, 16, 8, 32, generator=g)
    buf = torch.empty(x.numel() + 4, device=DEV)
    xd = buf[1:1 + x.numel()].view_as(x)                                        # 4 bytes off a 16-byte boundary
    xd.copy_(x)
    got = ops.conv_head(xd, layer, act="sigmoid")
    assert layer.last_plan[2] == "f32" and rel_err(got, torch.sigmoid(F.conv2d(x, w, b))) < 2e-5
    wide = pack.PackedConv("wide", torch.randn(8, 16, 1, 1, generator=g), None, DEV)   # more than 4 output channels
    x = torch.randn(1, 16, 8, 32, generator=g)
    got = ops.conv_head(x.to(DEV), wide)
    assert wide.last_plan[2] == "f32" and rel_err(got, F.conv2d(x, wide._weight)) < 2e-5


@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2), (2, 1, 1)])
def test_upsample_trilinear(factors):
    x = torch.randn(2, 5, 4, 6, 7, generator=torch.Generator().manual_seed(1))
    ref = F.interpolate(x, scale_factor=tuple(float(f) for f in factors), mode="trilinear")
    got = ops.upsample_trilinear(x.to(DEV), factors)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-6


@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2), (2, 1, 2), (1, 1, 2)])
@pytest.mark.parametrize("shape", [(2, 5, 4, 6, 8), (1, 3, 2, 4, 2), (1, 2, 3, 5, 4), (2, 4, 4, 16, 32), (1, 2, 1, 1, 64), (1, 1, 2, 2, 6)])
def test_upsample_trilinear_even_widths_take_the_block_kernel(shape, factors):
    """width factor 2 on an even width: one thread per block of up to 4 x 2 x 2 outputs that share their inputs
    (csrc/resample.hip).  Bit for bit what the one-output-per-thread kernel computes (reached here through an output pointer
    that is not 16-byte aligned), first / last pairs and quads of every axis included; both within 1e-6 of ATen's CPU kernel"""
    from emoportraits_amd import hip
    lib = hip.load()
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5))
    ref = F.interpolate(x, scale_factor=tuple(float(f) for f in factors), mode="trilinear")
    xd = x.to(DEV)
    got = ops.upsample_trilinear(xd, factors)
    assert got.shape == ref.shape and got.data_ptr() % 16 == 0
    buf = torch.empty(ref.numel() + 4, device=DEV)
    plain = buf[1:1 + ref.numel()]                              # 4 bytes off: the generic kernel
    N, C, D, H, W = shape
    hip.check(lib.emo_upsample_trilinear_f32(hip.ptr(xd), hip.ptr(plain), N * C, D, H, W, *factors, hip.current_stream()), "upsample")
    assert torch.equal(got.flatten(), plain)
    assert rel_err(got, ref) < 1e-6


@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2)])
@pytest.mark.parametrize("shape", [(2, 64, 4, 4, 4), (3, 32, 2, 6, 8), (1, 128, 8, 8, 8), (2, 64, 8, 32, 32)])
def test_upsample_trilinear_leaves_the_groupnorm_sums_of_its_output(shape, factors):
    """ops.upsample_trilinear(..., gn_groups=32): the tensor is bit for bit the plain call's; the (scale, shift) GroupNorm makes
    of the sums the kernel left behind are those of a reduction pass over the tensor (both accumulate fp64 sums of the same
    fp32 values, in different orders: 1e-6 of the affine's magnitude), adaptive weights and group statistics included
    (WarpGenerator: F.interpolate -> ResBlock3d's first norm, warp_generator_resnet.py:163-166)"""
    g = torch.Generator().manual_seed(11)
    N, C = shape[:2]
    x = (torch.randn(*shape, generator=g) * 3.0 + 0.7).to(DEV)
    gamma, beta = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    ag, ab = torch.randn(N, C, generator=g).to(DEV), torch.randn(N, C, generator=g).to(DEV)
    plain = ops.upsample_trilinear(x, factors)
    out, sums = ops.upsample_trilinear(x, factors, gn_groups=32)
    assert sums is not None and sums.split >= 1 and torch.equal(out, plain)
    want = ops.groupnorm_affine(plain, gamma, beta, ag, ab, want_stats=True)
    got = ops.groupnorm_affine(out, gamma, beta, ag, ab, want_stats=True, stats=sums)
    for a, b in zip(got, want):
        assert rel_err(a, b.cpu()) < 1e-6
    ref = F.group_norm(plain.cpu().double(), 32, gamma.cpu().double(), beta.cpu().double(), 1e-5)
    ref = ref * ag.cpu().double().view(N, C, 1, 1, 1) + ab.cpu().double().view(N, C, 1, 1, 1)
    mine = plain.cpu().double() * got[0].cpu().double().view(N, C, 1, 1, 1) + got[1].cpu().double().view(N, C, 1, 1, 1)
    assert rel_err(mine, ref) < 1e-5
    with pytest.raises(ValueError):
        ops.groupnorm_affine(x, gamma, beta, stats=sums)          # the sums of another tensor


def test_upsample_trilinear_without_a_fused_form_returns_no_sums():
    x = torch.randn(1, 32, 2, 4, 5, generator=torch.Generator().manual_seed(12)).to(DEV)      # odd width
    out, sums = ops.upsample_trilinear(x, (1, 2, 2), gn_groups=32)
    assert sums is None and torch.equal(out, ops.upsample_trilinear(x, (1, 2, 2)))


@pytest.mark.parametrize("kernel", [(2, 1, 1), (1, 2, 2), (2, 2, 2)])
def test_avgpool3d(kernel):
    x = torch.randn(2, 3, 4, 6, 8, generator=torch.Generator().manual_seed(2))
    assert rel_err(ops.avgpool(x.to(DEV), kernel), F.avg_pool3d(x, kernel, kernel)) < 1e-6


def test_avgpool2d_and_add():
    x = torch.randn(2, 3, 6, 8, generator=torch.Generator().manual_seed(3))
    assert rel_err(ops.avgpool(x.to(DEV), (2, 2)), F.avg_pool2d(x, 2)) < 1e-6
    y = torch.randn(3, 6, 8)
    assert rel_err(ops.add(x.to(DEV), y.to(DEV), 0.5), (x + y[None]) * 0.5) < 1e-7


def test_small_gemm_and_projector():
    g = torch.Generator().manual_seed(4)
    A = torch.randn(70, 300, generator=g)
    for NN in (1, 2, 4, 16):
        B = torch.randn(3, 300, NN, generator=g)
        got = ops.small_gemm(A.to(DEV), B.to(DEV), NN)
        assert rel_err(got, torch.einsum("mk,bkn->bmn", A, B)) < 1e-5
    T = torch.randn(2, 10, 16, generator=g)
    V = torch.randn(3, 16, 2, generator=g)
    nor = torch.tensor([0, 0, 0, 1, 1, 1, 1, 2, 2, 2], dtype=torch.int32)
    gamma, beta = torch.randn(10, generator=g), torch.randn(10, generator=g)
    ag, ab = ops.projector_finalize(T.to(DEV), V.to(DEV), nor.to(DEV), gamma.to(DEV), beta.to(DEV))
    P = torch.einsum("brk,rkj->brj", T, V[nor.long()])
    assert rel_err(ag, gamma[None] + P[..., 0]) < 1e-5
    assert rel_err(ab, beta[None] + P[..., 1]) < 1e-5


def test_pose_theta_and_pack(golden_dir):
    import numpy as np
    p = dict(np.load(os.path.join(golden_dir, "pose_theta.npz")))
    t = lambda k: torch.from_numpy(p[k]).to(DEV)
    got = ops.pose_theta(t("scale"), t("rotation"), t("translation")).cpu()
    assert (got - torch.from_numpy(p["theta"])).abs().max().item() < 2e-6
    got1 = ops.pose_theta(t("scale")[:, :1].contiguous(), t("rotation"), t("translation")).cpu()
    assert (got1 - torch.from_numpy(p["theta_scalar_scale"])).abs().max().item() < 2e-6
    img = torch.rand(2, 3, 16, 24, generator=torch.Generator().manual_seed(9)) * 1.4 - 0.2
    ref = img.clamp(0, 1).mul(255).byte().permute(0, 2, 3, 1)
    assert torch.equal(ops.pack_rgb8(img.to(DEV)).cpu(), ref)


def test_resize2d_windows_is_one_launch_of_the_per_frame_crops():
    """ops.resize2d_windows (ABI 9): a batch of frames, each with its own crop window (notebooks/infer.py:301-352 crops every frame
    around its own face box), in one launch -- bit for bit what ops.resize2d gives frame by frame"""
    g = torch.Generator().manual_seed(14)
    x = torch.rand(5, 3, 96, 128, generator=g).to(DEV)
    wins = [(10, 5, 80, 80), (40, 16, 64, 64), (0, 0, 128, 96), (100, 60, 28, 36), (7, 9, 50, 40)]
    got = ops.resize2d_windows(x, (64, 64), wins, "bicubic", clamp01=True)
    for i, w in enumerate(wins):
        assert torch.equal(got[i:i + 1], ops.resize2d(x[i:i + 1], (64, 64), "bicubic", window=w, clamp01=True))
    dev_wins = torch.tensor(wins, dtype=torch.int32, device=DEV)
    assert torch.equal(ops.resize2d_windows(x, (64, 64), dev_wins, "bicubic", clamp01=True), got)
    with pytest.raises(ValueError):
        ops.resize2d_windows(x, (64, 64), [(100, 60, 40, 36)] * 5)                      # a window that leaves the frame


@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
@pytest.mark.parametrize("sizes", [((37, 53), (64, 64)), ((128, 96), (48, 80)), ((64, 64), (64, 64)), ((256, 256), (512, 512))])
def test_resize2d_matches_interpolate(mode, sizes):
    (H, W), (Ho, Wo) = sizes
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(H + Wo))
    ref = F.interpolate(x, size=(Ho, Wo), mode=mode, align_corners=False)
    got = ops.resize2d(x.to(DEV), (Ho, Wo), mode)
    assert got.shape == ref.shape
    # source coordinates reach ~128: one ulp of the coordinate (7.6e-6, FMA contraction of the CPU build) times the
    # unit gradient of a random-noise image
    assert (got.cpu() - ref).abs().max().item() <= 1e-5
