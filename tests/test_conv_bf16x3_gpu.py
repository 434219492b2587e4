"""GPU parity of emo_conv_igemm_bf16x3 (csrc/conv_igemm_bf16x3.h): the fp32 3x3 convolution computed on the bf16 matrix pipes
from exact three-way operand splits.  It is held to the SAME bound as the exact-fp32 MFMA kernel -- 2e-5 * max|ref| against
torch CPU fp32 (tests/test_kernels_gpu.py) -- and, against an fp64 reference, to the fp32 kernel's own error."""
import math

import pytest
import torch
import torch.nn.functional as F

from emoportraits_amd import ops, pack
from test_kernels_gpu import DEV, rel_err, run_conv

pytestmark = pytest.mark.gpu


CASES = [
    dict(N=2, Cin=40, Cout=120, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True, res=True),         # ragged channel group
    dict(N=2, Cin=64, Cout=64, dims=(32, 32), k=3, cfg=3, ups=True, res=True, res_ups=True, act="tanh"),
    dict(N=1, Cin=128, Cout=128, dims=(128, 128), k=3, cfg=3, affine=True, relu_in=True),
    dict(N=1, Cin=72, Cout=190, dims=(8, 64, 64), k=3, cfg=3, affine=True, relu_in=True, bias=False),   # 3-D: depth taps as stages
    dict(N=1, Cin=96, Cout=112, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True, ksplit=3),
    dict(N=1, Cin=24, Cout=64, dims=(256, 256), k=3, cfg=3, affine=True, relu_in=True, res=True),
    dict(N=1, Cin=32, Cout=64, dims=(64, 64), k=3, cfg=3, ups=True, affine=True, relu_in=True),
    dict(N=2, Cin=16, Cout=96, dims=(32, 32), k=3, cfg=3, ups=True, affine=True),                       # one stage, no ReLU
    dict(N=1, Cin=8, Cout=64, dims=(4, 64), k=3, cfg=3),                                               # half a stage, one tile
    dict(N=3, Cin=48, Cout=200, dims=(8, 128), k=3, cfg=3, act="sigmoid"),
    dict(N=1, Cin=512, Cout=64, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True),                 # 32 stages
    dict(N=2, Cin=128, Cout=64, dims=(16, 32, 32), k=3, cfg=3, affine=True, relu_in=True, res=True),   # 8 x 32 tiles, 3-D
    dict(N=1, Cin=24, Cout=120, dims=(8, 32), k=3, cfg=3, bias=False),                                 # 8 x 32, one tile per plane
    dict(N=6, Cin=64, Cout=320, dims=(64, 128), k=3, cfg=3, affine=True, relu_in=True),                # 960 tiles: persistent blocks
    dict(N=2, Cin=64, Cout=128, dims=(6, 16, 16), k=3, cfg=3, affine=True, relu_in=True, res=True),    # 16 x 16 tiles, 3-D
    dict(N=3, Cin=40, Cout=104, dims=(32, 16), k=3, cfg=3, affine=True, act="tanh"),                   # 16 x 16, two tiles per plane
]


BM32_CASES = [
    dict(N=2, Cin=64, Cout=32, dims=(8, 64, 64), k=3, cfg=3, affine=True, relu_in=True, res=True),      # the WarpGenerator's layers
    dict(N=1, Cin=32, Cout=32, dims=(128, 128), k=3, cfg=3, affine=True, relu_in=True),
    dict(N=3, Cin=16, Cout=32, dims=(16, 64), k=3, cfg=3, bias=False, res=True),                        # one stage per item
    dict(N=2, Cin=48, Cout=24, dims=(32, 64), k=3, cfg=3, affine=True, relu_in=True),                   # ragged channel tile
    dict(N=6, Cin=32, Cout=32, dims=(64, 128), k=3, cfg=3, affine=True, relu_in=True, res=True),        # 768 items: chains
    dict(N=2, Cin=32, Cout=3, dims=(4, 64, 64), k=3, cfg=3, affine=True, relu_in=True, act="tanh"),     # the warp head
]


@pytest.mark.parametrize("case", BM32_CASES)
def test_fp16_split_on_32_row_channel_tiles(case):
    """layers with at most 32 output channels run a 32-row channel tile (csrc/conv_igemm_bf16x3.h, BMT = 32; block config F as
    the tile id) instead of a half-empty 64-row one: same bound, deterministic, and the launch plan says so"""
    e, got, ref = run_conv(seed=31, precision="f16x2", **case)
    print("PARITY conv f16x2, 32-row channel tiles:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
    assert e < 2e-5, e
    e2, got2, _ = run_conv(seed=31, precision="f16x2", **case)
    assert torch.equal(got, got2), "two launches on the same input differ: a race in the pipeline"
    layer = pack.PackedConv("t", torch.randn(case["Cout"], case["Cin"], *([3] * len(case["dims"]))), None, DEV, precision="f16x2")
    assert layer.plan_for(64, case["dims"][-2], case["dims"][-1])[0] == (pack.CFG_F if pack.F16X2_BM32 else pack.CFG_D)


def test_fp16_split_32_row_tiles_statistics_and_range_check():
    """tile statistics from the 32-row tile give the GroupNorm affine of a direct reduction; an out-of-range input raises the
    overflow word and the guarded bf16x3 launch (64-row tile) rewrites output and statistics: bit-identical to a plain bf16x3 launch"""
    g = torch.Generator().manual_seed(39)
    N, Cin, Cout, H, W = 2, 64, 32, 16, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(N, Cin, H, W, generator=g)
    l2 = pack.PackedConv("bm32", w, b, DEV, precision="f16x2")
    ops.clear_overflow_flags(DEV)
    out, st = ops.conv_igemm(x.to(DEV), l2, relu_in=True, want_stats=True)
    assert ops.overflow_events(DEV) == {} and rel_err(out, F.conv2d(F.relu(x), w, b, padding=1)) < 2e-5
    s1, h1 = ops.groupnorm_affine(out, stats=st)
    s0, h0 = ops.groupnorm_affine(out)
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item() and (h1 - h0).abs().max().item() <= 2e-6
    xa = x.clone()
    xa[1, 7, 5, 33] = 5000.0
    got, sg = ops.conv_igemm(xa.to(DEV), l2, relu_in=True, want_stats=True)
    assert list(ops.overflow_events(DEV).values()) == ["bm32"]
    # the exact launch the guard issued, on its own: the bf16 split on the (half-empty) 64-row tile
    want = torch.empty_like(got)
    stw = torch.empty_like(sg.stats)
    from emoportraits_amd import hip
    lib = hip.load()
    rc = lib.emo_conv_igemm_bf16x3(hip.ptr(xa.to(DEV)), hip.ptr(l2.packed(pack.CFG_D, "bf16x3")), hip.ptr(l2.bias), None, None, None,
                                   hip.ptr(want), N, Cin, Cout, 1, H, W, 1, 3, 3, 0, 1, 0, 0, pack.CFG_D, 1, None, hip.ptr(stw),
                                   hip.current_stream(), None)
    hip.check(rc, "emo_conv_igemm_bf16x3")
    assert torch.equal(got.cpu(), want.cpu()) and torch.equal(sg.stats.cpu(), stw.cpu())
    ops.clear_overflow_flags(DEV)


def test_fp16_split_on_a_half_empty_channel_tile():
    """32 output channels (the WarpGenerator's last 3-D block: pack.supports_bf16x3(..., "f16x2")) in the general epilogue's forms
    (activation); the bf16 split declines such a layer"""
    for case in (dict(N=2, Cin=64, Cout=32, dims=(4, 64, 64), k=3, cfg=3, affine=True, relu_in=True, res=True),
                 dict(N=1, Cin=32, Cout=32, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True, act="tanh")):
        e, got, ref = run_conv(seed=23, precision="f16x2", **case)
        print("PARITY conv f16x2, 32 of 64 tile rows:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
        assert e < 2e-5, e
    with pytest.raises(ValueError):
        pack.PackedConv("t", torch.randn(32, 64, 3, 3), None, DEV, precision="bf16x3")


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("case", CASES)
def test_conv_bf16x3_meets_the_fp32_kernel_bound(case, precision):
    """(f16x2: the opt-in two-term fp16 split of the scaled operands, same kernel with SPLIT = 2, same bound)"""
    e, got, ref = run_conv(seed=21, precision=precision, **case)
    assert got.shape == ref.shape
    print(f"PARITY conv {precision}:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
    assert e < 2e-5, e
    e2, got2, _ = run_conv(seed=21, precision=precision, **case)
    assert torch.equal(got, got2), "two launches on the same input differ: a race in the pipeline"


# The shapes the "error against fp64" statement of the bench line (config.conv_arithmetic) rests on: the round-5 shape, the
# decoder's longest accumulation (512 -> 512, K = 4608: 12 of the step's 31 split launches), the fused-upsample form, the largest
# map and a 3-D layer of the WarpGenerator (depth taps as K stages).  bench.py quotes the WORST ratio of this list
# (profiles/r6_parity.txt).
FP64_SHAPES = [
    dict(tag="192->128 @64^2 (round 5's shape)", N=2, Cin=192, Cout=128, dims=(64, 64), ups=False, seed=5),
    dict(tag="512->512 @64^2, K = 4608", N=2, Cin=512, Cout=512, dims=(64, 64), ups=False, seed=6),
    dict(tag="512->320, x2 upsample fused @64^2", N=1, Cin=512, Cout=320, dims=(64, 64), ups=True, seed=7),
    dict(tag="128->128 @256^2", N=1, Cin=128, Cout=128, dims=(256, 256), ups=False, seed=8),
    dict(tag="128->64 @32^3 (3-D)", N=1, Cin=128, Cout=64, dims=(32, 32, 32), ups=False, seed=9),
]


@pytest.mark.parametrize("shape", FP64_SHAPES, ids=[s["tag"] for s in FP64_SHAPES])
def test_conv_bf16x3_is_as_close_to_fp64_as_the_fp32_kernel(shape):
    """error against an fp64 convolution: the split kernels' may not exceed the exact-fp32 MFMA kernel's (plus rounding of the
    output value itself) -- on every shape of FP64_SHAPES, not on one"""
    import os
    g = torch.Generator().manual_seed(shape["seed"])
    N, Cin, Cout, dims = shape["N"], shape["Cin"], shape["Cout"], shape["dims"]
    three_d = len(dims) == 3
    x = torch.relu(torch.randn(N, Cin, *dims, generator=g) * 3 + 0.5)
    w = torch.randn(Cout, Cin, *([3] * len(dims)), generator=g) / math.sqrt(Cin * 3 ** len(dims))
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if shape["ups"] else x
    ref = (F.conv3d if three_d else F.conv2d)(xin.double(), w.double(), padding=1)
    scale = ref.abs().mean().item()
    outs = {}
    for prec in ("f32", "bf16x3", "f16x2"):
        layer = pack.PackedConv(prec, w, None, DEV, cfg=3, precision=prec)
        y = ops.conv_igemm(x.to(DEV), layer, ups=shape["ups"])
        assert layer.last_plan[2] == prec, layer.last_plan
        err = (y.cpu().double() - ref).abs()
        outs[prec] = (err.mean().item() / scale, err.max().item() / scale)
    r16, r3 = outs["f16x2"][0] / outs["f32"][0], outs["bf16x3"][0] / outs["f32"][0]
    print("PARITY conv vs fp64 [%s] (rel mean, rel max): fp32 MFMA %.2e %.2e | bf16x3 %.2e %.2e | f16x2 %.2e %.2e | mean-error ratio "
          "to the fp32 MFMA kernel: f16x2 %.3f bf16x3 %.3f" % ((shape["tag"],) + outs["f32"] + outs["bf16x3"] + outs["f16x2"] + (r16, r3)))
    # (the default mode: the bound is what the bench line advertises plus slack for the seed, not a factor a regression could
    # hide in)
    assert outs["f16x2"][0] <= 1.25 * outs["f32"][0] + 1e-8 and outs["f16x2"][1] <= 2.0 * outs["f32"][1] + 1e-7
    assert outs["bf16x3"][0] <= 1.1 * outs["f32"][0] + 1e-8        # mean error: no worse than the fp32 MFMA kernel
    assert outs["bf16x3"][1] <= 2.0 * outs["f32"][1] + 1e-7        # worst element of 1e6 (a tail statistic: factor 2)


def test_conv_bf16x3_layer_plan_and_tile_statistics():
    """a layer built with precision='bf16x3' runs the split kernel where one of its tiles (4 x 64, 8 x 32, 16 x 16) fits and the
    exact-fp32 kernel on other maps (whose tile statistics tests/test_kernels_gpu.py covers); the split kernel writes the
    GroupNorm tile statistics of what it stored"""
    g = torch.Generator().manual_seed(4)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    layer = pack.PackedConv("l", w, None, DEV, precision="bf16x3")
    assert layer.plan_for(64, 64, 64)[2] == "bf16x3" and layer.plan_for(2, 16, 16)[2] == "bf16x3" and layer.plan_for(18, 48, 48)[2] == "f32"
    for hw in (64, 16):
        x = torch.randn(2, 32, hw, hw, generator=g).to(DEV)
        out, st = ops.conv_igemm(x, layer, want_stats=True, ksplit=1)
        assert st is not None
        s1, h1 = ops.groupnorm_affine(out, stats=st)
        s0, h0 = ops.groupnorm_affine(out)
        assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item() and (h1 - h0).abs().max().item() <= 2e-6
        assert rel_err(out, F.conv2d(x.cpu(), w, padding=1)) < 2e-5
    with pytest.raises(ValueError):
        pack.PackedConv("bad", torch.zeros(64, 12, 3, 3), None, DEV, precision="bf16x3")   # Cin not a multiple of 8
    with pytest.raises(ValueError):
        pack.PackedConv("bad", torch.zeros(64, 16, 1, 1), None, DEV, precision="bf16x3")   # 3x3 only


def test_conv_bf16x3_decoder_layer_at_bench_size_agrees_with_the_fp32_kernel():
    """128 -> 128 at 512 x 512, 4 frames (the bench's largest layer shape): split kernel against the exact-fp32 MFMA kernel"""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 128, 512, 512, generator=g).to(DEV)
    w = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    sc = (torch.rand(4, 128, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(4, 128, generator=g) * 0.3).to(DEV)
    a = ops.conv_igemm(x, pack.PackedConv("a", w, None, DEV, precision="f32"), sc, sh, relu_in=True)
    b = ops.conv_igemm(x, pack.PackedConv("b", w, None, DEV, precision="bf16x3"), sc, sh, relu_in=True)
    d = (a - b).abs().max().item() / a.abs().max().item()
    print(f"PARITY conv bf16x3 vs fp32 MFMA kernel, 128->128 @512^2 x4: {d:.2e}")
    assert d < 1e-5


def test_conv_f16x2_contract_small_values_and_saturation(monkeypatch):
    """the fp16 two-term split: values far below 1 keep their fp32 accuracy in absolute terms (subnormal second terms lose
    nothing that matters); inputs beyond 65504 / in_scale saturate in the UNCHECKED kernel (EMO_F16X2_GUARD=0 semantics: no
    overflow word, no guarded recomputation) -- the range the device-side check of the default mode guards"""
    g = torch.Generator().manual_seed(6)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    layer = pack.PackedConv("s", w, None, DEV, cfg=3, precision="f16x2")
    x = torch.randn(1, 32, 8, 64, generator=g) * torch.logspace(-6, 1, 32).view(1, 32, 1, 1)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    ops.clear_overflow_flags(DEV)
    got = ops.conv_igemm(x.to(DEV), layer).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    assert ops.overflow_events(DEV) == {}
    monkeypatch.setattr(ops, "F16X2_GUARD", False)
    big = torch.full((1, 32, 8, 64), 3000.0)                   # 3000 * 32 > 65504: clipped to 2047
    got = ops.conv_igemm(big.to(DEV), layer).cpu()
    lim = 65504.0 / pack.F16X2_IN_SCALE
    ref_sat = F.conv2d(torch.full((1, 32, 8, 64), lim), w, padding=1)
    assert (got - ref_sat).abs().max().item() <= 1e-4 * ref_sat.abs().max().item()


@pytest.mark.parametrize("case", [dict(relu_in=True, affine=True, res=False, ksplit=None, ups=False),
                                  dict(relu_in=False, affine=False, res=True, ksplit=None, ups=True),
                                  dict(relu_in=True, affine=True, res=True, ksplit=3, ups=False)])
def test_conv_f16x2_overflow_is_detected_and_recomputed(case):
    """adversarial activations for the fp16 split (include/emo_hip.h, emo_conv_igemm_f16x2): one staged value beyond +-2047 raises
    the layer's overflow word on the device, and the guarded emo_conv_igemm_bf16x3 launch behind it rewrites output and tile
    statistics -- BIT-IDENTICAL to a plain bf16x3 launch of the layer.  In range, the word stays 0 and the result is the fp16
    split's.  No host synchronisation is involved in the decision."""
    g = torch.Generator().manual_seed(31)
    N, Cin, Cout, H, W = 2, 48, 128, 16, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(N, Cin, H, W, generator=g)
    ups = case["ups"]
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    sc = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV) if case["affine"] else None
    sh = (torch.randn(N, Cin, generator=g) * 0.2).to(DEV) if case["affine"] else None
    res = torch.randn(N, Cout, Ho, Wo, generator=g).to(DEV) if case["res"] else None
    l2 = pack.PackedConv("adv", w, b, DEV, cfg=3, precision="f16x2")
    l3 = pack.PackedConv("adv3", w, b, DEV, cfg=3, precision="bf16x3")
    kw = dict(relu_in=case["relu_in"], ups=ups, res=res, ksplit=case["ksplit"], want_stats=True)

    def run(layer, xin):
        out, st = ops.conv_igemm(xin.to(DEV), layer, sc, sh, **kw)
        return out.cpu(), (None if st is None else st.stats.cpu())

    ops.clear_overflow_flags(DEV)
    in_range, st_in = run(l2, x)
    assert ops.overflow_events(DEV) == {}, "in-range input raised the overflow word"
    exact, st_exact = run(l3, x)
    assert not torch.equal(in_range, exact)                       # (the fp16 split ran: its rounding differs from the bf16 split's)
    assert (in_range - exact).abs().max().item() <= 2e-5 * exact.abs().max().item()

    for big in (5000.0, -5000.0, float("inf")):
        if case["relu_in"] and big < 0 and case["affine"]:
            pass                                                  # (clamped away by the ReLU, still flagged: conservative)
        xa = x.clone()
        xa[1, 7, 5, 33] = big                                     # one element; halo of two tiles when H is tiled by 4
        ops.clear_overflow_flags(DEV)
        got, st_got = run(l2, xa)
        ev = ops.overflow_events(DEV)
        assert list(ev.values()) == ["adv"], (big, ev)
        want, st_want = run(l3, xa)
        assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(want, nan=7.0)), big
        if st_got is not None:
            assert torch.equal(torch.nan_to_num(st_got, nan=7.0), torch.nan_to_num(st_want, nan=7.0))
    # the word is sticky until cleared: the next (in-range) call still recomputes -- exact result, never a wrong one
    again, _ = run(l2, x)
    assert torch.equal(again, exact)
    ops.clear_overflow_flags(DEV)
    again, _ = run(l2, x)
    assert torch.equal(again, in_range)


# the two-tile kernel (csrc/conv_igemm_f16x2_ct2.h: two 64-channel output tiles per work item on one converted patch).  It takes a
# layer's channel-tile PAIRS when there are two items per CU; EMO_CONV_CT2_MIN_ITEMS=1 makes every eligible launch take it, so
# that small shapes exercise it too.  Cases: one pair; pairs + an odd last tile (192, 320 channels: the single-tile kernel runs
# that tile with cot0 set); one-stage and two-stage items; many items per block on a few samples (chained items, chains broken
# at sample boundaries); every residual form; the fused upsample; depth taps as stages
CT2_CASES = [
    dict(N=1, Cin=128, Cout=128, dims=(128, 128), k=3, cfg=3, affine=True, relu_in=True, res=True),
    dict(N=2, Cin=48, Cout=192, dims=(64, 64), k=3, cfg=3, affine=True, relu_in=True),
    dict(N=3, Cin=64, Cout=320, dims=(32, 64), k=3, cfg=3, affine=True, relu_in=True, res=True),
    dict(N=2, Cin=16, Cout=128, dims=(16, 64), k=3, cfg=3, bias=False),                                # one stage per item
    dict(N=2, Cin=32, Cout=256, dims=(8, 64), k=3, cfg=3, affine=True, relu_in=True),                 # two stages
    dict(N=2, Cin=40, Cout=128, dims=(64, 128), k=3, cfg=3, affine=True, relu_in=True, res=True),     # ragged channel group
    dict(N=2, Cin=64, Cout=192, dims=(32, 32), k=3, cfg=3, ups=True, affine=True, relu_in=True, res=True, res_ups=True),
    dict(N=1, Cin=96, Cout=128, dims=(64, 64), k=3, cfg=3, ups=True, affine=True, relu_in=True),
    dict(N=1, Cin=24, Cout=128, dims=(6, 64, 64), k=3, cfg=3, affine=True, relu_in=True, res=True),   # 3-D: depth taps as stages
    dict(N=5, Cin=64, Cout=128, dims=(128, 128), k=3, cfg=3, affine=True, relu_in=True),              # 1280 items: chains
]


@pytest.mark.parametrize("kernel", ["w8", "ct2"])
@pytest.mark.parametrize("case", CT2_CASES)
def test_conv_f16x2_two_tile_kernel_is_bit_identical_to_the_single_tile_kernel(case, kernel, monkeypatch):
    """same products in the same order into the same accumulator sets: the pair kernels -- conv_igemm_f16x2_w8_kernel (eight waves,
    two per SIMD: the default, round 6) and conv_igemm_bf16x3_ct2_kernel (four waves; EMO_CONV_W8=0) -- and the single-tile kernel
    must agree bit for bit, output and GroupNorm tile statistics; and the fp32 bound against torch CPU holds"""
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    monkeypatch.setenv("EMO_CONV_CT2", "1")
    monkeypatch.setenv("EMO_CONV_W8", "1" if kernel == "w8" else "0")
    e, got, ref = run_conv(seed=27, precision="f16x2", **case)
    print(f"PARITY conv f16x2 two-tile kernel [{kernel}]:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
    assert e < 2e-5, e
    e2, got2, _ = run_conv(seed=27, precision="f16x2", **case)
    assert torch.equal(got, got2), "two launches on the same input differ: a race in the pipeline"
    monkeypatch.setenv("EMO_CONV_CT2", "0")
    e1, single, _ = run_conv(seed=27, precision="f16x2", **case)
    assert torch.equal(got, single), (got - single).abs().max().item()


@pytest.mark.parametrize("kernel", ["w8", "ct2"])
def test_conv_f16x2_two_tile_kernel_statistics_and_range_check(kernel, monkeypatch):
    """tile statistics written by the pair kernels equal the single-tile kernel's bit for bit; an out-of-range input raises the
    layer's overflow word from the pair kernel as well, and the guarded bf16x3 launch rewrites the whole layer"""
    monkeypatch.setenv("EMO_CONV_W8", "1" if kernel == "w8" else "0")
    g = torch.Generator().manual_seed(33)
    N, Cin, Cout, H, W = 2, 48, 192, 32, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(N, Cin, H, W, generator=g)
    sc = (torch.rand(N, Cin, generator=g) + 0.5).to(DEV)
    sh = (torch.randn(N, Cin, generator=g) * 0.2).to(DEV)
    l2 = pack.PackedConv("ct2", w, b, DEV, cfg=3, precision="f16x2")
    l3 = pack.PackedConv("ct2_3", w, b, DEV, cfg=3, precision="bf16x3")

    def run(layer, xin):
        out, st = ops.conv_igemm(xin.to(DEV), layer, sc, sh, relu_in=True, want_stats=True)
        return out.cpu(), st.stats.cpu()

    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    ops.clear_overflow_flags(DEV)
    monkeypatch.setenv("EMO_CONV_CT2", "0")
    o1, s1 = run(l2, x)
    monkeypatch.setenv("EMO_CONV_CT2", "1")
    o2, s2 = run(l2, x)
    assert torch.equal(o1, o2) and torch.equal(s1, s2)
    assert ops.overflow_events(DEV) == {}
    xa = x.clone()
    xa[1, 7, 5, 33] = 5000.0
    got, sg = run(l2, xa)
    assert list(ops.overflow_events(DEV).values()) == ["ct2"]
    want, sw = run(l3, xa)
    assert torch.equal(got, want) and torch.equal(sg, sw)
    ops.clear_overflow_flags(DEV)


# pointwise layers on the fp16 split (csrc/conv_igemm_f16x2_p1.h): 32-channel stages, two channel tiles per item, an odd last tile
# in a half-empty pair, chained items; every residual form; affine + ReLU inputs; 3-D tensors as batches of planes
P1_CASES = [
    dict(N=1, Cin=64, Cout=128, dims=(64, 64), k=1, cfg=None),
    dict(N=2, Cin=192, Cout=320, dims=(32, 64), k=1, cfg=None, bias=False, res=True),                 # 5 channel tiles: half-empty pair
    dict(N=2, Cin=128, Cout=192, dims=(64, 128), k=1, cfg=None, affine=True, relu_in=True),
    dict(N=3, Cin=320, Cout=128, dims=(16, 64), k=1, cfg=None, res=True),
    dict(N=1, Cin=64, Cout=256, dims=(4, 64, 64), k=1, cfg=None, affine=True, relu_in=True, res=True), # 3-D
    dict(N=4, Cin=1536, Cout=512, dims=(64, 64), k=1, cfg=None, bias=False),                          # the decoder's entry convolution
]


@pytest.mark.parametrize("case", P1_CASES)
def test_conv_f16x2_pointwise_kernel(case, monkeypatch):
    monkeypatch.setenv("EMO_F16X2_P1_MIN_ITEMS", "1")
    e, got, ref = run_conv(seed=29, precision="f16x2", **case)
    print("PARITY conv f16x2 pointwise kernel:", case["Cin"], case["Cout"], case["dims"], f"{e:.2e}")
    assert e < 2e-5, e
    e2, got2, _ = run_conv(seed=29, precision="f16x2", **case)
    assert torch.equal(got, got2), "two launches on the same input differ: a race in the pipeline"
    e32, got32, _ = run_conv(seed=29, precision="f32", **case)
    assert not torch.equal(got, got32), "the fp32 MFMA kernel ran: the pointwise launch was not planned onto the split kernel"


def test_conv_f16x2_pointwise_plan_statistics_and_range_check(monkeypatch):
    """the planner routes a pointwise layer of an f16x2 model onto the split kernel only in its launch form (enough pair items, no
    activation, no fused upsample); its tile statistics (two half entries per 256-position tile, the fp32 MFMA kernel's
    128-position layout) give the GroupNorm affine of a direct reduction; an out-of-range input raises the layer's overflow word
    and the guarded fp32 MFMA launch rewrites output and statistics: bit-identical to a plain fp32 launch"""
    monkeypatch.setenv("EMO_F16X2_P1_MIN_ITEMS", "1")
    g = torch.Generator().manual_seed(35)
    N, Cin, Cout, H, W = 2, 64, 192, 32, 64
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    b = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(N, Cin, H, W, generator=g)
    l2 = pack.PackedConv("pw", w, b, DEV, precision="f16x2")
    l1 = pack.PackedConv("pw32", w, b, DEV, precision="f32")
    assert l2.pointwise_split and l2.plan_for(N * H * W // 128, H, W)[2] == "f16x2"
    assert l2.plan_for(N * H * W // 128, H, W, act="tanh")[2] == "f32" and l2.plan_for(N * H * W // 128, 2 * H, 2 * W, ups=True)[2] == "f32"
    monkeypatch.setenv("EMO_F16X2_P1_MIN_ITEMS", "512")
    assert l2.plan_for(N * H * W // 128, H, W)[2] == "f32"            # (too few items: the fp32 MFMA kernel, K split)
    monkeypatch.setenv("EMO_F16X2_P1_MIN_ITEMS", "1")
    with pytest.raises(ValueError):
        pack.PackedConv("bad", torch.zeros(64, 64, 1, 1), None, DEV, precision="f16x2")     # one channel tile: no pair
    ops.clear_overflow_flags(DEV)
    out, st = ops.conv_igemm(x.to(DEV), l2, want_stats=True)
    assert l2.last_plan[2] == "f16x2" and st is not None and ops.overflow_events(DEV) == {}
    assert rel_err(out, F.conv2d(x, w, b)) < 2e-5
    s1, h1 = ops.groupnorm_affine(out, stats=st)
    s0, h0 = ops.groupnorm_affine(out)
    assert (s1 - s0).abs().max().item() <= 2e-6 * s0.abs().max().item() and (h1 - h0).abs().max().item() <= 2e-6
    xa = x.clone()
    xa[1, 7, 5, 33] = 5000.0
    got, sg = ops.conv_igemm(xa.to(DEV), l2, want_stats=True)
    assert list(ops.overflow_events(DEV).values()) == ["pw"]
    want, sw = ops.conv_igemm(xa.to(DEV), l1, want_stats=True)
    assert torch.equal(got.cpu(), want.cpu()) and torch.equal(sg.stats.cpu(), sw.stats.cpu())
    ops.clear_overflow_flags(DEV)


@pytest.mark.parametrize("ksplit", [None, 3])
def test_conv_f16x2_guarded_recomputation_with_the_residual_updated_in_place(ksplit):
    """out aliases res (a residual updated in place, the form nets.ResBlock used for its convolved skip): the guarded bf16x3
    launch reads the residual AFTER the fp16-split launch has written the output.  ops.conv_igemm must not hand the first
    launch's result to the second as its residual (conv_exact + conv_clipped + res): bit-identical to a plain bf16x3 launch
    with a separate output, whether the range check fires or not"""
    g = torch.Generator().manual_seed(37)
    N, Cin, Cout, H, W = 2, 48, 128, 16, 64
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    x = torch.randn(N, Cin, H, W, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    l2 = pack.PackedConv("alias", w, b, DEV, cfg=3, precision="f16x2")
    l3 = pack.PackedConv("alias3", w, b, DEV, cfg=3, precision="bf16x3")
    xa = x.clone()
    xa[1, 7, 5, 33] = 5000.0
    want = ops.conv_igemm(xa.to(DEV), l3, res=r.to(DEV), ksplit=ksplit).cpu()
    ops.clear_overflow_flags(DEV)
    rd = r.to(DEV)
    got = ops.conv_igemm(xa.to(DEV), l2, res=rd, out=rd, ksplit=ksplit).cpu()
    assert list(ops.overflow_events(DEV).values()) == ["alias"]
    assert torch.equal(got, want)
    with pytest.raises(ValueError):                               # (a convolution over its own input is never valid)
        xd = xa.to(DEV)[:, :48].contiguous()
        ops.conv_igemm(xd, pack.PackedConv("inplace", w[:48], None, DEV, cfg=3, precision="f16x2"), out=xd)
    ops.clear_overflow_flags(DEV)


def test_overflow_flag_pool_is_one_per_device_whatever_the_spelling():
    """'cuda' and 'cuda:<current>' name the same pool (advisor, round 4: a HotPath built with device='cuda' cleared and
    reported a pool the launches never wrote)"""
    ops.clear_overflow_flags("cuda")
    slot = pack.overflow_flag_slot("cuda", "spelling")
    idx = torch.cuda.current_device()
    assert pack._flag_pool("cuda") is pack._flag_pool(f"cuda:{idx}") is pack._flag_pool(torch.device("cuda", idx))
    pack._flag_pool(f"cuda:{idx}")[0][slot] = 1
    assert ops.overflow_events("cuda").get(slot) == "spelling"
    ops.clear_overflow_flags("cuda")
    assert ops.overflow_events(f"cuda:{idx}") == {}


def test_conv_bf16x3_operand_contract():
    """non-finite and extreme operands of the bf16 split, as include/emo_hip.h states them: finite values up to the largest
    finite bf16 are split exactly (FLT_MAX-scale inputs give what the exact-fp32 kernel gives); +-inf SATURATE at +-3.39e38
    instead of turning into NaN through the residual inf - inf; a NaN is staged as the lower clamp bound (as in the fp32
    kernel); signed zeros and fp32 subnormals behave like the fp32 kernel to within 2^-126 per product"""
    g = torch.Generator().manual_seed(41)
    Cin, Cout, H, W = 16, 64, 4, 64
    w = torch.zeros(Cout, Cin, 3, 3)
    w[:, 0, 1, 1] = torch.linspace(-1, 1, Cout)                   # centre tap of channel 0 only: out[co] = w * x[0]
    l3 = pack.PackedConv("c3", w, None, DEV, cfg=3, precision="bf16x3")
    l1 = pack.PackedConv("c1", w, None, DEV, cfg=3, precision="f32")
    BF16_MAX = 3.3895313892515355e38

    def both(x, **kw):
        return ops.conv_igemm(x.to(DEV), l3, **kw).cpu(), ops.conv_igemm(x.to(DEV), l1, **kw).cpu()

    x = torch.zeros(1, Cin, H, W)
    x[0, 0, 1, 5], x[0, 0, 1, 6], x[0, 0, 2, 7] = 3.0e38, -3.0e38, BF16_MAX   # finite, at the top of the range: exact split
    a, b = both(x)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    x = torch.zeros(1, Cin, H, W)
    x[0, 0, 1, 5], x[0, 0, 2, 9] = float("inf"), float("-inf")
    a, b = both(x)
    assert not torch.isnan(a).any(), "inf - inf in the split residual"
    want = torch.zeros_like(a)
    want[0, :, 1, 5] = w[:, 0, 1, 1] * BF16_MAX
    want[0, :, 2, 9] = w[:, 0, 1, 1] * -BF16_MAX
    assert (a - want).abs().max().item() <= 1e-6 * BF16_MAX       # saturated, finite
    assert torch.isinf(b[0, -1, 1, 5])                            # the exact-fp32 kernel passes inf on (documented difference)
    x = torch.zeros(1, Cin, H, W)
    x[0, 0, 1, 5] = float("nan")
    a, b = both(x, relu_in=True)
    assert torch.equal(a, b) and not torch.isnan(a).any()         # NaN -> lower clamp bound (0 with ReLU) in both kernels
    x = torch.full((1, Cin, H, W), -0.0)
    x[0, 0, 1, 5], x[0, 0, 1, 6], x[0, 0, 1, 7] = 1e-40, -3e-39, 1.1754942e-38   # fp32 subnormals
    a, b = both(x)
    assert (a - b).abs().max().item() <= 2e-38 and (a.double() - b.double()).abs().max().item() <= 2 ** -126 * 1.01
