"""The DOMINANT kernels of the hot path -- fp32 convolutions on the 16-bit matrix pipes (csrc/conv_igemm_bf16x3.h: the two-term
fp16 split with its device-side range check and the exact three-term bf16 split; csrc/conv_igemm_f16x2_ct2.h: two channel tiles
per work item; csrc/conv_igemm_f16x2_p1.h: pointwise layers; csrc/conv_igemm_f16.h: the opt-in fp16-operand mode) -- run on the
CPU from copies of the product's own sources through the C ABI, against fp64 convolutions.  SURVEY.md section 8 rows a5 .. a10.

tests/emul/convlib.py builds the library (ROCm's clang++, the stand-in runtime of tests/emul/hipshim in its threaded mode:
v_mfma_f32_32x32x16_{f16,bf16}, the DPP row operations of the statistics and LDS-DMA are modelled; pinned loads are plain loads,
`s_waitcnt; s_barrier` a barrier of the block).  What this checks without a GPU: operand conversion and range check, LDS images
and fragment addressing, the persistent item loop with chained items, channel-tile pairs on one converted patch, 32-row channel
tiles, depth taps as K stages, the fused upsample, the straight-line and the general epilogue, tile statistics, the guarded exact
recomputation.  What it cannot check: pipelining (a load is complete when issued here) -- the ISA audit's and the GPU tests' job.
"""
import ctypes
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "emul"))
import convlib  # noqa: E402
from emoportraits_amd import pack  # noqa: E402

pytestmark = pytest.mark.skipif(not convlib.available(), reason="needs ROCm clang++ and the built product library (weight packing asks it for tile sizes)")
ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}
CFG_D, CFG_F = 3, 5


@pytest.fixture(scope="module")
def lib():
    return convlib.build()


def _buf(t):
    a = np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else t)
    raw = np.empty(a.nbytes + 64, np.uint8)
    off = (-raw.ctypes.data) % 16
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _half_bits(t):
    return t.view(torch.int16) if t.dtype in (torch.float16, torch.bfloat16) else t


class Case:
    """one convolution: operands, fp64 reference, and launches of it in the split modes"""

    def __init__(self, N, Cin, Cout, dims, k=3, affine=True, relu_in=True, ups=False, res=False, res_ups=False, bias=True, seed=0, amp=1.0):
        g = torch.Generator().manual_seed(seed)
        self.three_d = len(dims) == 3
        self.N, self.Cin, self.Cout, self.k, self.ups, self.relu_in, self.res_ups = N, Cin, Cout, k, ups, relu_in, res_ups
        self.x = torch.randn(N, Cin, *dims, generator=g) * amp
        kd = k if self.three_d else 1
        wshape = (Cout, Cin, k, k, k) if self.three_d else (Cout, Cin, k, k)
        self.w = torch.randn(*wshape, generator=g) / math.sqrt(Cin * k * k * kd)
        self.b = torch.randn(Cout, generator=g) if bias else None
        self.scale = self.shift = None
        xin = self.x
        if affine:
            self.scale, self.shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g) * 0.3
            bs = (N, Cin) + (1,) * len(dims)
            xin = self.x * self.scale.view(bs) + self.shift.view(bs)
        if relu_in:
            xin = F.relu(xin)
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        ref = (F.conv3d if self.three_d else F.conv2d)(xin.double(), self.w.double(), None if self.b is None else self.b.double(), padding=k // 2)
        self.r = None
        if res:
            rshape = list(ref.shape)
            if res_ups:
                rshape[-1] //= 2
                rshape[-2] //= 2
            self.r = torch.randn(*rshape, generator=g)
            ref = ref + (F.interpolate(self.r, scale_factor=2, mode="nearest") if res_ups else self.r).double()
        self.ref = ref.numpy()
        self.D, self.H, self.W = (dims if self.three_d else (1,) + tuple(dims))
        self.kd = kd

    def launch(self, lib, mode, cfg=CFG_D, stats=False, flag=None, run_if=None, act="none", out=None):
        arr = lambda t: None if t is None else _buf(t)
        xa, ba, sc, sh, ra = _buf(self.x), arr(self.b), arr(self.scale), arr(self.shift), arr(self.r)
        out = _buf(np.full(self.ref.shape, np.nan, np.float32)) if out is None else out
        Hl, Wl = (2 * self.H, 2 * self.W) if self.ups else (self.H, self.W)
        st = None
        if stats:
            cnt = 128 if (mode == "f16x2" and self.k == 1) else 256
            st = _buf(np.full((self.N, self.D * Hl * Wl // cnt, self.Cout, 2), np.nan, np.float32))
        common = [_p(ba), _p(sc), _p(sh), _p(ra), _p(out), self.N, self.Cin, self.Cout, self.D, self.H, self.W, self.kd, self.k, self.k,
                  int(self.ups), int(self.relu_in), ACT[act], int(self.res_ups), cfg, 1, None, _p(st), None]
        if mode == "f16x2":
            flat, ws = (pack.pack_weight_f16x2_1x1(self.w) if self.k == 1 else pack.pack_weight_f16x2(self.w, bm=32 if cfg == CFG_F else None))
            wpk = _buf(_half_bits(flat))
            rc = lib.emo_conv_igemm_f16x2(_p(xa), _p(wpk), *common, ctypes.c_float(pack.F16X2_IN_SCALE), ctypes.c_float(ws), _p(flag))
        elif mode == "bf16x3":
            wpk = _buf(_half_bits(pack.pack_weight_bf16x3(self.w)))
            rc = lib.emo_conv_igemm_bf16x3(_p(xa), _p(wpk), *common, _p(run_if))
        elif mode == "f16":
            wpk = _buf(_half_bits(pack.pack_weight_f16(self.w, cfg)))
            rc = lib.emo_conv_igemm_f16acc32(_p(xa), _p(wpk), *common)
        elif mode == "f16w8":
            flat, ws = pack.pack_weight_f16w8(self.w)
            wpk = _buf(_half_bits(flat))
            rc = lib.emo_conv_igemm_f16w8(_p(xa), _p(wpk), *common, ctypes.c_float(ws))
        elif mode == "f16w8r":        # ABI 10: the pairs on the eight-wave kernel, the odd last tile on the older fp16-operand kernel
            flat, ws = pack.pack_weight_f16w8(self.w)
            wpk, wold = _buf(_half_bits(flat)), _buf(_half_bits(pack.pack_weight_f16(self.w, cfg)))
            rc = lib.emo_conv_igemm_f16w8_rest(_p(xa), _p(wpk), _p(wold), *common, ctypes.c_float(ws))
        else:
            raise ValueError(mode)
        assert rc == 0, (mode, rc)
        return out, st

    def err(self, out, act="none"):
        ref = {"none": lambda t: t, "tanh": np.tanh, "sigmoid": lambda t: 1 / (1 + np.exp(-t))}[act](self.ref)
        return np.abs(out - ref).max() / max(1.0, np.abs(ref).max())


def _bits(a):
    return a.view(np.uint32)


# ---- single-tile kernel, every position-tile shape, both splits ---------------------------------------------------------------
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("dims,N", [((32, 64), 1), ((16, 32), 1), ((16, 16), 2), ((4, 128), 1)])      # (32 x 64: 16 items, two chained per block)
def test_split_kernel_tile_shapes(lib, mode, dims, N):
    """4 x 64, 8 x 32, 16 x 16 position tiles; 40 input channels (a ragged last 16-channel stage), 72 output channels (a ragged
    second channel tile) -- a persistent block walks several chained items"""
    c = Case(N, 40, 72, dims, res=True, seed=dims[1])
    out, _ = c.launch(lib, mode)
    assert c.err(out) < 2e-5


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_split_kernel_fused_upsample_residual_forms_and_general_epilogue(lib, mode):
    c = Case(1, 32, 64, (4, 32), ups=True, res=True, res_ups=True, seed=1)             # nearest x2 in the gather, half-size residual
    assert c.err(c.launch(lib, mode)[0]) < 2e-5
    c = Case(1, 16, 64, (8, 64), ups=False, res=True, seed=2)
    assert c.err(c.launch(lib, mode, act="tanh")[0], "tanh") < 2e-5                   # an activation: the general epilogue
    c = Case(1, 16, 24, (8, 64), res=False, bias=False, affine=False, relu_in=False, seed=3)   # a 24-channel tile, plain operands
    assert c.err(c.launch(lib, mode)[0]) < 2e-5


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_split_kernel_3d_depth_taps_as_stages(lib, mode):
    c = Case(1, 16, 64, (3, 4, 64), res=True, seed=4)
    assert c.err(c.launch(lib, mode)[0]) < 2e-5
    c = Case(1, 24, 64, (2, 16, 16), res=False, seed=5)
    assert c.err(c.launch(lib, mode)[0]) < 2e-5


def test_split_kernel_tile_statistics(lib):
    """(mean, centred sum of squares) per 256-position tile and channel from the DPP reductions of the epilogue, recombined over the
    tiles (Chan) against the statistics of the whole channel; identical in both splits' layout"""
    c = Case(2, 16, 72, (8, 64), res=True, seed=6)
    for mode in ("f16x2", "bf16x3"):
        out, st = c.launch(lib, mode, stats=True)
        assert c.err(out) < 2e-5
        o = torch.from_numpy(out.copy()).double().view(2, 72, -1)
        s = torch.from_numpy(st.copy()).double()
        mean = s[..., 0].mean(1)
        m2 = s[..., 1].sum(1) + 256 * ((s[..., 0] - mean[:, None]) ** 2).sum(1)
        assert (mean - o.mean(-1)).abs().max().item() < 1e-5
        assert (m2 - ((o - o.mean(-1, keepdim=True)) ** 2).sum(-1)).abs().max().item() < 1e-3 * m2.abs().max().item()


# ---- two channel tiles per work item -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ["w8", "ct2"])
@pytest.mark.parametrize("cout,ups,N", [(128, False, 2), (192, False, 1), (128, True, 1)])
def test_two_tile_kernel_is_the_single_tile_kernel_bit_for_bit(lib, cout, ups, N, kernel, monkeypatch):
    """The two-tile kernels -- conv_igemm_f16x2_w8_kernel (eight waves, two per SIMD: the default) and conv_igemm_bf16x3_ct2_kernel
    (four waves; EMO_CONV_W8=0) -- (a pair of 64-channel tiles on one converted patch; an odd last tile on the single-tile kernel
    as a second launch) against the single-tile kernel on every tile (EMO_CONV_CT2=0): output, tile statistics and overflow word.
    N = 2: sixteen pair items on eight persistent blocks, two chained items each; N = 1: one item per block"""
    dims = (16, 32) if ups else (32, 64)                # eight position tiles per sample
    c = Case(N, 16, cout, dims, ups=ups, res=True, res_ups=False, seed=cout)
    flag_a, flag_b = _buf(np.zeros(4, np.int32)), _buf(np.zeros(4, np.int32))
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "8")
    monkeypatch.setenv("EMO_CONV_CT2", "1")
    monkeypatch.setenv("EMO_CONV_W8", "1" if kernel == "w8" else "0")
    out_a, st_a = c.launch(lib, "f16x2", stats=True, flag=flag_a)
    monkeypatch.setenv("EMO_CONV_CT2", "0")
    out_b, st_b = c.launch(lib, "f16x2", stats=True, flag=flag_b)
    assert c.err(out_a) < 2e-5
    assert np.array_equal(_bits(out_a), _bits(out_b)) and np.array_equal(_bits(st_a), _bits(st_b)) and np.array_equal(flag_a, flag_b)


W8_FORMS = [
    dict(N=2, Cin=40, Cout=128, dims=(16, 64), res=True),                                  # three stages (the last one ragged), chained items
    dict(N=1, Cin=16, Cout=128, dims=(32, 64), res=False),                                 # one stage per item (unchained), no residual
    dict(N=1, Cin=24, Cout=128, dims=(8, 32), ups=True, res=True, res_ups=True),           # fused upsample, half-size residual
    dict(N=1, Cin=16, Cout=256, dims=(3, 8, 64), res=True),                                # depth taps as K stages, two pairs
    dict(N=2, Cin=32, Cout=128, dims=(16, 64), res=True, affine=False, relu_in=False, bias=False),
    dict(N=1, Cin=32, Cout=128, dims=(16, 64), res=True, amp=3000.0),                      # out of range: the overflow word
]


@pytest.mark.parametrize("form", W8_FORMS)
def test_eight_wave_two_tile_kernel_launch_forms(lib, form, monkeypatch):
    """conv_igemm_f16x2_w8_kernel in every launch form of the decoders -- residual forms, fused upsample, depth taps, ragged last
    stage, one-stage items, chains across items and their breaks at sample boundaries, plain operands, an operand range violation
    -- bit for bit the single-tile kernel: output, tile statistics, overflow word"""
    kw = dict(form)
    N, Cin, Cout, dims = kw.pop("N"), kw.pop("Cin"), kw.pop("Cout"), kw.pop("dims")
    c = Case(N, Cin, Cout, dims, seed=Cin + Cout, **kw)
    flag_a, flag_b = _buf(np.zeros(4, np.int32)), _buf(np.zeros(4, np.int32))
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    monkeypatch.setenv("EMO_CONV_CT2", "1")
    monkeypatch.setenv("EMO_CONV_W8", "1")
    out_a, st_a = c.launch(lib, "f16x2", stats=True, flag=flag_a)
    monkeypatch.setenv("EMO_CONV_CT2", "0")
    out_b, st_b = c.launch(lib, "f16x2", stats=True, flag=flag_b)
    assert not np.isnan(out_a).any() and not np.isnan(st_a).any()
    if "amp" not in form:
        assert c.err(out_a) < 2e-5 and flag_a[0] == 0
    else:
        assert flag_a[0] != 0
    assert np.array_equal(_bits(out_a), _bits(out_b)) and np.array_equal(_bits(st_a), _bits(st_b)) and np.array_equal(flag_a, flag_b)


# ---- 32-row channel tiles, pointwise layers -----------------------------------------------------------------------------------
def test_32_row_channel_tiles(lib):
    c = Case(1, 32, 32, (2, 8, 64), res=True, seed=8)                                   # the WarpGenerator's 32 -> 32 3-D layer form
    out, st = c.launch(lib, "f16x2", cfg=CFG_F, stats=True)
    assert c.err(out) < 2e-5 and not np.isnan(st).any()
    c = Case(1, 32, 3, (2, 4, 64), res=False, seed=9)                                   # the warp head: 3 of 32 rows are real
    assert c.err(c.launch(lib, "f16x2", cfg=CFG_F, act="tanh")[0], "tanh") < 2e-5


def test_pointwise_kernel(lib):
    """conv_igemm_bf16x3_p1_kernel: 1 x 1 layers on the fp16 split, 32-channel stages, two channel tiles per item (192 channels:
    the third tile in a pair with zero weights), statistics as two half entries per 256-position tile"""
    c = Case(1, 64, 192, (16, 64), k=1, res=True, seed=10)
    out, st = c.launch(lib, "f16x2", stats=True)
    assert c.err(out) < 2e-5
    o = torch.from_numpy(out.copy()).double().view(1, 192, -1)
    s = torch.from_numpy(st.copy()).double()
    mean = s[..., 0].mean(1)
    m2 = s[..., 1].sum(1) + 128 * ((s[..., 0] - mean[:, None]) ** 2).sum(1)
    assert (mean - o.mean(-1)).abs().max().item() < 1e-5
    assert (m2 - ((o - o.mean(-1, keepdim=True)) ** 2).sum(-1)).abs().max().item() < 1e-3 * m2.abs().max().item()


# ---- range check and guarded recomputation -------------------------------------------------------------------------------------
def test_range_check_and_guarded_exact_recomputation(lib):
    """operands beyond the fp16 range of the scaled split raise the layer's overflow word; the guarded bf16x3 launch behind it
    recomputes the layer exactly; with the word down the guarded launch leaves the output alone"""
    c = Case(1, 16, 64, (8, 64), res=True, seed=11, amp=4000.0)
    flag = _buf(np.zeros(4, np.int32))
    out, _ = c.launch(lib, "f16x2", flag=flag)
    assert flag[0] != 0
    c.launch(lib, "bf16x3", run_if=flag, out=out)
    plain, _ = c.launch(lib, "bf16x3")
    assert np.array_equal(_bits(out), _bits(plain)) and c.err(out) < 2e-5
    calm = Case(1, 16, 64, (8, 64), res=True, seed=12)
    flag = _buf(np.zeros(4, np.int32))
    out, _ = calm.launch(lib, "f16x2", flag=flag)
    assert flag[0] == 0
    before = out.copy()
    calm.launch(lib, "bf16x3", run_if=flag, out=out)
    assert np.array_equal(_bits(out), _bits(before))


# ---- the opt-in fp16-operand mode (reduced precision: BASELINE configs[4]) ----------------------------------------------------
@pytest.mark.parametrize("k,dims", [(3, (8, 64)), (1, (8, 64)), (3, (2, 128))])
def test_fp16_operand_kernels(lib, k, dims):
    c = Case(1, 32, 72, dims, k=k, res=True, seed=13 + k)
    out, _ = c.launch(lib, "f16")
    assert c.err(out) < 2e-3                                                          # fp16 operands, fp32 accumulation


F16W8_FORMS = [
    dict(N=2, Cin=40, Cout=128, dims=(16, 64), res=True),                                  # chained items, a ragged last stage
    dict(N=1, Cin=16, Cout=192, dims=(32, 64), res=True),                                  # three channel tiles: a half-empty last pair
    dict(N=1, Cin=24, Cout=64, dims=(32, 64), res=False),                                  # ONE channel tile: every pair is half empty
    dict(N=1, Cin=24, Cout=128, dims=(8, 32), ups=True, res=True, res_ups=True),           # fused upsample, half-size residual
    dict(N=1, Cin=16, Cout=320, dims=(2, 8, 64), res=True),                                # depth taps, five tiles
]


@pytest.mark.parametrize("form", F16W8_FORMS)
def test_plain_fp16_operands_on_the_eight_wave_kernel(lib, form, monkeypatch):
    """emo_conv_igemm_f16w8 (conv_igemm_f16x2_w8.h, NPROD = 1: the reduced-precision mode of BASELINE configs[4] on the decoders'
    launch form) against an fp64 convolution at the fp16-operand bound, against the older fp16-operand kernel (the same operand
    rounding: they agree far inside that bound), with tile statistics; an odd last channel tile runs in a half-empty pair"""
    kw = dict(form)
    N, Cin, Cout, dims = kw.pop("N"), kw.pop("Cin"), kw.pop("Cout"), kw.pop("dims")
    c = Case(N, Cin, Cout, dims, seed=Cin + Cout, **kw)
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    out, st = c.launch(lib, "f16w8", stats=True)
    assert not np.isnan(out).any() and not np.isnan(st).any()
    assert c.err(out) < 2e-3
    old, _ = c.launch(lib, "f16")
    assert np.abs(out - old).max() <= 2e-5 * max(1.0, np.abs(old).max())
    o = torch.from_numpy(out.copy()).double().view(N, Cout, -1)
    s = torch.from_numpy(st.copy()).double()
    mean = s[..., 0].mean(1)
    m2 = s[..., 1].sum(1) + 256 * ((s[..., 0] - mean[:, None]) ** 2).sum(1)
    assert (mean - o.mean(-1)).abs().max().item() < 1e-5
    assert (m2 - ((o - o.mean(-1, keepdim=True)) ** 2).sum(-1)).abs().max().item() < 1e-3 * m2.abs().max().item()


F16W8_REST_FORMS = [
    dict(N=2, Cin=40, Cout=192, dims=(8, 64), res=True),                                   # three tiles: one pair + the last tile
    dict(N=1, Cin=16, Cout=320, dims=(4, 128), res=False),                                 # five tiles; the older kernel tiles 2 x 128
    dict(N=1, Cin=24, Cout=192, dims=(4, 32), ups=True, res=True, res_ups=True),           # fused upsample, half-size residual
    dict(N=1, Cin=16, Cout=192, dims=(2, 4, 64), res=True),                                # depth taps
]


@pytest.mark.parametrize("form", F16W8_REST_FORMS)
def test_plain_fp16_odd_tile_count_pairs_and_rest(lib, form, monkeypatch):
    """emo_conv_igemm_f16w8_rest (ABI 10): a layer with an odd number of channel tiles -- its pairs on the eight-wave kernel, its last
    tile on the older fp16-operand kernel (ConvArgs::cot0 there, ::cot_end here), every output element and every tile statistic
    written exactly once (the buffers start as NaN) -- against fp64 at the fp16-operand bound and against the one-launch form
    with the half-empty last pair"""
    kw = dict(form)
    N, Cin, Cout, dims = kw.pop("N"), kw.pop("Cin"), kw.pop("Cout"), kw.pop("dims")
    c = Case(N, Cin, Cout, dims, seed=Cin + Cout + 1, **kw)
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    out, st = c.launch(lib, "f16w8r", stats=True)
    assert not np.isnan(out).any() and not np.isnan(st).any()
    assert c.err(out) < 2e-3
    one, st1 = c.launch(lib, "f16w8", stats=True)
    assert np.abs(out - one).max() <= 2e-5 * max(1.0, np.abs(one).max())
    assert np.array_equal(out[:, :Cout - 64], one[:, :Cout - 64])                          # the pairs: the same kernel, the same items
    o = torch.from_numpy(out.copy()).double().view(N, Cout, -1)
    s = torch.from_numpy(st.copy()).double()
    mean = s[..., 0].mean(1)
    m2 = s[..., 1].sum(1) + 256 * ((s[..., 0] - mean[:, None]) ** 2).sum(1)
    assert (mean - o.mean(-1)).abs().max().item() < 1e-5
    assert (m2 - ((o - o.mean(-1, keepdim=True)) ** 2).sum(-1)).abs().max().item() < 1e-3 * m2.abs().max().item()


def test_plain_fp16_pairs_and_rest_declines_even_and_single_tile_counts(lib, monkeypatch):
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    for cout in (64, 128):
        c = Case(1, 16, cout, (4, 64), res=False, seed=3)
        wpk = _buf(_half_bits(pack.pack_weight_f16w8(c.w)[0]))
        wold = _buf(_half_bits(pack.pack_weight_f16(c.w, CFG_D)))
        out = _buf(np.zeros(c.ref.shape, np.float32))
        args = [_p(_buf(c.x)), _p(wpk), _p(wold), None, None, None, None, _p(out), 1, 16, cout, 1, 4, 64, 1, 3, 3, 0, 1, 0, 0, CFG_D, 1, None, None, None]
        assert lib.emo_conv_igemm_f16w8_rest(*args, ctypes.c_float(1.0)) == -2             # EMO_ERR_UNSUPPORTED


def test_plain_fp16_eight_wave_kernel_declines_other_launch_forms(lib):
    c = Case(1, 16, 64, (16, 32), res=False, seed=1)                                      # 8 x 32 position tiles
    wpk = _buf(_half_bits(pack.pack_weight_f16w8(c.w)[0]))
    out = _buf(np.zeros(c.ref.shape, np.float32))
    args = [_p(_buf(c.x)), _p(wpk), None, None, None, None, _p(out), 1, 16, 64, 1, 16, 32, 1, 3, 3, 0, 1, 0, 0, CFG_D, 1, None, None, None]
    assert lib.emo_conv_igemm_f16w8(*args, ctypes.c_float(1.0)) == -2                      # EMO_ERR_UNSUPPORTED


# ---- seeded random launch forms --------------------------------------------------------------------------------------------------
def test_random_launch_forms_of_the_split_kernels(lib, monkeypatch):
    """8 seeded random combinations of split, tile shape, 2-D / 3-D, channel counts (ragged stages and tiles), upsample, residual
    form, affine, ReLU, bias, activation, tile statistics and batch -- with channel-tile pairs always on the two-tile kernel.
    (Two ten-minute runs of the same generator, 721 cases, found no mismatch: tools of the round, not kept.)"""
    import random
    monkeypatch.setenv("EMO_CONV_CT2_MIN_ITEMS", "1")
    rnd = random.Random(2025)
    done = 0
    while done < 8:
        mode = rnd.choice(["f16x2", "f16x2", "bf16x3"])
        three = rnd.random() < 0.3
        tile = rnd.choice([(4, 64), (8, 32), (16, 16), (2, 128)])
        rows = tile[0] * rnd.choice([1, 2])
        ups = (not three) and tile[1] in (64, 128) and rows % 2 == 0 and rnd.random() < 0.25
        dims = (rnd.choice([1, 2, 3]), rows, tile[1]) if three else ((rows // 2, tile[1] // 2) if ups else (rows, tile[1]))
        Cin, Cout = rnd.choice([8, 16, 24, 40]), rnd.choice([8, 24, 64, 72, 128, 136])
        res = rnd.random() < 0.6
        kw = dict(affine=rnd.random() < 0.7, relu_in=rnd.random() < 0.7, ups=ups, res=res, res_ups=res and ups and rnd.random() < 0.5,
                  bias=rnd.random() < 0.8, seed=done)
        stats, act, N = rnd.random() < 0.4, rnd.choice(["none", "none", "tanh"]), rnd.choice([1, 2])
        Hl, Wl = (2 * dims[-2], 2 * dims[-1]) if ups else dims[-2:]
        if (Wl % 64 == 0 and Hl % 4) or (Wl == 32 and Hl % 8) or (Wl == 16 and Hl % 16):
            continue
        c = Case(N, Cin, Cout, dims, **kw)
        out, st = c.launch(lib, mode, stats=stats and (c.D * Hl * Wl) % 256 == 0, act=act)
        assert c.err(out, act) < 2e-5 and not np.isnan(out).any() and (st is None or not np.isnan(st).any()), (mode, N, Cin, Cout, dims, kw, act)
        done += 1


# ---- persistent grids on a part whose CU count is not a multiple of 8 ----------------------------------------------------------
@pytest.mark.parametrize("cus", ["12", "4", "1"])
def test_persistent_grid_covers_every_item_whatever_the_cu_count(lib, cus):
    """the split kernels hand out their items in eight per-XCD ranges (block b walks the range of XCD b % 8): emo_cu_count()
    (csrc/common.h) rounds the device's CU count down to a multiple of 8, at least 8, so that min(items, CUs) blocks hold all
    eight residues.  With the raw count a 12-, 4- or 1-CU device left whole ranges uncomputed (NaN fill below) -- found by this
    emulation.  The count is cached per process: a child process per device"""
    import subprocess
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_conv_split_emul as T\n"
        "lib = T.convlib.build()\n"
        "for mode, dims in (('f16x2', (32, 64)), ('bf16x3', (16, 64))):\n"
        "    c = T.Case(1, 16, 72, dims, res=True, seed=3)\n"
        "    out, _ = c.launch(lib, mode)\n"
        "    assert not np.isnan(out).any() and c.err(out) < 2e-5, (mode, float(np.isnan(out).mean()))\n"
        "os.environ['EMO_CONV_CT2_MIN_ITEMS'] = '1'\n"
        "c = T.Case(2, 16, 128, (32, 64), res=True, seed=4)\n"
        "out, _ = c.launch(lib, 'f16x2')\n"
        "assert not np.isnan(out).any() and c.err(out) < 2e-5\n"
        "print('ok')\n") % (HERE, ROOT)
    env = dict(os.environ, HIPSHIM_CUS=cus)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
