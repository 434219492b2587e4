"""The exact-fp32 implicit-GEMM convolution of the hot path (csrc/conv_igemm.h: every block config, 1x1 / 3x3 / 3x3x3 / 1x7
taps, fused GroupNorm affine + ReLU, nearest x2 upsample, residual forms, activations, K split, tile statistics), run on the CPU
from the product's own sources through its C ABI (emo_conv_igemm_f32) -- SURVEY.md section 8 rows a5 .. a10 without a GPU.

tests/emul/convlib.py compiles copies of the sources as host C++ (ROCm's clang++, the stand-in <hip/hip_runtime.h> of
tests/emul/hipshim in its threaded mode: the threads of a block are OS threads, __syncthreads is a barrier,
v_mfma_f32_32x32x2_f32 is an exchange between the 64 lanes of a wave with the hardware's operand / result layout, global_load_lds
copies into the block's LDS buffer) and rewrites in the copies what only the GPU toolchain understands: the inline-asm helpers
(pinned loads become plain loads, waits and scheduling fences nothing: the host run checks index arithmetic, LDS image, fragment
addressing and epilogue, not pipelining -- that is the ISA audit's and the GPU tests' job), the dynamic shared-memory
declaration and the occupancy attribute.  The product never loads this library, and the product sources are not touched.
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
from emoportraits_amd import pack  # noqa: E402

sys.path.insert(0, os.path.join(HERE, "emul"))
import convlib  # noqa: E402

ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}
CFG = {"A": 0, "B": 1, "C": 2, "D": 3, "E": 4, "F": 5}

pytestmark = pytest.mark.skipif(not convlib.available(), reason="needs ROCm clang++ and the built product library (weight packing asks it for the tile sizes)")


@pytest.fixture(scope="module")
def lib():
    return convlib.build()


def _buf(t):
    a = np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)
    raw = np.empty(a.size + 8, np.float32)
    off = (-(raw.ctypes.data // 4)) % 4
    out = raw[off:off + a.size].reshape(a.shape)
    out[...] = a
    return out


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def run_conv(lib, N, Cin, Cout, dims, k, cfg, affine=False, relu_in=False, ups=False, res=False, res_ups=False, bias=True, act="none",
             seed=0, ksplit=1, stats=False, k1x7=False):
    g = torch.Generator().manual_seed(seed)
    three_d = len(dims) == 3
    x = torch.randn(N, Cin, *dims, generator=g)
    kd = k if three_d else 1
    wshape = (Cout, Cin, k, k, k) if three_d else ((Cout, Cin, 7, 7) if k1x7 else (Cout, Cin, k, k))
    w = torch.randn(*wshape, generator=g) / math.sqrt(Cin * int(np.prod(wshape[2:])))
    b = torch.randn(Cout, generator=g) if bias else None
    scale = shift = None
    xin = x
    if affine:
        scale, shift = torch.rand(N, Cin, generator=g) + 0.5, torch.randn(N, Cin, generator=g) * 0.3
        bs = (N, Cin) + (1,) * len(dims)
        xin = x * scale.view(bs) + shift.view(bs)
    if relu_in:
        xin = F.relu(xin)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    padding = 3 if k1x7 else k // 2
    ref = (F.conv3d if three_d else F.conv2d)(xin.double(), w.double(), None if b is None else b.double(), padding=padding)
    r = None
    if res:
        rshape = list(ref.shape)
        if res_ups:
            rshape[-1] //= 2
            rshape[-2] //= 2
        r = torch.randn(*rshape, generator=g)
        ref = ref + (F.interpolate(r, scale_factor=2, mode="nearest") if res_ups else r).double()
    ref = {"none": lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": F.relu}[act](ref)
    D, H, W = (dims if three_d else (1,) + tuple(dims))
    kh, kw, KD = k, k, kd
    if k1x7:
        # a 7x7 2-D convolution runs as a depth-7 convolution over the image ROWS (encoder.py: from_rgb): weight [Co, Ci, 7, 1, 7],
        # the image viewed as [N, C, D = H, 1, W]
        w = w.unsqueeze(3)
        D, H, kh, kw, KD = H, 1, 1, 7, 7
    wpk = _buf(pack.pack_weight(w, cfg))
    xa, out = _buf(x), _buf(np.full(ref.shape, np.nan, np.float32))
    arr = lambda t: None if t is None else _buf(t)
    ba, sc, sh, ra = arr(b), arr(scale), arr(shift), arr(r)
    ws = _buf(np.zeros((ksplit, out.size), np.float32)) if ksplit > 1 else None
    lib.emo_conv_tile_positions.restype = ctypes.c_int
    cnt = lib.emo_conv_tile_positions(cfg)
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    st = _buf(np.full((N, D * Hl * Wl // cnt, Cout, 2), np.nan, np.float32)) if stats else None
    rc = lib.emo_conv_igemm_f32(_p(xa), _p(wpk), _p(ba), _p(sc), _p(sh), _p(ra), _p(out), N, Cin, Cout, D, H, W, KD, kh, kw, int(ups),
                                int(relu_in), ACT[act], int(res_ups), cfg, ksplit, _p(ws), _p(st), None)
    assert rc == 0, rc
    err = np.abs(out - ref.numpy()).max() / max(1.0, np.abs(ref.numpy()).max())
    return err, out, st, cnt


@pytest.mark.parametrize("cfg", ["A", "B", "C"])
@pytest.mark.parametrize("hw", [16, 32, 64, 128])          # (8-wide maps: a depth tile of two slices, test_3x3x3)
def test_3x3_every_tile_shape(lib, cfg, hw):
    err, *_ = run_conv(lib, 1 if hw > 32 else 2, 8, 40, (8 if hw > 32 else hw, hw), 3, CFG[cfg], seed=hw)
    assert err < 2e-5


@pytest.mark.parametrize("cfg,dims", [("D", (4, 64)), ("D", (8, 32)), ("D", (2, 128)), ("E", (4, 128)), ("F", (4, 64)), ("F", (2, 128))])
def test_3x3_wide_position_tiles(lib, cfg, dims):
    err, *_ = run_conv(lib, 1, 12, 72 if cfg != "F" else 24, dims, 3, CFG[cfg], affine=True, relu_in=True, res=True, seed=dims[1])
    assert err < 2e-5


@pytest.mark.parametrize("kw", [dict(affine=True, relu_in=True), dict(ups=True), dict(ups=True, res=True, res_ups=False), dict(res=True, res_ups=True, ups=True),
                                dict(act="tanh"), dict(act="sigmoid", bias=False), dict(ksplit=2), dict(affine=True, relu_in=True, ksplit=3, res=True)])
def test_3x3_launch_forms(lib, kw):
    """the fused input affine + ReLU, the nearest x2 upsample in the gather, both residual forms, the activations, K split"""
    err, *_ = run_conv(lib, 1 if kw.get("ups") else 2, 20, 40, (16, 16), 3, CFG["B"], seed=len(kw), **kw)
    assert err < 2e-5


@pytest.mark.parametrize("cfg", ["A", "B", "C"])
def test_1x1_and_1x7(lib, cfg):
    err, *_ = run_conv(lib, 1, 40, 70, (8, 32), 1, CFG[cfg], affine=True, relu_in=True, seed=3)      # 16-byte aligned: quad staging
    assert err < 2e-5
    err, *_ = run_conv(lib, 1, 24, 33, (2, 8, 16), 1, CFG[cfg], res=True, seed=4)                      # 3-D positions
    assert err < 2e-5
    if cfg != "C":
        err, *_ = run_conv(lib, 1, 3, 40, (6, 128), 7, CFG[cfg], k1x7=True, seed=5)                    # the encoder's 7x7 stem (rows of 128)
        assert err < 2e-5


@pytest.mark.parametrize("cfg,dims,cin,cout", [("B", (4, 8, 8), 8, 40), ("A", (2, 16, 16), 12, 72), ("D", (2, 8, 32), 8, 64), ("F", (3, 4, 64), 16, 32)])
def test_3x3x3(lib, cfg, dims, cin, cout):
    """3-D layers: a depth tile of two slices on 8-wide maps, depth taps as K stages on the wide tiles"""
    err, *_ = run_conv(lib, 1, cin, cout, dims, 3, CFG[cfg], affine=True, relu_in=True, res=True, seed=7)
    assert err < 2e-5


@pytest.mark.parametrize("cfg,dims", [("B", (16, 16)), ("D", (4, 64)), ("C", (8, 32))])
def test_tile_statistics_of_the_epilogue(lib, cfg, dims):
    """gn_stats [N][T][C][2] = (mean, centred sum of squares) of every tile's output values per channel: what
    emo_groupnorm_affine_from_tiles_f32 combines instead of reading the tensor"""
    err, out, st, cnt = run_conv(lib, 2, 8, 40, dims, 3, CFG[cfg], res=True, stats=True, seed=9)
    assert err < 2e-5
    N, C = out.shape[:2]
    o = torch.from_numpy(out.copy()).double().view(N, C, -1)
    s = torch.from_numpy(st.copy()).double()                                   # [N, T, C, 2]
    # recombine the tiles (Chan) and compare with the statistics of the whole channel
    mean = s[..., 0].mean(1)
    m2 = s[..., 1].sum(1) + cnt * ((s[..., 0] - mean[:, None]) ** 2).sum(1)
    assert (mean - o.mean(-1)).abs().max().item() < 1e-5
    assert (m2 - ((o - o.mean(-1, keepdim=True)) ** 2).sum(-1)).abs().max().item() < 1e-3 * m2.abs().max().item()
