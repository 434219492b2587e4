"""The embedders' convolution kernel (csrc/conv_generic.hip: im2col gather with the producer's norm + ReLU folded in, LDS tiles,
v_mfma_f32_32x32x2_f32, split K with a fixed-order reduction) run on the CPU from the product's own source -- SURVEY.md section 8
row f1 without a GPU.

Compiled as host C++ by ROCm's clang++ against the stand-in <hip/hip_runtime.h> of tests/emul/hipshim in its threaded mode: the
threads of a block are OS threads, __syncthreads is a barrier, and the MFMA instruction is an exchange between the 64 lanes of a
wave with the operand / result layout of the hardware instruction (which is what the kernel's fragment addressing and its
epilogue rely on).  Compared with torch's CPU convolution on the layer forms of the torchvision ResNets
(identity_embedder.py:59-69, expression_embedder.py:424-439, head_pose_regressor.py:21-32).  The product never loads this library.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from emoportraits_amd import pack  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SHIM = os.path.join(HERE, "emul", "hipshim")
SRC = os.path.join(ROOT, "emoportraits_amd", "csrc", "conv_generic.hip")

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="ROCm clang++ not installed")


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(HERE, "emul", "_build", "libconv_generic_emul_threads.so")
    deps = [SRC, os.path.join(SHIM, "hip", "hip_runtime.h"), os.path.join(ROOT, "emoportraits_amd", "csrc", "common.h")]
    if not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-DHIPSHIM_THREADS", "-pthread", "-I" + SHIM, "-w", "-shared", "-fPIC",
                        "-o", out, "-x", "c++", SRC], check=True)
    return ctypes.CDLL(out)


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("N,cin,hw,cout,k,stride,pad,affine,relu,bias,splits", [
    (2, 3, (18, 20), 64, 7, 2, 3, False, False, False, 1),       # the stem: 7x7 / 2
    (2, 16, (9, 10), 40, 3, 1, 1, True, True, True, 1),          # BasicBlock conv with the producer's norm + ReLU folded in; ragged Cout
    (3, 24, (8, 8), 72, 3, 2, 1, True, True, False, 3),          # the strided 3x3 of a stage's first block, K split three ways
    (2, 32, (8, 8), 130, 1, 2, 0, False, False, True, 1),        # 1x1 / 2 downsample branch, three channel tiles
    (5, 40, (4, 4), 9, 4, 1, 0, True, False, True, 4),           # a 4x4 map consumed whole (the fc as a convolution), split K, 9 outputs
    (1, 8, (5, 7), 16, 3, 1, 1, False, False, False, 2),         # fewer columns than one tile
])
def test_generic_convolution_kernel(lib, N, cin, hw, cout, k, stride, pad, affine, relu, bias, splits):
    g = torch.Generator().manual_seed(cin * k + cout)
    x = torch.randn(N, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    xin = x
    sc = sh = None
    if affine:
        sc, sh = torch.rand(N, cin, generator=g) + 0.5, torch.randn(N, cin, generator=g) * 0.3
        xin = x * sc.view(N, cin, 1, 1) + sh.view(N, cin, 1, 1)
        if relu:
            xin = F.relu(xin)
    ref = F.conv2d(xin.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad).numpy()
    wt = np.ascontiguousarray(pack.pack_generic(w).numpy())
    arr = lambda t: None if t is None else np.ascontiguousarray(t.numpy())
    out = np.full(ref.shape, np.nan, np.float32)
    lib.emo_conv2d_generic_splits.restype = ctypes.c_int
    want = lib.emo_conv2d_generic_splits(N, cin, hw[0], hw[1], cout, k, k, stride, pad)
    assert want >= 1
    ws = np.full((splits, out.size), np.nan, np.float32) if splits > 1 else None
    rc = lib.emo_conv2d_generic_f32(_p(arr(x)), _p(wt), _p(arr(b)), _p(arr(sc)), _p(arr(sh)), _p(out), N, cin, hw[0], hw[1], cout, k, k,
                                    stride, pad, int(relu), splits, _p(ws), None)
    assert rc == 0
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_generic_convolution_refusals(lib):
    x, wt, out = np.zeros((1, 4, 6, 6), np.float32), np.zeros((36, 64), np.float32), np.zeros((1, 8, 6, 6), np.float32)
    call = lambda *a: lib.emo_conv2d_generic_f32(*a)
    assert call(_p(x), _p(wt), None, _p(x), None, _p(out), 1, 4, 6, 6, 8, 3, 3, 1, 1, 0, 1, None, None) == -1      # scale without shift
    assert call(_p(x), _p(wt), None, None, None, _p(out), 1, 4, 6, 6, 8, 3, 3, 1, 1, 0, 2, None, None) == -1       # split K without a workspace
    assert call(_p(x), _p(wt), None, None, None, _p(out), 1, 4, 2, 2, 8, 7, 7, 1, 0, 0, 1, None, None) == -1       # window larger than the map
