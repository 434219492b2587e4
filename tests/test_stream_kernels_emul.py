"""The barrier-free helper kernels of the hot path, run on the CPU from the product's own sources.

tests/emul/stream_kernels_emul.cpp compiles emoportraits_amd/csrc/resample.hip and conv_head.hip as host C++ (a launch becomes a
loop over blocks and threads: tests/emul/hipshim) and exports their C-ABI entry points on host memory.  What is checked here is
the kernels' index logic and arithmetic -- work decomposition, edge handling, operation order -- against torch's CPU operators,
without a GPU; the same entry points are checked on the GPU in tests/test_kernels_gpu.py.  The product never loads this library.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "emul", "stream_kernels_emul.cpp")
LIB = os.path.join(HERE, "emul", "_build", "libstream_emul.so")
ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}


@pytest.fixture(scope="module")
def lib():
    deps = [SRC, os.path.join(HERE, "emul", "hipshim", "hip", "hip_runtime.h")] + [
        os.path.join(ROOT, "emoportraits_amd", "csrc", f) for f in ("resample.hip", "conv_head.hip", "common.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(HERE, "emul", "hipshim"),
                        "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    return ctypes.CDLL(LIB)


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _aligned(n, offset_floats=0):
    """float32 buffer of n elements whose address is 16-byte aligned (+ offset_floats * 4 bytes)"""
    raw = np.empty(n + 8, np.float32)
    start = (-(raw.ctypes.data // 4) % 4 + offset_floats) % 4
    if offset_floats and start == 0:
        start = offset_floats
    out = raw[start:start + n]
    assert (out.ctypes.data % 16 == 0) == (offset_floats % 4 == 0)
    return out


def upsample(lib, x, factors, misalign=False):
    NC, (D, H, W) = x.shape[0], x.shape[1:]
    fd, fh, fw = factors
    out = _aligned(NC * D * fd * H * fh * W * fw, 1 if misalign else 0)
    xin = _aligned(x.size)
    xin[:] = x.ravel()
    rc = lib.emo_upsample_trilinear_f32(_p(xin), _p(out), ctypes.c_int64(NC), D, H, W, fd, fh, fw, None)
    assert rc == 0
    return out.reshape(NC, D * fd, H * fh, W * fw).copy()


@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2), (2, 1, 2), (1, 1, 2)])
@pytest.mark.parametrize("shape", [(10, 4, 6, 8), (3, 2, 4, 2), (2, 3, 5, 4), (8, 4, 16, 32), (2, 1, 1, 64), (1, 2, 2, 6), (64, 4, 4, 4),
                                   (2, 32, 64, 64)])
def test_block_upsampling_kernel_is_the_generic_kernel_bit_for_bit(lib, shape, factors):
    """upsample_trilinear_w2_kernel (one thread per block of up to 4 x 2 x 2 outputs; the first / last quad of a row through the
    same code with clamped columns) against the one-output-per-thread kernel (reached through an output pointer that is not
    16-byte aligned) and against ATen's CPU kernel: first / last pairs of every axis, runs of several volumes, split runs"""
    x = np.random.default_rng(5).standard_normal(shape).astype(np.float32)
    got = upsample(lib, x, factors)
    plain = upsample(lib, x, factors, misalign=True)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32))
    ref = F.interpolate(torch.from_numpy(x)[None], scale_factor=tuple(float(f) for f in factors), mode="trilinear")[0].numpy()
    assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2)])
@pytest.mark.parametrize("N,C,G,dims", [(2, 64, 32, (4, 4, 4)), (1, 32, 32, (2, 6, 8)), (1, 64, 32, (8, 32, 32)), (3, 8, 2, (2, 2, 2))])
def test_upsampling_with_groupnorm_sums_writes_the_same_tensor(lib, N, C, G, dims, factors):
    """the work decomposition of the fused entry point (runs = (sample, group), slices of a run): the TENSOR it writes is the
    plain call's bit for bit.  (Its sums pass through wave shuffles and a barrier, which this thread-by-thread run does not
    model: they are checked on the GPU, tests/test_kernels_gpu.py.)"""
    D, H, W = dims
    fd, fh, fw = factors
    x = np.random.default_rng(7).standard_normal((N * C, D, H, W)).astype(np.float32)
    plain = upsample(lib, x, factors)
    xin = _aligned(x.size)
    xin[:] = x.ravel()
    out = _aligned(plain.size)
    partial = np.zeros(N * G * 64 * 2, np.float64)
    split = ctypes.c_int(0)
    rc = lib.emo_upsample_trilinear_gn_sums_f32(_p(xin), _p(out), N, C, G, D, H, W, fd, fh, fw, _p(partial), ctypes.c_int64(partial.nbytes),
                                                ctypes.byref(split), None)
    assert rc == 0 and 1 <= split.value <= 64
    assert np.array_equal(out.view(np.uint32), plain.ravel().view(np.uint32))
    # odd widths / width factor 1 are refused (the caller runs the two operations one after the other)
    assert lib.emo_upsample_trilinear_gn_sums_f32(_p(xin), _p(out), N, C, G, D, H, W, fd, fh, 1, _p(partial), ctypes.c_int64(partial.nbytes),
                                                  ctypes.byref(split), None) == -2


@pytest.mark.parametrize("kernel", [(2, 1, 1), (1, 2, 2), (2, 2, 2), (1, 4, 4)])
def test_avgpool_add(lib, kernel):
    x = np.random.default_rng(2).standard_normal((6, 4, 8, 8)).astype(np.float32)
    kd, kh, kw = kernel
    out = np.empty((6, 4 // kd, 8 // kh, 8 // kw), np.float32)
    assert lib.emo_avgpool_f32(_p(x), _p(out), ctypes.c_int64(6), 4, 8, 8, kd, kh, kw, None) == 0
    ref = F.avg_pool3d(torch.from_numpy(x)[None], kernel, kernel)[0].numpy()
    assert np.abs(out - ref).max() < 1e-6
    b = np.random.default_rng(3).standard_normal(4 * 8 * 8).astype(np.float32)
    s = np.empty_like(x)
    assert lib.emo_add_f32(_p(x), _p(b), _p(s), ctypes.c_int64(x.size), ctypes.c_int64(b.size), ctypes.c_float(0.5), None) == 0
    assert np.array_equal(s, ((x + b.reshape(1, 4, 8, 8)) * np.float32(0.5)).astype(np.float32))


@pytest.mark.parametrize("bicubic", [0, 1])
def test_resize2d(lib, bicubic):
    x = np.random.default_rng(4).random((3, 20, 24)).astype(np.float32)
    out = np.empty((3, 13, 31), np.float32)
    rc = lib.emo_resize2d_f32(_p(x), ctypes.c_int64(20 * 24), ctypes.c_int64(24), _p(out), ctypes.c_int64(3), 20, 24, 13, 31, bicubic, 0, None)
    assert rc == 0
    ref = F.interpolate(torch.from_numpy(x)[None], size=(13, 31), mode="bicubic" if bicubic else "bilinear", align_corners=False)[0].numpy()
    assert np.abs(out - ref).max() < 2e-6


@pytest.mark.parametrize("act", ["none", "tanh", "sigmoid", "relu"])
@pytest.mark.parametrize("N,cin,cout,dims,affine,relu_in", [
    (2, 128, 3, (16, 24), True, True),         # the image head's form
    (3, 40, 4, (2, 6, 8), True, False),        # 3-D positions, a channel count off the unroll of 8, no ReLU
    (1, 7, 1, (4, 4), False, True),            # fewer channels than one unrolled group, no affine
    (2, 64, 2, (9, 256), False, False),        # more than one thread block per sample, the last one partly idle
])
def test_conv_head_stream_kernel(lib, N, cin, cout, dims, affine, relu_in, act):
    """conv_head_kernel (csrc/conv_head.hip) against torch's CPU convolution of the same operands: 2e-5 of the output's
    magnitude, the bound its GPU test and the implicit-GEMM kernel it replaces are held to"""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(N, cin, *dims, generator=g)
    w = torch.randn(cout, cin, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    scale = shift = None
    xin = x
    if affine:
        scale, shift = torch.rand(N, cin, generator=g) + 0.5, torch.randn(N, cin, generator=g) * 0.3
        bs = (N, cin) + (1,) * len(dims)
        xin = x * scale.view(bs) + shift.view(bs)
    if relu_in:
        xin = F.relu(xin)
    ref = torch.einsum("oc,nc...->no...", w.double(), xin.double()) + b.double().view((1, cout) + (1,) * len(dims))
    ref = {"none": lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": F.relu}[act](ref).numpy()
    S = int(np.prod(dims))
    xa = _aligned(x.numel())
    xa[:] = x.numpy().ravel()
    out = _aligned(N * cout * S)
    sc = None if scale is None else np.ascontiguousarray(scale.numpy())
    sh = None if shift is None else np.ascontiguousarray(shift.numpy())
    wn, bn = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(b.numpy())
    rc = lib.emo_conv_head_f32(_p(xa), _p(wn), _p(bn), _p(sc), _p(sh), _p(out), N, cin, cout, ctypes.c_int64(S), int(relu_in), ACT[act], None)
    assert rc == 0
    got = out.reshape(ref.shape)
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_conv_head_refuses_what_it_does_not_cover(lib):
    x, w, out = _aligned(16 * 8), np.zeros((5, 16), np.float32), _aligned(5 * 8)
    assert lib.emo_conv_head_f32(_p(x), _p(w), None, None, None, _p(out), 1, 16, 5, ctypes.c_int64(8), 0, 0, None) == -2      # 5 output channels
    assert lib.emo_conv_head_f32(_p(x), _p(w), None, None, None, _p(out), 1, 16, 3, ctypes.c_int64(6), 0, 0, None) == -2      # not whole quads
    assert lib.emo_conv_head_f32(_p(x[1:]), _p(w), None, None, None, _p(out), 1, 15, 3, ctypes.c_int64(8), 0, 0, None) == -3  # alignment
    assert lib.emo_conv_head_f32(_p(x), _p(w), None, _p(w), None, _p(out), 1, 16, 3, ctypes.c_int64(8), 0, 0, None) == -1     # scale without shift
