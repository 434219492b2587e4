"""The barrier-free helper kernels of the hot path, run on the CPU from the product's own sources.

emoportraits_amd/csrc/{resample,conv_head,embed_ops,smallops,groupnorm}.hip -- the product's sources, untouched -- are compiled
as host C++ against a stand-in <hip/hip_runtime.h> (tests/emul/hipshim) and export their C-ABI entry points on host memory.  Two
builds: a launch as a plain loop over blocks and threads (barrier-free kernels), and the threads of a block as OS threads with
real barriers and wave shuffles (the block reductions of the GroupNorm statistics, the wave-reduced GEMM).  What is checked is
the kernels' index logic and arithmetic -- work decomposition, edge handling, operation order -- against torch's CPU operators
and the reference's golden vectors, without a GPU; the same entry points are checked on the GPU in tests/test_kernels_gpu.py and
tests/test_embedders_gpu.py.  The product never loads these libraries.
"""
import ctypes
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "emoportraits_amd", "csrc")
SHIM = os.path.join(HERE, "emul", "hipshim")
SOURCES = [os.path.join(CSRC, f) for f in ("resample.hip", "conv_head.hip", "embed_ops.hip", "smallops.hip", "groupnorm.hip")]
ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}
sys.path.insert(0, os.path.join(HERE, "emul"))
import emulibs  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    """sequential mode: one GPU thread after the other (barrier-free kernels)"""
    return emulibs.stream(False)


@pytest.fixture(scope="module")
def tlib():
    """threaded mode: the threads of a block are OS threads, __syncthreads / __shfl_* are real exchanges (block reductions)"""
    return emulibs.stream(True)


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
                                   (2, 32, 64, 64), (2, 3, 20, 12), (1, 5, 30, 10)])      # (quads per row that do not divide 256)
def test_block_upsampling_kernel_is_the_generic_kernel_bit_for_bit(lib, shape, factors):
    """upsample_trilinear_w2_kernel (one thread per block of up to 4 x 2 x 2 outputs; the first / last quad of a row through the
    same code with clamped columns) against the one-output-per-thread kernel (reached through an output pointer that is not
    16-byte aligned) and against ATen's CPU kernel: first / last pairs of every axis, runs of several volumes, split runs.
    FINITE inputs: on a factor-1 axis the block kernel copies tap 0 into tap 1 instead of loading v[i1] for a weight of 0, so a
    non-finite neighbour does not propagate as it does through `1 * v[i0] + 0 * v[i1]` (csrc/resample.hip)"""
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


@pytest.mark.parametrize("kernel", [(2, 1, 1), (1, 2, 2), (2, 2, 2), (1, 4, 4), (3, 1, 2)])
@pytest.mark.parametrize("dims", [(6, 8, 8), (6, 4, 16), (12, 6, 40)])
def test_avgpool_four_outputs_per_thread_is_the_generic_kernel_bit_for_bit(lib, kernel, dims):
    """avgpool_x4_kernel (window width 1 or 2 on rows of whole quads) against the one-output-per-thread kernel (reached through
    an input pointer that is not 16-byte aligned) and against ATen's CPU kernel"""
    D, H, W = dims
    kd, kh, kw = kernel
    if D % kd or H % kh or W % kw:
        pytest.skip("window does not tile the volume")
    x = np.random.default_rng(2).standard_normal((5, D, H, W)).astype(np.float32)
    xa, xm = _aligned(x.size), _aligned(x.size, 1)
    xa[:] = x.ravel()
    xm[:] = x.ravel()
    oshape = (5, D // kd, H // kh, W // kw)
    out, plain = _aligned(int(np.prod(oshape))), _aligned(int(np.prod(oshape)))
    assert lib.emo_avgpool_f32(_p(xa), _p(out), ctypes.c_int64(5), D, H, W, kd, kh, kw, None) == 0
    assert lib.emo_avgpool_f32(_p(xm), _p(plain), ctypes.c_int64(5), D, H, W, kd, kh, kw, None) == 0
    assert np.array_equal(out.view(np.uint32), plain.view(np.uint32))
    ref = F.avg_pool3d(torch.from_numpy(x)[None], kernel, kernel)[0].numpy()
    assert np.abs(out.reshape(oshape) - ref).max() < 1e-6


def test_add(lib):
    x = np.random.default_rng(2).standard_normal((6, 4, 8, 8)).astype(np.float32)
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


@pytest.mark.parametrize("bicubic", [0, 1])
def test_resize2d_windows_is_the_single_window_kernel_bit_for_bit(lib, bicubic):
    """emo_resize2d_windows_f32 (ABI 9: one crop window per frame of a batch, one launch) against one emo_resize2d_f32 call per
    frame on that frame's window, and against F.interpolate on the cropped frame"""
    rng = np.random.default_rng(6)
    N, C, H, W, Ho, Wo = 3, 3, 40, 56, 16, 16
    x = rng.random((N, C, H, W)).astype(np.float32)
    wins = np.array([[10, 5, 30, 30], [0, 0, 56, 40], [33, 17, 8, 12]], dtype=np.int32)          # (x0, y0, w, h)
    out = np.empty((N, C, Ho, Wo), np.float32)
    rc = lib.emo_resize2d_windows_f32(_p(x), ctypes.c_int64(H * W), ctypes.c_int64(W), _p(wins), _p(out), N, C, Ho, Wo, bicubic, 1, None)
    assert rc == 0
    for i, (x0, y0, w, h) in enumerate(wins.tolist()):
        one = np.empty((C, Ho, Wo), np.float32)
        first = ctypes.c_void_p(x[i].ctypes.data + 4 * (y0 * W + x0))
        assert lib.emo_resize2d_f32(first, ctypes.c_int64(H * W), ctypes.c_int64(W), _p(one), ctypes.c_int64(C), h, w, Ho, Wo, bicubic, 1, None) == 0
        assert np.array_equal(one.view(np.uint32), out[i].view(np.uint32))
        ref = F.interpolate(torch.from_numpy(x[i:i + 1, :, y0:y0 + h, x0:x0 + w].copy()), size=(Ho, Wo),
                            mode="bicubic" if bicubic else "bilinear", align_corners=False).clamp(0, 1)[0].numpy()
        assert np.abs(out[i] - ref).max() < 2e-6
    assert lib.emo_resize2d_windows_f32(_p(x), ctypes.c_int64(H * W), ctypes.c_int64(W), None, _p(out), N, C, Ho, Wo, bicubic, 1, None) == -1


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


# ---- embedder helpers (csrc/embed_ops.hip) and the small operators (csrc/smallops.hip), sequential build ----------------------
@pytest.mark.parametrize("k,stride,pad,relu,affine", [(3, 2, 1, True, True), (2, 2, 0, False, False), (3, 1, 1, False, True)])
def test_maxpool2d_with_folded_norm(lib, k, stride, pad, relu, affine):
    g = torch.Generator().manual_seed(k + stride)
    x = torch.randn(6, 11, 14, generator=g)
    sc, sh = (torch.rand(6, generator=g) + 0.5, torch.randn(6, generator=g)) if affine else (None, None)
    xin = x * sc.view(6, 1, 1) + sh.view(6, 1, 1) if affine else x
    xin = F.relu(xin) if relu else xin
    ref = F.max_pool2d(xin[None], k, stride, pad)[0].numpy()
    out = np.empty(ref.shape, np.float32)
    xn = np.ascontiguousarray(x.numpy())
    rc = lib.emo_maxpool2d_f32(_p(xn), _p(None if sc is None else sc.numpy()), _p(None if sh is None else sh.numpy()), _p(out),
                               ctypes.c_int64(6), 11, 14, k, stride, pad, int(relu), None)
    assert rc == 0 and np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("with_b,relu", [(True, True), (True, False), (False, True)])
def test_affine_add_relu(lib, with_b, relu):
    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(10, 33, generator=g), torch.randn(10, 33, generator=g)
    sa, ta, sb, tb = (torch.randn(10, generator=g) for _ in range(4))
    ref = a * sa[:, None] + ta[:, None]
    if with_b:
        ref = ref + (b * sb[:, None] + tb[:, None])
    ref = F.relu(ref) if relu else ref
    out = np.empty((10, 33), np.float32)
    arr = lambda t: np.ascontiguousarray(t.numpy())
    rc = lib.emo_affine_add_relu_f32(_p(arr(a)), _p(arr(sa)), _p(arr(ta)), _p(arr(b)) if with_b else None, _p(arr(sb)) if with_b else None,
                                     _p(arr(tb)) if with_b else None, _p(out), ctypes.c_int64(10), ctypes.c_int64(33), int(relu), None)
    assert rc == 0 and np.abs(out - ref.numpy()).max() < 1e-5


def test_grid_sample2d_explicit_grid_and_theta(lib):
    """F.grid_sample 4-D bilinear / zeros / align_corners=False with an explicit grid; and the affine form of the expression
    embedder (expression_embedder.py:221-231): grid = theta @ (lin[x], lin[y], 1)"""
    g = torch.Generator().manual_seed(4)
    img = torch.randn(2, 3, 9, 12, generator=g)
    grid = torch.rand(2, 7, 5, 2, generator=g) * 2.4 - 1.2
    ref = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False).numpy()
    out = np.empty(ref.shape, np.float32)
    arr = lambda t: np.ascontiguousarray(t.numpy())
    assert lib.emo_grid_sample2d_f32(_p(arr(img)), _p(arr(grid)), None, None, _p(out), None, 2, 3, 9, 12, 7, 5, None) == 0
    assert np.abs(out - ref).max() < 1e-5
    theta = torch.tensor([[[0.9, 0.1, 0.05], [-0.1, 1.1, -0.02]], [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    lin = torch.linspace(-1, 1, 8)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    base = torch.stack([xx, yy, torch.ones_like(xx)], -1).view(1, 64, 3)
    tg = base.expand(2, -1, -1).bmm(theta.transpose(1, 2)).view(2, 8, 8, 2)
    ref = F.grid_sample(img, tg, mode="bilinear", padding_mode="zeros", align_corners=False).numpy()
    out, gout = np.empty(ref.shape, np.float32), np.empty((2, 8, 8, 2), np.float32)
    assert lib.emo_grid_sample2d_f32(_p(arr(img)), None, _p(arr(theta)), _p(arr(lin)), _p(out), _p(gout), 2, 3, 9, 12, 8, 8, None) == 0
    assert np.abs(gout - tg.numpy()).max() < 1e-6 and np.abs(out - ref).max() < 1e-5


def test_pose_theta_against_the_reference_golden(lib):
    """utils/point_transforms.py:188-242 get_transform_matrix: tests/golden/pose_theta.npz holds the reference's own output"""
    gz = np.load(os.path.join(HERE, "golden", "pose_theta.npz"))
    scale, rot, tr, want = (np.ascontiguousarray(gz[k], dtype=np.float32) for k in ("scale", "rotation", "translation", "theta"))
    B = want.shape[0]
    out = np.empty((B, 4, 4), np.float32)
    assert lib.emo_pose_theta_f32(_p(scale), scale.shape[1] if scale.ndim == 2 else 1, _p(rot), _p(tr), _p(out), B, None) == 0
    assert np.abs(out - want.reshape(B, 4, 4)).max() < 1e-5
    inv = np.empty_like(out)
    assert lib.emo_mat4_inverse_f32(_p(out), _p(inv), B, None) == 0
    assert np.abs(inv - np.linalg.inv(out.astype(np.float64))).max() < 1e-4


def test_projector_finalize_and_rgb8_packing(lib):
    rng = np.random.default_rng(1)
    B, R, E, NN = 3, 10, 16, 2
    T, V = rng.standard_normal((B, R, E)).astype(np.float32), rng.standard_normal((NN, E, 2)).astype(np.float32)
    nor = np.array([0] * 6 + [1] * 4, np.int32)
    gamma, beta = rng.standard_normal(R).astype(np.float32), rng.standard_normal(R).astype(np.float32)
    ag, ab = np.empty((B, R), np.float32), np.empty((B, R), np.float32)
    assert lib.emo_projector_finalize_f32(_p(T), _p(V), _p(nor), _p(gamma), _p(beta), _p(ag), _p(ab), B, R, E, None) == 0
    d = np.einsum("bre,re->br", T.astype(np.float64), V[nor][:, :, 0]), np.einsum("bre,re->br", T.astype(np.float64), V[nor][:, :, 1])
    assert np.abs(ag - (gamma + d[0])).max() < 1e-5 and np.abs(ab - (beta + d[1])).max() < 1e-5
    img = (rng.random((2, 3, 5, 8)) * 1.4 - 0.2).astype(np.float32)
    u8 = np.empty((2, 5, 8, 3), np.uint8)
    assert lib.emo_pack_rgb8(_p(img), _p(u8), 2, 5, 8, None) == 0
    want = (torch.from_numpy(img).clamp(0, 1).mul(255).byte()).permute(0, 2, 3, 1).numpy()       # ToPILImage: mul(255).byte()
    assert np.array_equal(u8, want)
    back = np.empty((2, 3, 5, 8), np.float32)
    assert lib.emo_unpack_rgb8(_p(u8), _p(back), 2, 5, 8, None) == 0
    assert np.array_equal(back, (u8.astype(np.float32) / np.float32(255)).transpose(0, 3, 1, 2))


# ---- block reductions, threaded build -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,C,G,S,ada", [(2, 8, 4, 96, True), (1, 6, 3, 37, False), (2, 4, 2, 5000, True)])
def test_groupnorm_affine_block_reduction(tlib, N, C, G, S, ada):
    """gn_partial_kernel (wave shuffles + shared memory + barrier; split runs) + gn_finalize_kernel against F.group_norm, with
    the adaptive weights of AdaptiveGroupNorm (reference quirk: the static affine applied twice, oracle/restate.py)"""
    g = torch.Generator().manual_seed(S)
    x = torch.randn(N, C, S, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ag, ab = (torch.randn(N, C, generator=g), torch.randn(N, C, generator=g)) if ada else (None, None)
    ref = F.group_norm(x.double(), G, gamma.double(), beta.double(), 1e-5)
    if ada:
        ref = ref * ag.double()[:, :, None] + ab.double()[:, :, None]
    arr = lambda t: None if t is None else np.ascontiguousarray(t.numpy())
    xs = _aligned(x.numel())
    xs[:] = x.numpy().ravel()
    scale, shift = np.empty((N, C), np.float32), np.empty((N, C), np.float32)
    mean, rstd = np.empty((N, G), np.float32), np.empty((N, G), np.float32)
    tlib.emo_groupnorm_workspace_bytes.restype = ctypes.c_int64
    need = tlib.emo_groupnorm_workspace_bytes(N, G)
    ws = np.zeros(need // 8, np.float64)
    rc = tlib.emo_groupnorm_affine_f32(_p(xs), N, C, ctypes.c_int64(S), G, ctypes.c_float(1e-5), _p(arr(gamma)), _p(arr(beta)), _p(arr(ag)),
                                       _p(arr(ab)), ctypes.c_int64(C), _p(scale), _p(shift), _p(mean), _p(rstd), _p(ws), ctypes.c_int64(need), None)
    assert rc == 0
    got = x.double() * torch.from_numpy(scale).double()[:, :, None] + torch.from_numpy(shift).double()[:, :, None]
    assert (got - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    xg = x.double().view(N, G, -1)
    assert np.abs(mean - xg.mean(-1).numpy()).max() < 1e-6
    assert np.abs(rstd - (1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5)).numpy()).max() < 1e-5


@pytest.mark.parametrize("factors", [(2, 2, 2), (1, 2, 2)])
def test_upsampling_sums_are_the_sums_of_its_output(tlib, factors):
    """upsample_trilinear_w2_kernel<true>: the fp64 (sum, sum of squares) slices it leaves per (sample, group) add up to the
    sums of the tensor it wrote, and emo_groupnorm_affine_from_sums_f32 makes of them what the reduction pass makes of the tensor"""
    N, C, G, (D, H, W) = 2, 8, 4, (3, 4, 6)
    fd, fh, fw = factors
    x = np.random.default_rng(3).standard_normal((N * C, D, H, W)).astype(np.float32)
    xin = _aligned(x.size)
    xin[:] = x.ravel()
    Do, Ho, Wo = D * fd, H * fh, W * fw
    out = _aligned(N * C * Do * Ho * Wo)
    partial = np.zeros(N * G * 64 * 2, np.float64)
    split = ctypes.c_int(0)
    rc = tlib.emo_upsample_trilinear_gn_sums_f32(_p(xin), _p(out), N, C, G, D, H, W, fd, fh, fw, _p(partial), ctypes.c_int64(partial.nbytes),
                                                 ctypes.byref(split), None)
    assert rc == 0
    ref = F.interpolate(torch.from_numpy(x)[None], scale_factor=tuple(float(f) for f in factors), mode="trilinear")[0].numpy()
    assert np.abs(out.reshape(ref.shape) - ref).max() < 1e-6
    sums = partial.reshape(N * G, 64, 2)[:, :split.value].sum(1)
    o64 = out.astype(np.float64).reshape(N * G, -1)
    assert np.allclose(sums[:, 0], o64.sum(1), rtol=1e-12, atol=1e-9) and np.allclose(sums[:, 1], (o64 * o64).sum(1), rtol=1e-12)
    gamma, beta = np.linspace(0.5, 1.5, C).astype(np.float32), np.linspace(-1, 1, C).astype(np.float32)
    scale, shift = np.empty((N, C), np.float32), np.empty((N, C), np.float32)
    rc = tlib.emo_groupnorm_affine_from_sums_f32(_p(partial), split.value, N, C, ctypes.c_int64(Do * Ho * Wo), G, ctypes.c_float(1e-5), _p(gamma),
                                                 _p(beta), None, None, ctypes.c_int64(0), _p(scale), _p(shift), None, None, None)
    assert rc == 0
    t = torch.from_numpy(out.reshape(N, C, -1).copy()).double()
    want = F.group_norm(t, G, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5)
    got = t * torch.from_numpy(scale).double()[:, :, None] + torch.from_numpy(shift).double()[:, :, None]
    assert (got - want).abs().max().item() < 1e-5


@pytest.mark.parametrize("NN", [1, 2, 4, 16])
def test_small_gemm_wave_reduction(tlib, NN):
    """small_gemm_kernel: one wave per output row, lanes stride over k, butterfly reduction by __shfl_xor"""
    rng = np.random.default_rng(NN)
    M, K, B = 6, 150, 2
    A, Bm = rng.standard_normal((M, K)).astype(np.float32), rng.standard_normal((B, K, NN)).astype(np.float32)
    Cm = np.empty((B, M, NN), np.float32)
    rc = tlib.emo_small_gemm_f32(_p(A), _p(Bm), _p(Cm), M, K, NN, B, ctypes.c_int64(K * NN), ctypes.c_int64(M * NN), None)
    assert rc == 0
    ref = np.einsum("mk,bkn->bmn", A.astype(np.float64), Bm.astype(np.float64))
    assert np.abs(Cm - ref).max() < 1e-4
