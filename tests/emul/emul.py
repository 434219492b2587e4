"""TEST INFRASTRUCTURE ONLY -- builds tests/emul/gs3d_tile_emul.cpp (the tile sampler's phases as host C++) and wraps it."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "gs3d_tile_emul.cpp")
_BUILD = os.path.join(_HERE, "_build")
_LIB = os.path.join(_BUILD, "libgs3d_emul.so")
_lib = None
MODES = {"grid": 0, "theta": 1, "delta": 2}
PAD = {"zeros": 0, "border": 1, "reflection": 2}


def build():
    os.makedirs(_BUILD, exist_ok=True)
    deps = [_SRC] + [os.path.join(_HERE, "..", "..", "emoportraits_amd", "csrc", f) for f in ("gs3d_tile.h", "gs3d_coord.h")]
    if os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(d) for d in deps):
        return _LIB
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", _LIB, _SRC], check=True)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def to_p4(vol):
    n, c, d, h, w = vol.shape
    return np.ascontiguousarray(vol.reshape(n, c // 4, 4, d, h, w).transpose(0, 1, 3, 4, 5, 2))


def from_p4(v):
    n, q, d, h, w, _ = v.shape
    return np.ascontiguousarray(v.transpose(0, 1, 5, 2, 3, 4).reshape(n, q * 4, d, h, w))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def lattice(n):
    import torch
    return torch.linspace(-1, 1, n).numpy().astype(np.float32)


def run(vol, *, grid=None, theta=None, delta=None, pad="zeros", in_p4=True, out_p4=False, threads=256, tile=(2, 3, 3),
        upb=3, cap_slots=2552, out_size=None):
    """vol [Nv,C,D,H,W] NCDHW (packed here when in_p4); tile = (log2 tx, log2 ty, log2 tz); returns (out NCDHW, stats)"""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    Nv, C, D, H, W = vol.shape
    if grid is not None:
        mode, g = "grid", np.ascontiguousarray(grid, np.float32)
        N, Do, Ho, Wo, _ = g.shape
    elif delta is not None:
        mode, g = "delta", np.ascontiguousarray(delta, np.float32)
        N, _, Do, Ho, Wo = g.shape
    else:
        mode, g = "theta", None
        theta = np.ascontiguousarray(theta, np.float32)
        N = theta.shape[0]
        Do, Ho, Wo = out_size or (D, H, W)
    lx, ly, lz = lattice(Wo), lattice(Ho), lattice(Do)
    src = to_p4(vol) if in_p4 else vol
    stride = 0 if (Nv == 1 and N > 1) else C * D * H * W
    out = np.full((N, C // 4, Do, Ho, Wo, 4) if out_p4 else (N, C, Do, Ho, Wo), np.nan, np.float32)
    txs, tys, tzs = tile
    vpt = (1 << (txs + tys + tzs)) // threads
    st = (ctypes.c_long * 6)()
    rc = lib().emul_gs3d_tile(_p(src), _p(g), _p(theta), _p(lx), _p(ly), _p(lz), _p(out), N, C, D, H, W, Do, Ho, Wo,
                              ctypes.c_long(stride), PAD[pad], MODES[mode], int(in_p4), int(out_p4), threads, vpt, txs, tys, tzs,
                              upb, cap_slots, st)
    assert rc == 0, rc
    stats = dict(zip(("blocks", "staged_passes", "direct_passes", "union_blocks", "slots_filled", "stages"), list(st)))
    return (from_p4(out) if out_p4 else out), stats
