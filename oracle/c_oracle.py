"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of the plain-C sampler oracle (oracle/grid_sample3d.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None
PAD = {"zeros": 0, "border": 1, "reflection": 2}


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def grid_sample3d(vol, grid, padding_mode="zeros"):
    """vol [Nv,C,D,H,W] float32 (Nv = N or 1 = shared), grid [N,Do,Ho,Wo,3] -> [N,C,Do,Ho,Wo]"""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    grid = np.ascontiguousarray(grid, dtype=np.float32)
    Nv, C, D, H, W = vol.shape
    N, Do, Ho, Wo, _ = grid.shape
    out = np.empty((N, C, Do, Ho, Wo), dtype=np.float32)
    stride = 0 if (Nv == 1 and N > 1) else C * D * H * W
    rc = lib().oracle_grid_sample3d_f32(_p(vol), _p(grid), _p(out), N, C, D, H, W, Do, Ho, Wo,
                                        ctypes.c_int64(stride), PAD[padding_mode])
    assert rc == 0
    return out


def affine_grid3d(theta, lin_x, lin_y, lin_z):
    """theta [N,3,4] -> grid [N,Do,Ho,Wo,3] = identity lattice @ theta^T as the reference's bmm computes it"""
    theta = np.ascontiguousarray(theta, dtype=np.float32)
    lx, ly, lz = (np.ascontiguousarray(a, dtype=np.float32) for a in (lin_x, lin_y, lin_z))
    N = theta.shape[0]
    grid = np.empty((N, lz.size, ly.size, lx.size, 3), dtype=np.float32)
    rc = lib().oracle_affine_grid3d_f32(_p(theta), _p(lx), _p(ly), _p(lz), _p(grid), N, lz.size, ly.size, lx.size)
    assert rc == 0
    return grid
