"""ctypes binding of libemoportraits_hip.so -- the C ABI declared in include/emo_hip.h.

The product path has NO fallback: if the library is missing or a kernel call fails, a RuntimeError is raised.
PyTorch is only plumbing here (device memory, streams): tensors are passed as raw device pointers.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EMO_HIP_LIB selects an alternative build of the same library (A/B measurements of kernel variants only)
LIB_PATH = os.environ.get("EMO_HIP_LIB") or os.path.join(_HERE, "lib", "libemoportraits_hip.so")

PAD_MODES = {"zeros": 0, "border": 1, "reflection": 2}
LAYOUT_NCDHW, LAYOUT_NDHWC, LAYOUT_P4 = 0, 1, 3
ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3}

_c_int, _c_i64, _c_void, _c_float = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float

# name -> argtypes; restype is int unless listed in _RESTYPES.  Mirrors include/emo_hip.h one to one
# (tests/test_abi.py checks header <-> this table <-> exported symbols).
SIGNATURES = {
    "emo_abi_version": [],
    "emo_build_info": [],
    "emo_device_cu_count": [],
    "emo_mfma_stream_f16": [_c_void, _c_int, _c_int, ctypes.POINTER(_c_i64), _c_void],
    "emo_grid_sample3d_f32": [_c_void] * 7 + [_c_int] * 8 + [_c_i64] + [_c_int] * 5 + [_c_void],
    "emo_affine_grid3d_f32": [_c_void] * 5 + [_c_int] * 4 + [_c_void],
    "emo_volume_repack_f32": [_c_void, _c_void, _c_int, _c_int, _c_int, _c_int, _c_void],
    "emo_groupnorm_workspace_bytes": [_c_int, _c_int],
    "emo_groupnorm_affine_f32": [_c_void, _c_int, _c_int, _c_i64, _c_int, _c_float] + [_c_void] * 4 + [_c_i64]
                                + [_c_void] * 5 + [_c_i64, _c_void],
    "emo_groupnorm_affine_from_tiles_f32": [_c_void, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_float] + [_c_void] * 4
                                           + [_c_i64] + [_c_void] * 5,
    "emo_groupnorm_affine_from_sums_f32": [_c_void, _c_int, _c_int, _c_int, _c_i64, _c_int, _c_float] + [_c_void] * 4
                                          + [_c_i64] + [_c_void] * 5,
    "emo_conv_pack_info": [_c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)],
    "emo_conv_tile_positions": [_c_int],
    "emo_conv_igemm_f32": [_c_void] * 7 + [_c_int] * 15 + [_c_void, _c_void, _c_void],
    "emo_conv_igemm_ksplit": [_c_int] * 11,
    "emo_conv_pack_info_f16": [_c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)],
    "emo_conv_igemm_f16acc32": [_c_void] * 7 + [_c_int] * 15 + [_c_void, _c_void, _c_void],
    "emo_conv_igemm_f16w8": [_c_void] * 7 + [_c_int] * 15 + [_c_void, _c_void, _c_void, ctypes.c_float],
    "emo_conv_igemm_f16w8_rest": [_c_void] * 8 + [_c_int] * 15 + [_c_void, _c_void, _c_void, ctypes.c_float],
    "emo_conv_pack_info_bf16x3": [_c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)],
    "emo_conv_igemm_bf16x3": [_c_void] * 7 + [_c_int] * 15 + [_c_void, _c_void, _c_void, _c_void],
    "emo_conv_igemm_f16x2": [_c_void] * 7 + [_c_int] * 15 + [_c_void, _c_void, _c_void, ctypes.c_float, ctypes.c_float, _c_void],
    "emo_conv_igemm_f32_guarded": [_c_void] * 7 + [_c_int] * 15 + [_c_void, _c_void, _c_void, _c_void],
    "emo_conv_head_f32": [_c_void] * 6 + [_c_int, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_void],
    "emo_upsample_trilinear_f32": [_c_void, _c_void, _c_i64] + [_c_int] * 6 + [_c_void],
    "emo_upsample_trilinear_gn_sums_f32": [_c_void, _c_void] + [_c_int] * 9 + [_c_void, _c_i64, ctypes.POINTER(_c_int), _c_void],
    "emo_avgpool_f32": [_c_void, _c_void, _c_i64] + [_c_int] * 6 + [_c_void],
    "emo_add_f32": [_c_void, _c_void, _c_void, _c_i64, _c_i64, _c_float, _c_void],
    "emo_resize2d_f32": [_c_void, _c_i64, _c_i64, _c_void, _c_i64] + [_c_int] * 6 + [_c_void],
    "emo_resize2d_windows_f32": [_c_void, _c_i64, _c_i64, _c_void, _c_void] + [_c_int] * 6 + [_c_void],
    "emo_conv2d_generic_f32": [_c_void] * 6 + [_c_int] * 11 + [_c_void, _c_void],
    "emo_conv2d_generic_splits": [_c_int] * 9,
    "emo_maxpool2d_f32": [_c_void] * 4 + [_c_i64] + [_c_int] * 6 + [_c_void],
    "emo_affine_add_relu_f32": [_c_void] * 7 + [_c_i64, _c_i64, _c_int, _c_void],
    "emo_grid_sample2d_f32": [_c_void] * 6 + [_c_int] * 6 + [_c_void],
    "emo_mat4_inverse_f32": [_c_void, _c_void, _c_int, _c_void],
    "emo_mul_mask_f32": [_c_void, _c_void, _c_void, _c_int, _c_int, _c_i64, _c_void],
    "emo_stage2_compose_f32": [_c_void] * 5 + [_c_int, _c_int, _c_i64, _c_void],
    "emo_small_gemm_f32": [_c_void] * 3 + [_c_int] * 4 + [_c_i64, _c_i64, _c_void],
    "emo_projector_finalize_f32": [_c_void] * 7 + [_c_int] * 3 + [_c_void],
    "emo_pose_theta_f32": [_c_void, _c_int, _c_void, _c_void, _c_void, _c_int, _c_void],
    "emo_pack_rgb8": [_c_void, _c_void, _c_int, _c_int, _c_int, _c_void],
    "emo_unpack_rgb8": [_c_void, _c_void, _c_int, _c_int, _c_int, _c_void],
}
_RESTYPES = {"emo_build_info": ctypes.c_char_p, "emo_groupnorm_workspace_bytes": _c_i64}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """dlopen the library (idempotent).  Raises HipLibraryError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m emoportraits_amd.build` (there is no CPU/eager fallback)")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _c_int)
    from . import _abi_version
    v = lib.emo_abi_version()
    if v != _abi_version.EMO_ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {v}, python {_abi_version.EMO_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


_ERR = {-1: "EMO_ERR_BAD_ARG", -2: "EMO_ERR_UNSUPPORTED", -3: "EMO_ERR_ALIGN"}


def check(rc, what):
    if rc != 0:
        name = _ERR.get(rc, f"hipError {rc}" if rc > 0 else f"error {rc}")
        raise RuntimeError(f"{what} failed: {name}")


def ptr(t):
    """device pointer of a tensor (or None)"""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda_f32(*tensors):
    import torch
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("emoportraits_amd ops run on the GPU only (got a %s tensor); there is no CPU path"
                               % t.device.type)
        if t.dtype != torch.float32:
            raise RuntimeError("expected float32, got %s" % t.dtype)
        if not t.is_contiguous():
            raise RuntimeError("expected a contiguous tensor")
