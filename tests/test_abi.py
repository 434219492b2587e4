"""CPU test: libemoportraits_hip.so loads and exports every symbol include/emo_hip.h declares (no kernel is launched)."""
import ctypes
import os
import re

from emoportraits_amd import hip, _abi_version

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "emo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emo_[a-z0-9_]+)\s*\(", src)))


def test_header_python_and_library_agree():
    names = _declared()
    assert names, "no declarations parsed"
    assert sorted(hip.SIGNATURES) == names, "emoportraits_amd/hip.py SIGNATURES out of sync with include/emo_hip.h"
    lib = ctypes.CDLL(hip.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in emo_hip.h but not exported"


def test_abi_version_and_build_info():
    lib = hip.load()
    assert lib.emo_abi_version() == _abi_version.EMO_ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "emo_hip.h")).read()
    assert int(re.search(r"#define EMO_ABI_VERSION (\d+)", hdr).group(1)) == _abi_version.EMO_ABI_VERSION
    assert b"gfx950" in lib.emo_build_info()


def test_argument_errors_are_reported_without_touching_the_gpu():
    lib = hip.load()
    # null pointers -> EMO_ERR_BAD_ARG before any launch
    rc = lib.emo_grid_sample3d_f32(None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, None)
    assert rc == -1
    rc = lib.emo_conv_igemm_f32(None, None, None, None, None, None, None, 1, 4, 4, 1, 8, 8, 1, 3, 3, 0, 0, 0, 0, 0, 1, None, None, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from emoportraits_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.grid_sample3d(torch.zeros(1, 4, 2, 2, 2), torch.zeros(1, 1, 1, 1, 3))
