"""TEST INFRASTRUCTURE ONLY -- builders of the host-compiled kernel libraries (the product's sources against the stand-in runtime
of tests/emul/hipshim) and a facade that presents them to emoportraits_amd as ITS library, so that the package's host code can be
run end to end on CPU tensors in a test (tests/test_hot_path_emul.py).  The product never imports this module.
"""
import ctypes
import os
import re
import subprocess

import torch

import convlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "emoportraits_amd", "csrc")
SHIM = os.path.join(HERE, "hipshim")
BUILD = os.path.join(HERE, "_build")
CLANG = convlib.CLANG
STREAM_SOURCES = [os.path.join(CSRC, f) for f in ("resample.hip", "conv_head.hip", "embed_ops.hip", "smallops.hip", "groupnorm.hip")]
_COMMON_DEPS = [os.path.join(SHIM, "hip", "hip_runtime.h"), os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "emo_hip.h")]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps)


def stream(threaded):
    """resample / conv_head / embed_ops / smallops / groupnorm: every source a translation unit of its own (`g++ -x c++ file.hip`),
    the flags that matter for the arithmetic from emoportraits_amd/build.py (-ffp-contract=off).  threaded=False: a launch is a loop
    over blocks and threads (barrier-free kernels only); True: OS threads, real barriers and wave exchanges"""
    out = os.path.join(BUILD, "libstream_emul_threads.so" if threaded else "libstream_emul.so")
    if _stale(out, STREAM_SOURCES + _COMMON_DEPS):
        os.makedirs(BUILD, exist_ok=True)
        extra = ["-DHIPSHIM_THREADS", "-pthread"] if threaded else []
        subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + SHIM, "-shared", "-fPIC"] + extra +
                       ["-o", out, "-x", "c++"] + STREAM_SOURCES, check=True)
    return ctypes.CDLL(out)


def sampler():
    """csrc/grid_sample3d.hip (ROCm's clang++: vector extensions), threaded; the dynamic shared-memory declaration rewritten to the
    stand-in's per-block buffer, the dispatchers of the LDS-staged tile kernels (another translation unit) stubs that refuse"""
    out = os.path.join(BUILD, "libsampler_emul_threads.so")
    src = os.path.join(CSRC, "grid_sample3d.hip")
    if _stale(out, [src, os.path.join(CSRC, "gs3d_coord.h"), __file__] + _COMMON_DEPS):
        gen = os.path.join(BUILD, "gen")
        os.makedirs(gen, exist_ok=True)
        text, n = re.subn(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];",
                          r"\1* const \2 = reinterpret_cast<\1*>(hipshim_dynamic_smem());", open(src).read())
        assert n == 1, "one dynamic shared-memory declaration expected in grid_sample3d.hip"
        open(os.path.join(gen, "grid_sample3d.hip"), "w").write(text)
        open(os.path.join(gen, "stubs.cpp"), "w").write(
            "#include <stdint.h>\n"
            "int emo_gs3d_tile_dispatch(const float*, const float*, const float*, const float*, const float*, const float*, float*, int, int, "
            "int, int, int, int, int, int, int64_t, int, int, int, int, int, void*) { return -2; }\n"
            "int emo_repack_p4_dispatch(const float*, float*, int, int, int, int, void*) { return -2; }\n")
        subprocess.run([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-DHIPSHIM_THREADS", "-pthread", "-I" + SHIM, "-I" + CSRC, "-w",
                        "-shared", "-fPIC", "-o", out, "-x", "c++", os.path.join(gen, "grid_sample3d.hip"), os.path.join(gen, "stubs.cpp")],
                       check=True)
    return ctypes.CDLL(out)


def available():
    return convlib.available()


class EmulatedLibrary:
    """libemoportraits_hip.so as the package sees it, every kernel entry point served by a host-compiled library; argument and
    result types from emoportraits_amd.hip.SIGNATURES (the table tests/test_abi.py pins to include/emo_hip.h)"""

    def __init__(self):
        from emoportraits_amd import hip
        self._libs = [convlib.build(), sampler(), stream(True)]
        self._sig, self._res = hip.SIGNATURES, hip._RESTYPES
        self._cache = {}
        self.calls = {}

    def __getattr__(self, name):
        if name.startswith("_") or name == "calls":
            raise AttributeError(name)
        if name not in self._cache:
            for lib in self._libs:
                try:
                    fn = getattr(lib, name)
                except AttributeError:
                    continue
                fn.argtypes = self._sig[name]
                fn.restype = self._res.get(name, ctypes.c_int)
                self._cache[name] = fn
                break
            else:
                raise AttributeError(f"{name}: not in any host-compiled library")
        fn = self._cache[name]

        def counted(*args):
            self.calls[name] = self.calls.get(name, 0) + 1
            return fn(*args)
        return counted


def install(monkeypatch_setattr=setattr):
    """route emoportraits_amd through an EmulatedLibrary on CPU tensors; returns it"""
    from emoportraits_amd import hip, ops
    lib = EmulatedLibrary()
    monkeypatch_setattr(hip, "load", lambda: lib)
    monkeypatch_setattr(hip, "require_cuda_f32", lambda *a, **k: None)
    monkeypatch_setattr(hip, "current_stream", lambda: None)
    monkeypatch_setattr(ops, "_lattice", lambda n, idx: torch.linspace(-1, 1, n))
    monkeypatch_setattr(ops, "_ada_views", lambda ag, ab, N, C: 0 if ag is None else ag.stride(0))
    monkeypatch_setattr(torch.cuda, "current_device", lambda: 0)
    return lib
