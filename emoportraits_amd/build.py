"""Builds libemoportraits_hip.so (hand-written gfx950 HIP kernels + the C ABI of include/emo_hip.h) in-tree.

    python -m emoportraits_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  Each .hip file is compiled to an object in parallel, then linked.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libemoportraits_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: the sampler's index arithmetic must not be contracted into FMAs (bit-parity with ATen);
# kernels that want FMAs ask for them explicitly (__fmaf_rn / MFMA).
# -pragma-unroll-threshold: the conv kernels index their accumulator / staging register arrays with loop counters of
# `#pragma unroll` loops; past LLVM's default size budget (16 K) a loop silently stays rolled, the indices become dynamic and
# the arrays fall into scratch memory (seen on the 64 x 512 tile: 576 B of scratch inside the K loop).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-mllvm", "-pragma-unroll-threshold=100000"]


# per-file extra flags.  gs3d_tile_pad_*: hipcc's SLP vectorizer turns the sampler's per-channel multiply / add into v_pk_*_f32
# and, to feed them, hoists a {w, w} splat of every corner weight out of the loops -- 16 extra VGPRs per owned voxel, i.e.
# spills at the 4-waves-per-SIMD budget (measured: 104 spilled VGPRs with, 0 without).
EXTRA_FLAGS = {"gs3d_tile_pad_zeros.hip": ["-fno-slp-vectorize"], "gs3d_tile_pad_border.hip": ["-fno-slp-vectorize"],
               "gs3d_tile_pad_reflection.hip": ["-fno-slp-vectorize"],
               # conv_igemm_bf16x3.h: SLP packs the split's subtractions into v_pk_add_f32, which beside MFMAs costs more than
               # the two scalar ops it replaces (MI355X_MICROARCH.md, filler prices of a one-wave-per-SIMD stream)
               "conv_inst_bf16x3_3x3.hip": ["-fno-slp-vectorize"], "conv_inst_f16x2_3x3.hip": ["-fno-slp-vectorize"],
               "conv_inst_f16x2_ct2.hip": ["-fno-slp-vectorize"], "conv_inst_f16x2_p1.hip": ["-fno-slp-vectorize"],
               "conv_inst_f16x2_w8.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    for p in sorted(paths):
        h.update(" ".join(EXTRA_FLAGS.get(os.path.basename(p), [])).encode())
    return h.hexdigest()


def _compile(src):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "emo_hip.h"))
    dig = _digest([src] + hdrs)
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    stamp = obj + ".sha"
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    extra = EXTRA_FLAGS.get(os.path.basename(src), [])
    cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build_variant(name, defines):
    """A/B build of the whole library with extra -D flags into lib/libemoportraits_hip_<name>.so (measurement only;
    selected at run time with EMO_HIP_LIB=<path>)."""
    global FLAGS, OBJDIR, LIB
    saved = (FLAGS, OBJDIR, LIB)
    try:
        FLAGS = FLAGS + ["-D" + d for d in defines]
        OBJDIR = os.path.join(LIBDIR, "obj_" + name)
        LIB = os.path.join(LIBDIR, f"libemoportraits_hip_{name}.so")
        return build(force=False, verbose=True)
    finally:
        FLAGS, OBJDIR, LIB = saved


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
