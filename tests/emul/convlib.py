"""TEST INFRASTRUCTURE ONLY -- the convolution library of the hot path (csrc/conv_api.hip, every conv_inst_*.hip, the kernel
headers conv_igemm*.h) compiled as host C++ from COPIES of the product's sources, for tests/test_conv_igemm_emul.py and
tests/test_conv_split_emul.py.

ROCm's clang++ compiles the copies against the stand-in <hip/hip_runtime.h> of tests/emul/hipshim in its threaded mode (the
threads of a block are OS threads; __syncthreads a barrier; the MFMA instructions, the DPP row operations and the wave shuffles
exchanges between the 64 lanes of a wave with the hardware's operand / result layouts; LDS-DMA a copy into the block's LDS
buffer).  What only the GPU toolchain understands is rewritten in the copies, statement by statement (REWRITES below): pinned
loads become plain loads, waits and scheduling fences nothing, `s_waitcnt ...; s_barrier` a barrier of the block, accumulation
register reads plain reads, the dynamic shared-memory declaration a pointer to the stand-in's per-block buffer (LDS byte
addresses are offsets into it), the occupancy attribute nothing.  The host run therefore checks the kernels' index arithmetic,
operand conversion, LDS images, fragment addressing, item scheduling, epilogues and statistics -- not their pipelining (a load
is complete when it is issued here): that is the ISA audit's and the GPU tests' job.  A statement the rewrites do not know fails
the build.  The product sources are not touched and the product never loads this library.
"""
import concurrent.futures
import ctypes
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "emoportraits_amd", "csrc")
SHIM = os.path.join(HERE, "hipshim")
GEN = os.path.join(HERE, "_build", "gen_convlib")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
PRODUCT_LIB = os.path.join(ROOT, "emoportraits_amd", "lib", "libemoportraits_hip.so")

HEADERS = ["conv_igemm.h", "conv_igemm_f16.h", "conv_igemm_bf16x3.h", "conv_split_pair_common.h", "conv_igemm_f16x2_ct2.h", "conv_igemm_f16x2_w8.h",
           "conv_igemm_f16x2_p1.h",
           "conv_dispatch.h"]
UNITS = ["conv_api.hip"] + sorted(f for f in os.listdir(CSRC) if f.startswith("conv_inst_") and f.endswith(".hip"))

# (regex, replacement) applied to every copied file; the counts are checked per file below
GENERIC = [
    # register "declarations" and scheduling fences: empty asm statements with one operand
    (r'asm volatile\(""\s*:\s*"[=+][vs]"\(([^;]*?)\)\);', r'(void)0;'),
    # waits
    (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\(n_\) : "memory"\)', r'((void)0)'),
    (r'asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\);', r';'),
    # wait + barrier of the block
    (r'asm volatile\("s_waitcnt vmcnt\(%0\) lgkmcnt\(0\)\\n\\ts_barrier" ::"n"\(n_\) : "memory"\)', r'__syncthreads()'),
    # accumulation registers are ordinary variables here
    (r'asm\("v_accvgpr_read_b32 %0, %1" : "=v"\(v\) : "a"\(acc_element\)\);', r'v = acc_element;'),
    # pinned global loads (conv_igemm.h)
    (r'asm volatile\(EMO_SGPR_HAZARD_NOP "global_load_dword %0, %1, %2" : "=v"\(v\) : "v"\(voff\), "s"\(sbase\) : "memory"\);',
     r'memcpy(&v, reinterpret_cast<const char*>(sbase) + voff, 4);'),
    (r'asm volatile\(EMO_SGPR_HAZARD_NOP "global_load_dwordx4 %0, %1, %2" : "=v"\(v\) : "v"\(voff\), "s"\(sbase\) : "memory"\);',
     r'memcpy(&v, reinterpret_cast<const char*>(sbase) + voff, 16);'),
    # LDS-DMA (conv_igemm_f16.h): 16 bytes per lane to the LDS byte address lds_dst + lane * 16
    (r'asm volatile\("s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %2\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, off\\n\\ts_mov_b32 m0, %0"\s*:\s*"=&s"\(keep\) : "v"\(gsrc\), "s"\(lds_dst\) : "memory"\);',
     r'(void)keep; hipshim_lds_dma16(gsrc, lds_dst);'),
    (r'asm volatile\("s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %3\\n\\ts_nop 2\\n\\tglobal_load_lds_dwordx4 %1, %2\\n\\ts_mov_b32 m0, %0"\s*:\s*"=&s"\(keep\) : "v"\(voff\), "s"\(sbase\), "s"\(lds_dst\) : "memory"\);',
     r'(void)keep; hipshim_lds_dma16(reinterpret_cast<const char*>(sbase) + voff, lds_dst);'),
    # raw buffer loads (conv_igemm_f16.h): resource base + soff + voff, no range limit
    (r'asm volatile\(EMO_SGPR_HAZARD_NOP "buffer_load_dword %0, %1, %2, %3 offen" : "=v"\(v\) : "v"\(voff\), "s"\(rsrc\), "s"\(soff\) : "memory"\);',
     r'memcpy(&v, hipshim_buffer_base(rsrc[0], rsrc[1]) + soff + voff, 4);'),
    (r'asm volatile\(EMO_SGPR_HAZARD_NOP "buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"\(v\) : "v"\(voff\), "s"\(rsrc\), "s"\(soff\) : "memory"\);',
     r'memcpy(&v, hipshim_buffer_base(rsrc[0], rsrc[1]) + soff + voff, 16);'),
    (r'asm volatile\(EMO_SGPR_HAZARD_NOP "buffer_load_dwordx4 %0, %2, %3, %4 offen\\n\\tbuffer_load_dwordx4 %1, %2, %3, %5 offen"\s*:\s*"=&v"\(v0\), "=&v"\(v1\) : "v"\(voff\), "s"\(rsrc\), "s"\(soff0\), "s"\(soff1\) : "memory"\);',
     r'memcpy(&v0, hipshim_buffer_base(rsrc[0], rsrc[1]) + soff0 + voff, 16); memcpy(&v1, hipshim_buffer_base(rsrc[0], rsrc[1]) + soff1 + voff, 16);'),
    # LDS: the dynamic buffer of the stand-in; LDS byte addresses are offsets into it
    (r'extern\s+__shared__\s+__attribute__\(\(aligned\(16\)\)\)\s+float\s+smem\[\];', r'float* const smem = reinterpret_cast<float*>(hipshim_dynamic_smem());'),
    (r'const unsigned smem_lds = \(unsigned\)\(size_t\)\(__attribute__\(\(address_space\(3\)\)\) char\*\)reinterpret_cast<char\*>\(smem\);',
     r'const unsigned smem_lds = 0u;'),
    (r'__attribute__\(\(amdgpu_waves_per_eu\(.*?\)\)\)', r''),
    # the epilogues of the split kernels transpose through a wave-private LDS region: written and read by the lanes of one wave
    # without a barrier in the source (lock step on the GPU) -- a barrier of the wave around the reads
    (r'(#pragma unroll\s*\n\s*for \(int it = 0; it < NIT; \+\+it\) v\[it\] = \*reinterpret_cast<const floatx4\*>\(scratch \+ \(4 \* it \+ g\) \* ROWF \+ 4 \* t\);)',
     r'hipshim_wave_sync();\n\1\nhipshim_wave_sync();'),
    # (the same in conv_igemm_f16x2_w8.h: 32-position passes, rows of 8 channels)
    (r'(#pragma unroll\s*\n\s*for \(int it = 0; it < 4; \+\+it\) v\[J\]\[it\] = \*reinterpret_cast<const floatx4\*>\(scratch \+ \(8 \* it \+ g8\) \* ROWF \+ 4 \* t8\);)',
     r'hipshim_wave_sync();\n\1\nhipshim_wave_sync();'),
]


def _strip_comments(text):
    return re.sub(r"//[^\n]*|/\*.*?\*/", "", text, flags=re.S)


def _inactive_timing_blocks(text):
    """the `#if EMO_S_TIMING == 2` variants of the barrier macros (measurement builds) are never compiled here"""
    return re.sub(r"#if EMO_S_TIMING == 2.*?#else", "#if 0\n#else", text, flags=re.S)


def available():
    return os.path.exists(CLANG) and os.path.exists(PRODUCT_LIB)


def build():
    # eight "compute units" (the stand-in's default): the persistent split kernels hand out their work items in eight contiguous
    # ranges, one per XCD (block b walks the range of XCD b % 8) -- their grids are min(items, CUs) blocks and assume a CU count that
    # is a multiple of 8, as MI355X has in every partition mode.  More than 8 items: several chained items per block.
    os.environ.setdefault("HIPSHIM_CUS", "8")
    out = os.path.join(GEN, "libconv_emul.so")
    deps = [os.path.join(CSRC, f) for f in HEADERS + UNITS + ["common.h"]] + [os.path.join(SHIM, "hip", "hip_runtime.h"), __file__]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return ctypes.CDLL(out)
    os.makedirs(GEN, exist_ok=True)
    for f in HEADERS + UNITS:
        text = _inactive_timing_blocks(open(os.path.join(CSRC, f)).read())
        for pat, rep in GENERIC:
            text = re.sub(pat, rep, text, flags=re.S)
        left = [m.group(0)[:80] for m in re.finditer(r"\basm\b[^\n]*", _strip_comments(text))]
        assert not left, f"{f}: inline-asm statements the emulation's rewrites do not know: {left}"
        open(os.path.join(GEN, f), "w").write(text)
    # (-ftrivial-auto-var-init=zero: a register the kernel "declares" with an empty asm statement and reads before it writes it is
    # garbage on the GPU and must not be undefined behaviour here)
    flags = [CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-ftrivial-auto-var-init=zero", "-DHIPSHIM_THREADS", "-pthread", "-I" + SHIM, "-I" + CSRC, "-w", "-fPIC"]

    def cc(f):
        obj = os.path.join(GEN, f + ".o")
        subprocess.run(flags + ["-c", "-x", "c++", os.path.join(GEN, f), "-o", obj], check=True, cwd=GEN)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(cc, UNITS))
    subprocess.run([CLANG, "-shared", "-pthread", "-o", out] + objs, check=True)
    return ctypes.CDLL(out)
