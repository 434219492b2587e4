"""CPU test (hipcc cross-compiles): the conv kernels hide their global loads in inline asm (conv_igemm.h, conv_igemm_f16.h), so
a register the compiler spills while such a load is in flight would be silent corruption.  tools/kernel_resources.py --audit
compiles every instantiation to ISA and checks that no scratch access lies between the first pinned load and the last MFMA,
and that no instruction touches the destination of a pinned load which the listing's hand-counted vmcnt waits have not
covered yet (the fp16-operand kernel keeps loads in flight across the stage boundary); and that no vector-memory instruction
of an asm statement reads a scalar register within five wait states of a vector-ALU write of it (the compiler reloads spilled
scalars with v_readlane wherever it likes, and does not pad asm statements: round 4's A/B builds computed garbage that way)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("EMO_SKIP_AUDIT") == "1", reason="needs hipcc")
def test_no_scratch_access_while_pinned_loads_are_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--audit"], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 violations" in r.stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("EMO_SKIP_AUDIT") == "1", reason="needs hipcc")
def test_the_measurement_build_of_the_split_kernel_passes_the_same_audit():
    """an A/B / measurement build is a different register allocation of the same source (python -m emoportraits_amd.build
    --variant x NAME=V  <->  --audit -DNAME=V): the unchained fp16 split is the one that exposed the scalar-register hazard"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--audit", "-DEMO_S_CHAIN=0"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 violations" in r.stdout


LISTING = """
\t#ASMSTART
\tglobal_load_dword v10, v1, s[2:3]
\t#ASMEND
\t#ASMSTART
\tglobal_load_dword v11, v1, s[4:5]
\t#ASMEND
.LBB0_1:
\t#ASMSTART
\ts_waitcnt vmcnt({first})
\t#ASMEND
\tv_fma_f32 v20, v10, v2, v3
\t#ASMSTART
\tglobal_load_dword v10, v1, s[2:3]
\t#ASMEND
\t#ASMSTART
\ts_waitcnt vmcnt({second})
\t#ASMEND
\tv_fma_f32 v21, v11, v2, v3
\t#ASMSTART
\tglobal_load_dword v11, v1, s[4:5]
\t#ASMEND
\tv_mfma_f32_32x32x16_f16 v[30:45], v[4:7], v[8:9], v[30:45]
\ts_cbranch_scc0 .LBB0_1
"""


def test_inflight_checker_on_a_synthetic_rolling_loop():
    """two registers reloaded right after their use: each wait may leave exactly one (the other register's) load outstanding"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K
    ok = LISTING.format(first=1, second=1).split("\n")
    assert K.inflight_reads(ok) == []
    late = K.inflight_reads(LISTING.format(first=2, second=1).split("\n"))
    assert late and late[0][2] == [10]
    second_pass_only = K.inflight_reads(LISTING.format(first=1, second=2).split("\n"))
    assert second_pass_only and second_pass_only[0][2] == [11]


SKIPPED_PIECE = """
\t#ASMSTART
\tglobal_load_dword v10, v1, s[2:3]
\t#ASMEND
.LBB0_1:
\t#ASMSTART
\tglobal_load_lds_dwordx4 v[4:5], off
\t#ASMEND
\ts_and_saveexec_b64 s[8:9], vcc
\ts_cbranch_execz .LBB0_3
\t#ASMSTART
\tglobal_load_lds_dwordx4 v[4:5], off
\t#ASMEND
.LBB0_3:
\ts_or_b64 exec, exec, s[8:9]
\t#ASMSTART
\ts_waitcnt vmcnt({n})
\t#ASMEND
\tv_fma_f32 v20, v10, v2, v3
\t#ASMSTART
\tglobal_load_dword v10, v1, s[2:3]
\t#ASMEND
\tv_mfma_f32_32x32x16_f16 v[30:45], v[4:7], v[8:9], v[30:45]
\ts_cbranch_scc0 .LBB0_1
"""


def test_inflight_checker_does_not_count_a_piece_some_waves_skip():
    """an LDS-DMA piece under an exec-masked branch is issued by some waves only: a wait that needs it to have been issued in
    order to cover the older load is flagged (vmcnt(2)), the count that holds for every wave (vmcnt(1)) is accepted"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K
    assert K.inflight_reads(SKIPPED_PIECE.format(n=1).split("\n")) == []
    bad = K.inflight_reads(SKIPPED_PIECE.format(n=2).split("\n"))
    assert bad and bad[0][2] == [10]
