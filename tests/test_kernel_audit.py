"""CPU test (hipcc cross-compiles): the conv kernels hide their global loads in inline asm (conv_igemm.h, conv_igemm_f16.h), so
a register the compiler spills while such a load is in flight would be silent corruption.  tools/kernel_resources.py --audit
compiles every instantiation to ISA and checks that no scratch access lies between the first pinned load and the last MFMA."""
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
