#!/bin/bash
# re-run of the GPU parity suite alone (after a test-only fix): refreshes gpurun_out/<tag>_pytest*.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/${TAG}_pytest_full.log > gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
