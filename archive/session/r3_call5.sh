#!/bin/bash
# round 3, call 5: full GPU parity suite + the bench line with extras
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 1700 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r3c5_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/r3c5_pytest_full.log > gpurun_out/r3c5_pytest.log
tail -5 gpurun_out/r3c5_pytest.log
( time timeout 900 python bench.py > gpurun_out/r3c5_bench.json 2> gpurun_out/r3c5_bench.err ) 2> gpurun_out/r3c5_bench.time
tail -3 gpurun_out/r3c5_bench.err; cat gpurun_out/r3c5_bench.time; cut -c1-600 gpurun_out/r3c5_bench.json
