#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_stage2_gpu.py tests/test_nets_gpu.py tests/test_infer_gpu.py -m gpu -q --timeout=900 -s 2>&1 | grep -v amdgpu.ids > gpurun_out/r1_call15_pytest_full.log
grep -a "PARITY stage2\|passed\|failed\|Error\|FAILED" gpurun_out/r1_call15_pytest_full.log > gpurun_out/r1_call15_pytest.log
timeout 300 python tools/bench_stage2.py 8 > gpurun_out/r1_call15_stage2.jsonl 2>&1
cat gpurun_out/r1_call15_pytest.log | tail -8; cat gpurun_out/r1_call15_stage2.jsonl | tail -2
