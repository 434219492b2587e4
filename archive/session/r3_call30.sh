#!/bin/bash
# first GPU run of the bf16x3 conv kernel: parity, microbench on the decoder shapes, driver pass both ways, end-to-end parity
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -q -s --timeout=300 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r3_bf16x3_pytest.log
tail -25 gpurun_out/r3_bf16x3_pytest.log
timeout 300 python tools/bench_conv.py 16 --quick --bf16x3-only 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bf16x3_conv.jsonl
cat gpurun_out/r3_bf16x3_conv.jsonl | cut -c1-200
timeout 200 python tools/bench_driver.py 512 16 --bf16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bf16x3_driver.jsonl
timeout 200 python tools/bench_driver.py 512 16 --f32 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_bf16x3_driver.jsonl
cat gpurun_out/r3_bf16x3_driver.jsonl
EMO_CONV_PRECISION=bf16x3 timeout 600 python -m pytest tests/test_bench_config_parity_gpu.py -q -s --timeout=500 2>&1 | grep -a "PARITY\|passed\|failed\|Error" > gpurun_out/r3_bf16x3_e2e_parity.log
cat gpurun_out/r3_bf16x3_e2e_parity.log
