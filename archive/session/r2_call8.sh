#!/bin/bash
# round-2 GPU call 8: the rewritten fp16-operand kernel (32x32x16 MFMA, lane-owns-8-channels staging, 64x256 tile)
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c8}
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 300 python tools/bench_conv.py 16 --quick --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_f16_conv.jsonl
timeout 300 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage2.jsonl
timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_f16_driver512.jsonl
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
grep -a "passed\|failed" gpurun_out/${T}_pytest.log | tail -2; grep -a "^FAILED" gpurun_out/${T}_pytest_full.log | head; cut -c1-400 gpurun_out/${T}_f16_conv.jsonl | head -4; cat gpurun_out/${T}_stage2.jsonl; cut -c1-200 gpurun_out/${T}_bench.json
