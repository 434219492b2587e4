#!/bin/bash
# round-2 GPU call 6: 256-position tiles on the 3-D WarpGenerator layers (configs 3 and 5 with KD = 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c6}
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python tools/bench_driver.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_driver512.jsonl
timeout 300 python tools/bench_conv.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv.jsonl
tail -3 gpurun_out/${T}_pytest.log; cut -c1-200 gpurun_out/${T}_bench.json; cat gpurun_out/${T}_driver512.jsonl | cut -c1-300
