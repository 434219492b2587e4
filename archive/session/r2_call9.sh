#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c9}
mkdir -p $R/gpurun_out; cd $R
timeout 300 python tools/diag_f16.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_diag_f16.jsonl
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_stage2_gpu.py -m gpu -q --timeout=900 -s -k "fp16 or f16" 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_f16.log
timeout 300 python tools/bench_conv.py 16 --quick --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_f16_conv.jsonl
timeout 300 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage2.jsonl
timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_f16_driver512.jsonl
cat gpurun_out/${T}_diag_f16.jsonl; grep -a "passed\|failed\|PARITY" gpurun_out/${T}_pytest_f16.log | cut -c1-220 | tail -14; python - <<PY
import json
for l in open("gpurun_out/${T}_f16_conv.jsonl"):
    if l.startswith("{"):
        x=json.loads(l); print(x["cin"],x["cout"],x["dims"],x["k"],x["ups"], x.get("hip_cfg3_tflops"), x.get("f16_cfg3_tflops"))
PY
cat gpurun_out/${T}_stage2.jsonl; cut -c1-300 gpurun_out/${T}_f16_driver512.jsonl
