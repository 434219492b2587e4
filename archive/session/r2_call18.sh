#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c18}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout=900 2>&1 | grep -v "amdgpu.ids" | tail -5 > gpurun_out/${T}_pytest_kernels.log
timeout 300 python tools/bench_conv.py 16 --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv.jsonl
timeout 300 python bench.py --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench.json
cat gpurun_out/${T}_pytest_kernels.log
python - <<PY
import json
for l in open("gpurun_out/${T}_conv.jsonl"):
    if l.startswith("{"):
        x=json.loads(l); print(x["cin"],x["cout"],x["dims"],x["k"],x["ups"], [x.get(f"hip_cfg{c}_tflops") for c in (0,1,3,5)])
x=json.loads(open("gpurun_out/${T}_bench.json").read())
print(x["value"], x["roofline"])
PY
