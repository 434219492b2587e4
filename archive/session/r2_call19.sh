#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c19}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_stage2_gpu.py -m gpu -q -x --timeout=900 2>&1 | grep -v "amdgpu.ids" | tail -8 > gpurun_out/${T}_pytest.log
timeout 300 python tools/bench_conv.py 16 --quick --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv.jsonl
timeout 300 python bench.py --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_bench.json
timeout 100 python tools/fit_conv_overhead.py f16 512 128 4 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/${T}_fit.jsonl
timeout 100 python tools/fit_conv_overhead.py f32 512 128 4 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/${T}_fit.jsonl
timeout 300 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage2.jsonl
cat gpurun_out/${T}_pytest.log
python - <<PY
import json
for l in open("gpurun_out/${T}_conv.jsonl"):
    if l.startswith("{"):
        x=json.loads(l); print(x["cin"],x["cout"],x["dims"],x["k"],x["ups"], [x.get(f"hip_cfg{c}_tflops") for c in (0,1,3,5)], x.get("f16_cfg3_tflops"))
x=json.loads(open("gpurun_out/${T}_bench.json").read())
print(x["value"], x["roofline"])
PY
cat gpurun_out/${T}_fit.jsonl gpurun_out/${T}_stage2.jsonl
