#!/bin/bash
# f16x2 (two-term fp16 split, SPLIT = 2 of the bf16x3 kernel): parity, microbench, driver pass, end-to-end parity of all modes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -q -s --timeout=300 2>&1 | grep -v amdgpu.ids | grep -a "PARITY\|passed\|failed\|^E " > gpurun_out/r3_f16x2_pytest.log
cat gpurun_out/r3_f16x2_pytest.log | tail -30
timeout 300 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_f16x2_conv.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r3_f16x2_conv.jsonl"):
    if l.startswith("{"):
        d = json.loads(l); print(d["cin"], d["cout"], d["dims"], d["ups"], "bf16x3", d.get("bf16x3_tflops"), "f16x2", d.get("f16x2_tflops"), d.get("f16x2_ms"))
PY
timeout 200 python tools/bench_driver.py 512 1 16 --f16x2 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_f16x2_driver.jsonl
cat gpurun_out/r3_f16x2_driver.jsonl
timeout 600 python -m pytest tests/test_bench_config_parity_gpu.py -q -s --timeout=500 -k trained_like 2>&1 | grep -a "PARITY\|passed\|failed\|Error\|^E " > gpurun_out/r3_f16x2_e2e_parity.log
cat gpurun_out/r3_f16x2_e2e_parity.log
