#!/bin/bash
# bf16x3 conv kernel, second run: two accumulator sets + per-row weight buffers; ablation builds (timing only) and SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -q -s --timeout=300 2>&1 | grep -v amdgpu.ids | grep -a "PARITY\|passed\|failed\|^E " > gpurun_out/r3_bf16x3_pytest2.log
cat gpurun_out/r3_bf16x3_pytest2.log
timeout 300 python tools/bench_conv.py 16 --quick --bf16x3-only 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bf16x3_conv2.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r3_bf16x3_conv2.jsonl"):
    d = json.loads(l); print("default", d["cin"], d["cout"], d["dims"], d["ups"], d.get("bf16x3_ms"), d.get("bf16x3_tflops"))
PY
for v in s_a1 s_a2 s_a4 s_a7 s_a15 s_nopin; do
  EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bf16x3_conv2_$v.jsonl
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
print(v, [json.loads(l).get("bf16x3_tflops") for l in open(f"gpurun_out/r3_bf16x3_conv2_{v}.jsonl")])
PY
done
bash tools/pmc_conv.sh r3_bf16x3_512c 512 512 64 64 0 bf16x3 > gpurun_out/r3_bf16x3_pmc_512c.log 2>&1
bash tools/pmc_conv.sh r3_bf16x3_128c 128 128 512 512 0 bf16x3 > gpurun_out/r3_bf16x3_pmc_128c.log 2>&1
cat gpurun_out/r3_bf16x3_512c_pmc_conv.json gpurun_out/r3_bf16x3_128c_pmc_conv.json
