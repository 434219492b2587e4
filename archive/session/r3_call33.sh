#!/bin/bash
# bf16x3 conv: persistent blocks A/B, 8 x 32 tiles, driver pass at 16 frames and 1 frame
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -q -s --timeout=300 2>&1 | grep -v amdgpu.ids | grep -a "PARITY\|passed\|failed\|^E " > gpurun_out/r3_bf16x3_pytest3.log
cat gpurun_out/r3_bf16x3_pytest3.log
for pz in 1 0; do
  EMO_CONV_BF16X3_PERSISTENT=$pz timeout 300 python tools/bench_conv.py 16 --quick --bf16x3-only 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bf16x3_conv3_p$pz.jsonl
  python - $pz <<'PY'
import json, sys
print("persistent", sys.argv[1], [(json.loads(l).get("bf16x3_ms"), json.loads(l).get("bf16x3_tflops")) for l in open(f"gpurun_out/r3_bf16x3_conv3_p{sys.argv[1]}.jsonl") if l.startswith("{")])
PY
done
timeout 300 python tools/bench_driver.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bf16x3_driver3.jsonl
cat gpurun_out/r3_bf16x3_driver3.jsonl
