#!/bin/bash
# 128 x 256 tile of the fp16-operand kernel (config 6): parity tests, then per-layer / stage-2 / driver-pass timing with and
# without EMO_F16_CFG_G=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c20}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=900 -k "fp16 or f16 or aligned" 2>&1 | grep -v "amdgpu.ids" | tail -12 > gpurun_out/${T}_pytest.log
cat gpurun_out/${T}_pytest.log
for g in 0 1; do
  echo "EMO_F16_CFG_G=$g"
  EMO_F16_CFG_G=$g timeout 200 python tools/bench_conv.py 16 --quick --f16-only 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_f16_conv_g$g.jsonl
  python - <<PY
import json
for l in open("gpurun_out/${T}_f16_conv_g$g.jsonl"):
    if l.startswith("{"):
        x=json.loads(l); print(x["cin"],x["cout"],x["dims"],x["k"],x["ups"], x.get("f16_cfg3_tflops"))
PY
  EMO_F16_CFG_G=$g timeout 200 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids | grep f16
done
