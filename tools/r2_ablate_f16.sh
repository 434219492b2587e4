#!/bin/bash
# fp16-operand kernel ablations (measurement only): which of fragment reads / staging / weight DMA bounds the K loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2ab}
mkdir -p $R/gpurun_out; cd $R
for v in "" _ab1 _ab2 _ab6 _ab7; do
  EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$v.so timeout 200 python tools/bench_conv.py 16 --quick --f16-only 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_f16_conv$v.jsonl
  echo "variant '$v'"
  python - <<PY
import json
for l in open("gpurun_out/${T}_f16_conv$v.jsonl"):
    if l.startswith("{"):
        x=json.loads(l); print(x["cin"],x["cout"],x["dims"],x["k"],x["ups"], x.get("f16_cfg3_tflops"))
PY
done
