#!/bin/bash
# sampler: fused multiply-add accumulation (variant bit 4) -- parity test, then the bench's sampler figures with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_grid_sample_gpu.py -m gpu -q -k "fma or kat or delta" 2>&1 | grep -v amdgpu.ids | tail -5
for v in 0 4; do
  EMO_SAMPLER_UV_VARIANT=$v EMO_SAMPLER_ROT_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2>> gpurun_out/r4_c17.err > gpurun_out/r4_c17_bench_v$v.json
  python - <<PY
import json
r = json.loads(open("gpurun_out/r4_c17_bench_v$v.json").read().strip().splitlines()[-1])
print("variant $v", r["value"], r["roofline_sampler"])
PY
done
