#!/bin/bash
# round 5, call 2: first run of the two-tile kernel (conv_igemm_f16x2_ct2.h): parity against the single-tile kernel and torch CPU,
# then the layer microbenchmark and the bench step with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -x -k "two_tile" 2>&1 | F | tail -15
echo "--- whole conv file"
timeout 900 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -x 2>&1 | F | tail -5
echo "--- bench, CT2 on / off"
for m in 1 0; do
EMO_CONV_CT2=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --strong-frames 0 > gpurun_out/r5_c2_bench_ct2_$m.json 2> gpurun_out/r5_c2_bench_ct2_$m.err; tail -c 400 gpurun_out/r5_c2_bench_ct2_$m.err | F
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c2_bench_ct2_$m.json").read().strip().splitlines()[-1])
print("CT2=$m", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"]["f16x2_layers_recomputed_after_range_check"])
PY
done
