#!/bin/bash
# round 5, call 8: first run of the pointwise kernel (conv_igemm_f16x2_p1.h): parity, then the bench with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -x -k "pointwise" 2>&1 | F | tail -15
echo "--- conv + bench-config parity + nets"
timeout 1200 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_bench_config_parity_gpu.py tests/test_nets_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | F | tail -5
echo "--- bench, pointwise kernel on / off"
for m in 1 0; do
EMO_F16X2_POINTWISE=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --strong-frames 0 > gpurun_out/r5_c8_bench_p1_$m.json 2> gpurun_out/r5_c8_bench_p1_$m.err; tail -c 400 gpurun_out/r5_c8_bench_p1_$m.err | F
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c8_bench_p1_$m.json").read().strip().splitlines()[-1])
print("P1=$m", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_per_step"], d["roofline"]["share_of_step"], {k:(v["achieved"], v["share_of_step"], v["launches_per_step"]) for k,v in d["roofline_other_convs"].items()}, d["config"]["f16x2_layers_recomputed_after_range_check"])
PY
done
