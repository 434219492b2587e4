#!/bin/bash
# round 5, call 12: 32-row channel tiles of the fp16 split (the WarpGenerator's two 32-channel 3-D layers): parity, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -x -k "32_row or half_empty" 2>&1 | F | tail -12
echo "--- conv + nets + bench-config + stage2"
timeout 1500 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_bench_config_parity_gpu.py tests/test_nets_gpu.py tests/test_stage2_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | F | tail -4
for m in 1 0; do
EMO_F16X2_BM32=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --strong-frames 0 > gpurun_out/r5_c12_bench_bm32_$m.json 2> gpurun_out/r5_c12_bench_bm32_$m.err; tail -c 300 gpurun_out/r5_c12_bench_bm32_$m.err | F
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c12_bench_bm32_$m.json").read().strip().splitlines()[-1])
print("BM32=$m", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_per_step"], d["roofline"]["share_of_step"])
PY
done
