#!/bin/bash
# round 5, call 14: the pointwise kernel with its patch loads three stages ahead (parity, microbenchmark rows); the bench line with
# the metered pass that sizes its head start from the host's enqueue time -- run right behind a test run, as the driver does
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_bench_config_parity_gpu.py tests/test_two_ranks_gpu.py -m gpu -q -x 2>&1 | F | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --strong-frames 0 > gpurun_out/r5_c14_bench.json 2> gpurun_out/r5_c14_bench.err; tail -c 300 gpurun_out/r5_c14_bench.err | F
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c14_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_per_step"], d["roofline"]["share_of_step"], {k:(v["achieved"], v["share_of_step"], v["launches_per_step"]) for k,v in d["roofline_other_convs"].items()})
print(d["config"]["metered_pass"])
PY
timeout 300 python tools/bench_conv.py 16 --f16x2 2>&1 | F | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if d.get('pointwise_on_the_source_grid'): print(d['cin'],d['cout'],d['dims'],d['pointwise_on_the_source_grid'])
"
