#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_grid_sample_gpu.py tests/test_sampler_tile_gpu.py tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -4
timeout 600 python tools/bench_sampler.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c17_sampler.jsonl
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps(dict(value=d['value'], sampler=d['roofline_sampler']['frac'], sampler_launch_ms=d['roofline_sampler']['avg_launch_ms'], conv=d['roofline']['frac'])))"; done
