#!/bin/bash
# round 3, call 7: shared-volume stencil kernel for the uv call: parity, sampler microbench, pair, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_grid_sample_gpu.py tests/test_sampler_tile_gpu.py tests/test_nets_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r3c7_pytest.log
tail -4 gpurun_out/r3c7_pytest.log
timeout 600 python tools/bench_sampler.py 4 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c7_sampler.jsonl
timeout 900 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r3c7_bench.json 2> gpurun_out/r3c7_bench.err
cut -c1-200 gpurun_out/r3c7_bench.json
