#!/bin/bash
# round 3, call 1: parity of the LDS-staged tile sampler on the GPU, then the tuning sweep vs the direct-gather kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_sampler_tile_gpu.py tests/test_grid_sample_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -25 > gpurun_out/r3c1_pytest.log
tail -5 gpurun_out/r3c1_pytest.log
timeout 900 python tools/bench_sampler_tile.py 16 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c1_sampler_tile.jsonl
tail -3 gpurun_out/r3c1_sampler_tile.jsonl
