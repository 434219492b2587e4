#!/bin/bash
# round 3, call 4: packed-4 brick-ordered direct gather: parity, then the pair at chunk 4 and 16 vs the round-2 kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_sampler_tile_gpu.py tests/test_grid_sample_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r3c4_pytest.log
tail -3 gpurun_out/r3c4_pytest.log
timeout 600 python tools/bench_sampler_pair.py 16 4 0.03 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c4_pair_chunk4.jsonl
timeout 600 python tools/bench_sampler_pair.py 16 16 0.03 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c4_pair_chunk16.jsonl
timeout 600 python tools/bench_sampler_pair.py 16 8 0.03 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c4_pair_chunk8.jsonl
timeout 600 python tools/bench_sampler_pair.py 16 4 0.6 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c4_pair_chunk4_wild.jsonl
