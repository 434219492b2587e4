#!/bin/bash
# round 3, call 3: 16-byte transposed NCDHW stores; the sampler pair at chunk 4 for a list of tunings
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_sampler_tile_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -5 > gpurun_out/r3c3_pytest.log
tail -3 gpurun_out/r3c3_pytest.log
timeout 900 python tools/bench_sampler_pair.py 16 4 0.03 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c3_pair.jsonl
tail -2 gpurun_out/r3c3_pair.jsonl
