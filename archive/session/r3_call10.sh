#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_grid_sample_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -4
for a in 0.01 0.02 0.03; do for c in 4 8 16; do timeout 300 python tools/bench_sampler_pair.py 16 $c $a 2>&1 | grep -v amdgpu; done; done > gpurun_out/r3c10_pair.jsonl
cat gpurun_out/r3c10_pair.jsonl | cut -c1-400
