#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_grid_sample_gpu.py -m gpu -q --timeout=600 2>&1 | tail -3 > gpurun_out/r1_call13_pytest.log
timeout 300 python tools/bench_sampler.py 16 64 > gpurun_out/r1_call13_sampler.jsonl 2>&1
tail -2 gpurun_out/r1_call13_pytest.log; grep -a "var1\|var2\|var9\|rot_theta_unshared/cl2ncdhw\"\|uv/cl\"" gpurun_out/r1_call13_sampler.jsonl | cut -c1-150
