#!/bin/bash
# persistent blocks by default (chaining removed): full GPU suite + bench in both split modes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r4_c15_tests.txt; tail -5 gpurun_out/r4_c15_tests.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c15_bench.err | tee gpurun_out/r4_c15_bench.json | cut -c1-200
EMO_CONV_PRECISION=bf16x3 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2>> gpurun_out/r4_c15_bench.err | tee gpurun_out/r4_c15_bench_bf16x3.json | cut -c1-200
