#!/bin/bash
# records of the final code: the one test that changed, the bench line alone (default + bf16 split), the per-layer
# microbenchmark, kernel trace + counter passes of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -k "layer_plan or case14 or case15" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
EMO_CONV_PRECISION=bf16x3 timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r4_bench_bf16x3.json 2>> gpurun_out/r4_bench.err
timeout 400 python tools/bench_conv.py 16 --bf16x3 --f16x2 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_conv.jsonl
bash tools/profile_bench.sh r4
cut -c1-300 gpurun_out/r4_bench.json
