#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3_bench_graph.json 2> gpurun_out/r3_bench_graph.err
tail -3 gpurun_out/r3_bench_graph.err | grep -v amdgpu; cut -c1-300 gpurun_out/r3_bench_graph.json
EMO_DIST_BACKEND=gloo EMO_FORCE_DEVICE=0 timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline > gpurun_out/r3_bench_2ranks_1gpu.json 2>> gpurun_out/r3_bench_graph.err
tail -c 400 gpurun_out/r3_bench_2ranks_1gpu.json | cut -c1-200
