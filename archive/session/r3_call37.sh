#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 500 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
cut -c1-300 gpurun_out/r3_bench.json
