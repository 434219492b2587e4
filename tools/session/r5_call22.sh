#!/bin/bash
# round 5, call 22: the final code once more (operand conversion of the fp16 split on pairs): the whole GPU suite + smoke, the
# counter passes and kernel trace of the bench command FIRST (so that the bench line quotes HBM traffic measured on these kernel
# sources), then the bench line with extras and the bf16-split line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
bash tools/gpu_validate.sh r5 tests
bash tools/profile_bench.sh r5
python tools/collect_traffic.py r5 > gpurun_out/r5_collect_traffic.log 2>&1; tail -3 gpurun_out/r5_collect_traffic.log
timeout 600 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
EMO_CONV_PRECISION=bf16x3 timeout 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r5_bench_bf16x3.json 2>> gpurun_out/r5_bench.err
cut -c1-300 gpurun_out/r5_bench.json
