#!/bin/bash
# round 5, call 19: the part of tools/gpu_validate.sh that `bench-short` leaves out, on the final code: R256 bench line, stage 2,
# sampler microbenchmark + counters, frames-in/frames-out pipeline, embedder parity + timing, fp16-operand mode evidence
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; TAG=r5
timeout 300 python bench.py --image-size 256 --batch 32 --no-cpu-baseline > gpurun_out/${TAG}_bench256.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_stage2.jsonl
timeout 300 python tools/bench_sampler.py 16 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_sampler.jsonl
timeout 300 python tools/bench_pipeline.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_pipeline.jsonl
timeout 300 python tools/probe_embedders.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_embedders.txt
bash tools/pmc_sampler.sh ${TAG}_ndhwc 16 0.05 ndhwc > gpurun_out/${TAG}_pmc_sampler_ndhwc.log 2>&1
bash tools/profile_f16.sh ${TAG}
tail -2 gpurun_out/${TAG}_stage2.jsonl | cut -c1-300; tail -1 gpurun_out/${TAG}_pipeline.jsonl | cut -c1-300; tail -8 gpurun_out/${TAG}_embedders.txt
