#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_grid_sample_gpu.py -m gpu -q --timeout=600 2>&1 | tail -3 > gpurun_out/r1_call11_pytest.log
timeout 300 python tools/bench_sampler.py 16 64 > gpurun_out/r1_call11_sampler.jsonl 2>&1
# world_size-2 run of bench.py on ONE GPU (gloo + forced device): exercises sharding, source broadcast, MAX-over-ranks timing
EMO_DIST_BACKEND=gloo EMO_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 > gpurun_out/r1_call11_bench_2ranks_1gpu.json 2> gpurun_out/r1_call11_bench_2ranks.err
bash tools/profile_bench.sh r1
tail -2 gpurun_out/r1_call11_pytest.log; tail -1 gpurun_out/r1_call11_bench_2ranks_1gpu.json | cut -c1-250; tail -3 gpurun_out/r1_call11_bench_2ranks.err
