#!/bin/bash
# graphs.Graphed with capture_error_mode="thread_local": the graph test, the 1-rank RCCL test, and a short bench run with an
# RCCL process group alive (forced 1-rank init) so that the watchdog thread exists during capture
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_infer_gpu.py -q -k graph --timeout=90 2>&1 | tail -1
EMO_DIST_FORCE_INIT=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 100 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r3_bench_rccl1.err | cut -c1-330
grep -v "amdgpu.ids" gpurun_out/r3_bench_rccl1.err | tail -3
