#!/bin/bash
# Full round-end validation on the GPU box (what the driver runs, plus the evidence kept under profiles/):
#   all GPU parity tests, smoke(), the bench line (N=1; R512 and R256), `python bench.py --gpus 2` on one GPU (gloo, test hook),
#   stage-2 throughput, sampler / conv microbenchmarks, driver-pass breakdown, images-in/images-out pipeline, embedder parity +
#   timing, rocprofv3 kernel stats + PMC passes of the bench, sampler L1/L2 counters, fp16-mode evidence.
# usage: gpurun -- 'bash tools/gpu_validate.sh r3'   then   python tools/collect_profiles.py r3
#        (or in two calls, each well inside one gpurun limit: `... r3 tests` = parity tests + smoke, `... r3 bench` = everything else;
#         `... r3 bench-short` skips the evidence that no change since the last full run touches: sampler counters, fp16 mode,
#         stage 2, R256, embedders, pipeline)
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3}
mkdir -p $R/gpurun_out; cd $R
if [ $PART = all ] || [ $PART = tests ]; then
timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${TAG}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/${TAG}_pytest_full.log > gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_smoke.log
tail -3 gpurun_out/${TAG}_pytest.log; tail -1 gpurun_out/${TAG}_smoke.log
fi
[ $PART = tests ] && exit 0
SHORT=0; [ $PART = bench-short ] && SHORT=1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
EMO_CONV_PRECISION=bf16x3 timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16x3.json 2>> gpurun_out/${TAG}_bench.err
EMO_DIST_BACKEND=gloo EMO_FORCE_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_2ranks_1gpu.json 2>> gpurun_out/${TAG}_bench.err
timeout 400 python tools/bench_conv.py 16 --bf16x3 --f16x2 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_conv.jsonl
timeout 300 python tools/bench_driver.py 512 1 4 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_driver512.jsonl
timeout 300 python tools/bench_driver.py 512 1 16 --f32 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_driver512_f32.jsonl
timeout 300 python tools/bench_driver.py 512 1 16 --bf16x3 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_driver512_bf16x3.jsonl
[ -x tools/microbench/mfma_stream ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -Wno-unused-value tools/microbench/mfma_stream.hip -o tools/microbench/mfma_stream 2>/dev/null
timeout 60 tools/microbench/mfma_stream > gpurun_out/${TAG}_mfma_stream.jsonl 2>&1
bash tools/profile_bench.sh ${TAG}
if [ $SHORT = 0 ]; then
timeout 300 python bench.py --image-size 256 --batch 32 --no-cpu-baseline > gpurun_out/${TAG}_bench256.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_stage2.jsonl
timeout 300 python tools/bench_sampler.py 16 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_sampler.jsonl
timeout 300 python tools/bench_pipeline.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_pipeline.jsonl
timeout 300 python tools/probe_embedders.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_embedders.txt
bash tools/pmc_sampler.sh ${TAG}_ndhwc 16 0.05 ndhwc > gpurun_out/${TAG}_pmc_sampler_ndhwc.log 2>&1
bash tools/pmc_sampler.sh ${TAG}_ndhwc_small 16 0.02 ndhwc > gpurun_out/${TAG}_pmc_sampler_ndhwc_small.log 2>&1
bash tools/pmc_sampler.sh ${TAG}_p4tile 16 0.03 p4 > gpurun_out/${TAG}_pmc_sampler_p4tile.log 2>&1
bash tools/profile_f16.sh ${TAG}
fi
[ -f gpurun_out/${TAG}_pytest.log ] && tail -3 gpurun_out/${TAG}_pytest.log
cut -c1-400 gpurun_out/${TAG}_bench.json
