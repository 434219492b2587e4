#!/bin/bash
# round-2 GPU call 4: tests; bench (64x256 conv tile default, sampler pairs in chunks of 4) and A/B switches; sampler microbench with
# the brick variant; rocprofv3 kernel stats + PMC of the bench (profile_bench.sh); fp16-mode evidence (profile_f16.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c4}
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
EMO_SAMPLER_UV_VARIANT=12 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_brick.json 2>> gpurun_out/${T}_bench.err
EMO_SAMPLER_CHUNK=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_nochunk.json 2>> gpurun_out/${T}_bench.err
timeout 300 python bench.py --image-size 256 --batch 32 --no-cpu-baseline > gpurun_out/${T}_bench256.json 2>> gpurun_out/${T}_bench.err
timeout 300 python tools/bench_sampler.py 16 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_sampler.jsonl
timeout 300 python tools/bench_pipeline.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_pipeline.jsonl
timeout 300 python tools/bench_driver.py 512 1 4 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_driver512.jsonl
timeout 300 python tools/bench_conv.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv.jsonl
bash tools/profile_bench.sh ${T}
bash tools/profile_f16.sh ${T}
tail -4 gpurun_out/${T}_pytest.log; cut -c1-200 gpurun_out/${T}_bench.json; cut -c1-200 gpurun_out/${T}_bench_brick.json; cut -c1-200 gpurun_out/${T}_bench_nochunk.json
