#!/bin/bash
# round-2 GPU call 2: tests after the fixes, CG8 sampler (tests + microbench + L2 counters), bench with the 64-row config preferred,
# images-in/out pipeline (animate_frames), stage-2 bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=r2c2
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
EMO_SAMPLER_LAYOUT=ndhwc timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ndhwc.json 2>> gpurun_out/${T}_bench.err
timeout 300 python tools/bench_sampler.py 16 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_sampler.jsonl
timeout 300 python tools/bench_pipeline.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_pipeline.jsonl
timeout 300 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stage2.jsonl
timeout 300 python tools/bench_driver.py 512 1 4 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_driver512.jsonl
bash tools/pmc_sampler.sh ${T} 16 0.05 cg8 > gpurun_out/${T}_pmc_sampler.log 2>&1
tail -5 gpurun_out/${T}_pytest.log; cut -c1-300 gpurun_out/${T}_bench.json; cut -c1-200 gpurun_out/${T}_bench_ndhwc.json
