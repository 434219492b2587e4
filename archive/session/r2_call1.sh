#!/bin/bash
# round-2 GPU call 1: full GPU test suite on the re-pipelined conv + fused GN statistics, bench, A/B against the round-1 schedule
# (libemoportraits_hip_pipe1.so), 2 ranks on one GPU through `python bench.py --gpus 2`, sampler L2 counters of the round-1 kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=r2c1
mkdir -p $R/gpurun_out; cd $R
rocprofv3 -L 2>/dev/null | grep -o "TC[CP]_[A-Za-z0-9_]*" | sort -u > gpurun_out/${T}_tc_counters.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_pipe1.so timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_pipe1.json 2>> gpurun_out/${T}_bench.err
timeout 300 python tools/bench_conv.py 16 --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_default.jsonl
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_pipe1.so timeout 300 python tools/bench_conv.py 16 --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_pipe1.jsonl
EMO_DIST_BACKEND=gloo EMO_FORCE_DEVICE=0 timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline > gpurun_out/${T}_bench_gpus2.json 2> gpurun_out/${T}_bench_gpus2.err
bash tools/pmc_sampler.sh ${T} 16 > gpurun_out/${T}_pmc_sampler.log 2>&1
tail -5 gpurun_out/${T}_pytest.log; cut -c1-300 gpurun_out/${T}_bench.json; cut -c1-200 gpurun_out/${T}_bench_pipe1.json; cut -c1-200 gpurun_out/${T}_bench_gpus2.json
