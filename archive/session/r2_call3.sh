#!/bin/bash
# round-2 GPU call 3: tests (CG8 sampler dispatch fixed, 64x256 conv tile), bench default / with the 64x256 tile / round-1 sampler
# layout, per-instantiation kernel trace of the bench, conv + sampler microbenchmarks, pipeline, L2 counters of the CG8 sampler
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c3}
mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/${T}_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED\|error" gpurun_out/${T}_pytest_full.log > gpurun_out/${T}_pytest.log
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
EMO_CONV_CFG_D=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_cfgD.json 2>> gpurun_out/${T}_bench.err
EMO_SAMPLER_LAYOUT=ndhwc timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_ndhwc.json 2>> gpurun_out/${T}_bench.err
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass > $R/gpurun_out/${T}_prof_kt.log 2>&1)
python tools/summarize_rocprof.py stats gpurun_out/prof_kt gpurun_out/${T}_kernel_stats.csv; rm -rf gpurun_out/prof_kt
(cd /tmp && EMO_CONV_CFG_D=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass > $R/gpurun_out/${T}_prof_kt_cfgD.log 2>&1)
python tools/summarize_rocprof.py stats gpurun_out/prof_kt gpurun_out/${T}_kernel_stats_cfgD.csv; rm -rf gpurun_out/prof_kt
timeout 300 python tools/bench_conv.py 16 --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv.jsonl
timeout 300 python tools/bench_sampler.py 16 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_sampler.jsonl
timeout 300 python tools/bench_pipeline.py 512 1 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_pipeline.jsonl
timeout 300 python tools/bench_driver.py 512 1 4 16 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_driver512.jsonl
bash tools/pmc_sampler.sh ${T} 16 0.05 cg8 > gpurun_out/${T}_pmc_sampler.log 2>&1
tail -4 gpurun_out/${T}_pytest.log; cut -c1-200 gpurun_out/${T}_bench.json; cut -c1-200 gpurun_out/${T}_bench_cfgD.json; cut -c1-200 gpurun_out/${T}_bench_ndhwc.json
