#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q --timeout=600 2>&1 | tail -3 > gpurun_out/r1_call14_pytest.log
for v in "" noxcd; do
  if [ -z "$v" ]; then unset EMO_HIP_LIB; tag=default; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; tag=$v; fi
  timeout 300 python tools/bench_conv.py 4 --quick > gpurun_out/r1_call14_conv_$tag.jsonl 2>&1
  timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r1_call14_bench_$tag.json 2>&1
done
unset EMO_HIP_LIB
timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch 32 > gpurun_out/r1_call14_bench_b32.json 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-source-pass > $R/gpurun_out/r1_call14_prof_fetch.log 2>&1
cd $R; python tools/summarize_rocprof.py pmc gpurun_out/prof_fetch gpurun_out/r1_call14_pmc_fetch.json; rm -rf gpurun_out/prof_fetch
tail -1 gpurun_out/r1_call14_pytest.log; for t in default noxcd b32; do echo $t; tail -1 gpurun_out/r1_call14_bench_$t.json | cut -c1-140; done
