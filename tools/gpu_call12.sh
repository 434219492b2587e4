#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for v in "" ws ws5; do
  if [ -z "$v" ]; then unset EMO_HIP_LIB; tag=default; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; tag=$v; fi
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -x 2>&1 | tail -3 > gpurun_out/r1_call12_pytest_$tag.log
  timeout 300 python tools/bench_conv.py 4 --quick > gpurun_out/r1_call12_conv_$tag.jsonl 2>&1
  timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r1_call12_bench_$tag.json 2>&1
done
for t in default ws ws5; do echo $t; tail -1 gpurun_out/r1_call12_pytest_$t.log; tail -1 gpurun_out/r1_call12_bench_$t.json | cut -c1-140; done
