#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q --timeout=900 -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r1_call9_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/r1_call9_pytest_full.log > gpurun_out/r1_call9_pytest.log
for v in "" glds0; do
  if [ -z "$v" ]; then unset EMO_HIP_LIB; tag=default; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; tag=$v; fi
  timeout 300 python tools/bench_conv.py 4 --quick > gpurun_out/r1_call9_conv_$tag.jsonl 2>&1
  timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r1_call9_bench_$tag.json 2>&1
done
unset EMO_HIP_LIB
timeout 300 python tools/bench_conv.py 4 > gpurun_out/r1_call9_conv_full.jsonl 2>&1
tail -4 gpurun_out/r1_call9_pytest.log; for t in default glds0; do echo $t; tail -1 gpurun_out/r1_call9_bench_$t.json | cut -c1-200; done
