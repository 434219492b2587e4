#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for v in "" ablate1 ablate2 ablate3; do
  if [ -z "$v" ]; then unset EMO_HIP_LIB; tag=default; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; tag=$v; fi
  timeout 300 python tools/bench_conv.py 4 --quick > gpurun_out/r1_call7_conv_$tag.jsonl 2>&1
done
tail -3 gpurun_out/r1_call7_conv_ablate3.jsonl
