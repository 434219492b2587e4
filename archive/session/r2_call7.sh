#!/bin/bash
# round-2 GPU call 7: A/B of the conv MFMA-stream variants (LDS operand prefetch, s_setprio) -- bench + conv microbench per library build
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2c7}
mkdir -p $R/gpurun_out; cd $R
for V in default ldspf prio both default2; do
  if [ "$V" = "default" ] || [ "$V" = "default2" ]; then unset EMO_HIP_LIB; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$V.so; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 8 > gpurun_out/${T}_bench_$V.json 2>> gpurun_out/${T}_bench.err
  timeout 300 python tools/bench_conv.py 16 --quick 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_conv_$V.jsonl
done
for V in default ldspf prio both default2; do cut -c1-160 gpurun_out/${T}_bench_$V.json; done
