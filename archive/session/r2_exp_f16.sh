#!/bin/bash
# measurement builds of the fp16-operand kernel (EMO_F16_EXPERIMENT): one block per CU / no MFMA-piece pinning
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for v in "" _x1 _x2; do
  echo "variant '$v'"
  EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$v.so timeout 120 python tools/fit_conv_overhead.py f16 512 128 4 2>&1 | grep -v amdgpu.ids
done
