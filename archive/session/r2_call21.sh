#!/bin/bash
# non-temporal output stores (-DEMO_CONV_NT_STORE=1) vs default: per-launch fixed cost of both conv kernels, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for v in "" _nt; do
  echo "variant '$v'"
  export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$v.so
  timeout 100 python tools/fit_conv_overhead.py f16 512 256 4 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 100 python tools/fit_conv_overhead.py f32 512 128 4 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass 2>/dev/null | tail -1 | cut -c1-110
done
