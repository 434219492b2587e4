#!/bin/bash
# round 6, call 11: non-temporal output stores in the split kernels' epilogues (-DEMO_CONV_NT_STORE=1), A B A B against the product
# build on one box: does keeping 128 KB of output per item out of the L2 leave more of it to the weight stream?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m emoportraits_amd.build --variant nt EMO_CONV_NT_STORE=1 > gpurun_out/r6_c11_build.log 2>&1; tail -1 gpurun_out/r6_c11_build.log
for i in 1 2; do
  for v in nt product; do
    if [ $v = nt ]; then export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_nt.so; else unset EMO_HIP_LIB; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sustained --strong-frames 0 > gpurun_out/r6_c11_bench_${v}_$i.json 2> gpurun_out/r6_c11_bench.err
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r6_c11_bench_${v}_$i.json") if l.startswith("{")][-1])
print("$v run $i:", d["value"], "fps", d["ms_per_step"], "ms  frac", d["roofline"]["frac"])
PY
  done
done
for v in nt product; do
  if [ $v = nt ]; then export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_nt.so; else unset EMO_HIP_LIB; fi
  echo "--- fp16 driver pass, $v"
  timeout 300 python tools/bench_driver.py 512 16 --f16 2>&1 | F | cut -c1-330
done
