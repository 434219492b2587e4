#!/bin/bash
# halo pixels on the idle quad lanes (no halo wave): conv parity tests, per-wave barrier accounting, layer microbenchmark, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r4_c6_tests.txt; tail -3 gpurun_out/r4_c6_tests.txt
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing2.so timeout 200 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c6_phase_waves.jsonl 2> gpurun_out/r4_c6_phase.err
timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c6_convbench.jsonl 2> gpurun_out/r4_c6_convbench.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c6_convbench.jsonl"):
    r = json.loads(l)
    print("  ", r["cin"], r["cout"], r["dims"], r["ups"], "bf16x3", r.get("bf16x3_tflops"), "f16x2", r.get("f16x2_tflops"))
PY
for p in bf16x3 f16x2; do
  EMO_CONV_PRECISION=$p timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c6_bench_$p.err | tee gpurun_out/r4_c6_bench_$p.json | cut -c1-200
done
