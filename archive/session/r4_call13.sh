#!/bin/bash
# persistent blocks with chained items (no prologue between items of one sample): parity, phase timing, A/B against EMO_S_CHAIN=0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py tests/test_stage2_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r4_c13_tests.txt; tail -4 gpurun_out/r4_c13_tests.txt
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 200 python tools/conv_phase_timing.py 16 --real > gpurun_out/r4_c13_phase.jsonl 2> gpurun_out/r4_c13_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c13_phase.jsonl"):
    r = json.loads(l)
    print(r["cin"], r["cout"], r["dims"], r["ups"], r["mode"], "ms", r["ms"], "TF", r["tflops"], "pro", r["prologue"]["med"], r["prologue"]["p90"], "k", r["kloop"]["med"], "epi", r["epilogue_issue"]["med"], "gap", r["gap_to_next_block"]["med"])
PY
for lib in "" _nochain; do
echo "lib$lib"
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$lib.so timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c13_convbench$lib.jsonl 2>> gpurun_out/r4_c13.err
python - <<PY
import json
for l in open("gpurun_out/r4_c13_convbench$lib.jsonl"):
    r = json.loads(l)
    print("  ", r["cin"], r["cout"], r["dims"], r["ups"], "bf16x3", r.get("bf16x3_tflops"), "f16x2", r.get("f16x2_tflops"))
PY
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$lib.so timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2>> gpurun_out/r4_c13.err | tee gpurun_out/r4_c13_bench$lib.json | cut -c1-200
EMO_CONV_PRECISION=bf16x3 EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$lib.so timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2>> gpurun_out/r4_c13.err | tee gpurun_out/r4_c13_bench_bf16x3$lib.json | cut -c1-200
done
