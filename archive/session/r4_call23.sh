#!/bin/bash
# straight-line epilogue for the decoders' launch form (compile-time act / residual form / alignment / full tile), accumulators
# read from their accumulation registers where they are used: parity, phase stamps, layer microbenchmark, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py tests/test_stage2_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r4_c23_tests.txt; tail -4 gpurun_out/r4_c23_tests.txt
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 200 python tools/conv_phase_timing.py 16 --real > gpurun_out/r4_c23_phase.jsonl 2> gpurun_out/r4_c23_phase.err
tail -3 gpurun_out/r4_c23_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c23_phase.jsonl"):
    r = json.loads(l)
    print(r["cin"], r["cout"], r["dims"], r["ups"], r["mode"], "ms", r["ms"], "TF", r["tflops"], "pro", r["prologue"]["med"], "k", r["kloop"]["med"], "epi", r["epilogue_issue"]["med"], "gap", r["gap_to_next_block"]["med"],
          "| res", r["epi_res_issue"]["med"], "h0", r["epi_half0"]["med"], "h1", r["epi_half1"]["med"], "tail", r["epi_tail"]["med"])
PY
timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c23_convbench.jsonl 2> gpurun_out/r4_c23_convbench.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c23_convbench.jsonl"):
    r = json.loads(l)
    print("  ", r["cin"], r["cout"], r["dims"], r["ups"], "bf16x3", r.get("bf16x3_tflops"), "f16x2", r.get("f16x2_tflops"))
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c23_bench.err | tee gpurun_out/r4_c23_bench.json | cut -c1-200
EMO_CONV_PRECISION=bf16x3 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c23_bench_bf16x3.err | tee gpurun_out/r4_c23_bench_bf16x3.json | cut -c1-200
