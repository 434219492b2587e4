#!/bin/bash
# table reads one step ahead (behind the conversion that last used the registers)
# conv parity, per-step cycles, microbenchmark, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r4_c10_tests.txt; tail -4 gpurun_out/r4_c10_tests.txt
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing3.so timeout 200 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c10_phase_steps.jsonl 2> gpurun_out/r4_c10_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c10_phase_steps.jsonl"):
    r = json.loads(l)
    if r["cin"] in (192, 512): print(r["cin"], r["cout"], r["dims"], r["ups"], r["mode"], "kloop/stage", r["kloop"]["med"] // (r["cin"] // 16), 'pro', r['prologue']['med'], 'epi', r['epilogue_issue']['med'], r.get("waves", {}).get("wave0"))
PY
timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c10_convbench.jsonl 2> gpurun_out/r4_c10_convbench.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c10_convbench.jsonl"):
    r = json.loads(l)
    print("  ", r["cin"], r["cout"], r["dims"], r["ups"], "bf16x3", r.get("bf16x3_tflops"), "f16x2", r.get("f16x2_tflops"))
PY
for p in bf16x3 f16x2; do
  EMO_CONV_PRECISION=$p timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c10_bench_$p.err | tee gpurun_out/r4_c10_bench_$p.json | cut -c1-200
done
