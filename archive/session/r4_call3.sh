#!/bin/bash
# row epilogue with batched residual loads, sat_flag / run_if guard (ABI 6), CU de-phasing A/B (EMO_CONV_STAGGER)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_c3_tests.txt; tail -6 gpurun_out/r4_c3_tests.txt
T=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so
for st in 0 1; do
  EMO_CONV_STAGGER=$st EMO_HIP_LIB=$T timeout 200 python tools/conv_phase_timing.py 16 --real >> gpurun_out/r4_c3_phase.jsonl 2>> gpurun_out/r4_c3_phase.err
  EMO_CONV_STAGGER=$st timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c3_convbench_st$st.jsonl 2>> gpurun_out/r4_c3_convbench.err
done
tail -2 gpurun_out/r4_c3_phase.err
python - <<'PY'
import json
for st in (0, 1):
    print("stagger", st)
    for l in open(f"gpurun_out/r4_c3_convbench_st{st}.jsonl"):
        r = json.loads(l)
        print("  ", r["cin"], r["cout"], r["dims"], r["ups"], "bf16x3", r.get("bf16x3_tflops"), "f16x2", r.get("f16x2_tflops"))
PY
