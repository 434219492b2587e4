#!/bin/bash
# phase stamps (EMO_S_TIMING=1) of the final kernels, real decoder launch form (residual + tile statistics)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 200 python tools/conv_phase_timing.py 16 --real > gpurun_out/r4_c20_phase.jsonl 2> gpurun_out/r4_c20_phase.err
tail -3 gpurun_out/r4_c20_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c20_phase.jsonl"):
    r = json.loads(l)
    print(r["cin"], r["cout"], r["dims"], r["ups"], r["mode"], "ms", r["ms"], "TF", r["tflops"], "pro", r["prologue"]["med"], "k", r["kloop"]["med"], "k/stage", r["kloop"]["med"] // (r["cin"] // 16), "epi", r["epilogue_issue"]["med"], "gap", r["gap_to_next_block"]["med"])
PY
