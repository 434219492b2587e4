#!/bin/bash
# cycles per step index of the split kernel's K loop (EMO_S_TIMING=3)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing3.so timeout 200 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c7_phase_steps.jsonl 2> gpurun_out/r4_c7_phase.err
tail -2 gpurun_out/r4_c7_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c7_phase_steps.jsonl"):
    r = json.loads(l)
    print(r["cin"], r["cout"], r["dims"], r["mode"], "kloop/stage", r["kloop"]["med"] // (r["cin"] // 16), r.get("waves", {}).get("wave0"), r.get("waves", {}).get("wave3"))
PY
