#!/bin/bash
# what the barrier steps of the K loop are made of: per-step cycles with the weight DMA (1), the s_barrier (2), the quad loads (4)
# taken out (timing-only builds, results wrong)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for v in 1 2 4; do
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_t3a$v.so timeout 200 python tools/conv_phase_timing.py 16 > gpurun_out/r4_c11_steps_ablate$v.jsonl 2>> gpurun_out/r4_c11.err
python - <<PY
import json
for l in open("gpurun_out/r4_c11_steps_ablate$v.jsonl"):
    r = json.loads(l)
    if r["cin"] in (512,): print("ablate $v", r["cin"], r["cout"], r["dims"], r["mode"], "ms", r["ms"], "kloop/stage", r["kloop"]["med"] // (r["cin"] // 16), r.get("waves", {}).get("wave0"))
PY
done
tail -2 gpurun_out/r4_c11.err
