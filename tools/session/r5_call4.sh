#!/bin/bash
# round 5, call 4: per-tile phase stamps of the single-tile and the two-tile kernel (measurement build), with and without the
# decoder's launch form (residual + tile statistics)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
for real in "--real" ""; do
EMO_HIP_LIB=emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 600 python tools/conv_phase_timing.py 16 $real --modes f16x2,ct2 > gpurun_out/r5_c4_phase$real.jsonl 2> gpurun_out/r5_c4_phase$real.err; tail -c 300 gpurun_out/r5_c4_phase$real.err | F
python - <<PY
import json
for l in open("gpurun_out/r5_c4_phase$real.jsonl"):
    d=json.loads(l)
    print("$real", d["cin"],d["cout"],d["dims"],d["ups"],d["mode"],"ms",d["ms"],"tf",d["tflops"],"pro",d["prologue"]["med"],"k",d["kloop"]["med"],"epi",d["epilogue_issue"]["med"],"res_issue",d["epi_res_issue"]["med"],"e0",d["epi_half0"]["med"],"e1",d["epi_half1"]["med"],"tail",d["epi_tail"]["med"],"gap",d["gap_to_next_block"]["med"],"clk",d["eff_clock_ghz"], "stages", d["stages"])
PY
done
