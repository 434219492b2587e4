#!/bin/bash
# round 6, call 5: phase stamps of the eight-wave kernel with plain fp16 operands (where do its half-stages go?), and of the fp16
# split's two kernels on all-zero operands (the schedule without the power limit)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m emoportraits_amd.build --variant timing EMO_S_TIMING=1 > gpurun_out/r6_c5_build_timing.log 2>&1; tail -1 gpurun_out/r6_c5_build_timing.log
export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so
timeout 600 python tools/conv_phase_timing.py 16 --real --modes f16w8 --shapes 0,4 2>&1 | F > gpurun_out/r6_c5_phase_f16w8.jsonl
timeout 600 python tools/conv_phase_timing.py 16 --real --modes ct2,w8 --shapes 0,4 --zeros 2>&1 | F > gpurun_out/r6_c5_phase_zeros.jsonl
timeout 600 python tools/conv_phase_timing.py 16 --real --modes f16w8 --shapes 0 --zeros 2>&1 | F > gpurun_out/r6_c5_phase_f16w8_zeros.jsonl
python - <<'PY'
import json
for f in ("r6_c5_phase_f16w8","r6_c5_phase_zeros","r6_c5_phase_f16w8_zeros"):
    print("---",f)
    for l in open(f"gpurun_out/{f}.jsonl"):
        if not l.startswith("{"): print(l.strip()[:300]); continue
        d=json.loads(l)
        print(d["mode"], d["cin"], d["cout"], d["dims"], "ms", d["ms"], "TF", d["tflops"], "pro", d["prologue"]["med"], "kloop", d["kloop"]["med"], "epi", d["epilogue_issue"]["med"], "gap", d["gap_to_next_block"]["med"], "clk", d.get("eff_clock_ghz"), "stages", d["stages"])
PY
