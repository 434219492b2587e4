#!/bin/bash
# round 5, call 9: the two-tile kernel with its first residual loads issued in front of the K loop: parity, phase stamps, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 600 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -x 2>&1 | F | tail -3
EMO_HIP_LIB=emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 600 python tools/conv_phase_timing.py 16 --real --modes f16x2,ct2 --shapes 0,1,3 > gpurun_out/r5_c9_phase.jsonl 2> gpurun_out/r5_c9_phase.err; tail -c 300 gpurun_out/r5_c9_phase.err | F
python - <<PY
import json
for l in open("gpurun_out/r5_c9_phase.jsonl"):
    d=json.loads(l)
    print(d["cin"],d["cout"],d["dims"],d["ups"],d["mode"],"ms",d["ms"],"tf",d["tflops"],"pro",d["prologue"]["med"],"k",d["kloop"]["med"],"epi",d["epilogue_issue"]["med"],"gap",d["gap_to_next_block"]["med"])
PY
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --strong-frames 0 > gpurun_out/r5_c9_bench.json 2> gpurun_out/r5_c9_bench.err; tail -c 400 gpurun_out/r5_c9_bench.err | F
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c9_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_per_step"], d["roofline"]["share_of_step"], {k:(v["achieved"], v["frac"], v["share_of_step"], v["launches_per_step"]) for k,v in d["roofline_other_convs"].items()})
PY
