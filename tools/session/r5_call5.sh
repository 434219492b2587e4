#!/bin/bash
# round 5, call 5: what the epilogue's 5.6 k extra cycles per tile (decoder launch form vs plain) are made of: residual alone,
# statistics alone; the two-tile kernel with the second tile's residual loads issued in front of the first tile's epilogue
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
for v in "--real" "--res" "--stats"; do
EMO_HIP_LIB=emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 600 python tools/conv_phase_timing.py 16 $v --modes f16x2,ct2 --shapes 0,3 > gpurun_out/r5_c5_phase$v.jsonl 2> gpurun_out/r5_c5_phase$v.err; tail -c 300 gpurun_out/r5_c5_phase$v.err | F
python - <<PY
import json
for l in open("gpurun_out/r5_c5_phase$v.jsonl"):
    d=json.loads(l)
    print("$v", d["cin"],d["cout"],d["dims"],d["ups"],d["mode"],"ms",d["ms"],"tf",d["tflops"],"pro",d["prologue"]["med"],"k",d["kloop"]["med"],"epi",d["epilogue_issue"]["med"],"res_issue",d["epi_res_issue"]["med"],"e0",d["epi_half0"]["med"],"e1",d["epi_half1"]["med"],"tail",d["epi_tail"]["med"],"gap",d["gap_to_next_block"]["med"])
PY
done
timeout 300 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -x -k "two_tile" 2>&1 | F | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --strong-frames 0 > gpurun_out/r5_c5_bench.json 2> gpurun_out/r5_c5_bench.err; tail -c 400 gpurun_out/r5_c5_bench.err | F
python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c5_bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["share_of_step"])
PY
