#!/bin/bash
# round 6, call 9: animate_frames with the next batch's upload on a copy stream: wrapper tests (incl. the multi-rank product path with
# smooth_pose), the frames-in / frames-out figure, and one more full bench line (another box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 1500 python -m pytest tests/test_infer_gpu.py tests/test_two_ranks_gpu.py -m gpu -q -s -k "not bench_strong" 2>&1 | F > gpurun_out/r6_c9_pytest_full.log
grep -a "passed\|failed\|Error\|FAILED\|PARITY two" gpurun_out/r6_c9_pytest_full.log | cut -c1-300 | tail -6
timeout 300 python tools/bench_pipeline.py 512 1 16 2>&1 | F | tee gpurun_out/r6_c9_pipeline.jsonl | cut -c1-260
timeout 900 python bench.py > gpurun_out/r6_c9_bench.json 2> gpurun_out/r6_c9_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6_c9_bench.json") if l.startswith("{")][-1])
r=d["roofline"]; x=d["extras"]
print("N1", d["value"], d["ms_per_step"], "frac", r["frac"], "sustained", r.get("sustained_peak"), r.get("frac_of_sustained"), "traffic", r["traffic"])
print(d["sustained_mfma"]["bare"])
for k in ("pipeline_frames_in_out_fps","emotion_driver_forward_fps","latency_b1_ms","stage1_f16_operands_fps","stage1_plus_stage2_f16_operands_fps","stage2_f16_fps","bf16x3_split_fps","fp32_mfma_everywhere_fps"):
    print(k, x.get(k))
PY
