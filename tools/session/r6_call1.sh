#!/bin/bash
# round 6, call 1: the parity holes of the round-5 verdict (fp16-operand mode vs the ORACLE, the fp64-closeness statement on five layer
# shapes, smooth_pose as a global scan under 1 / 2 / 8 ranks), the batched crop-window launch, ABI 9 -- and a full bench line of the
# unchanged kernels with the new fields (roofline.sustained_peak / frac_of_sustained, host_affinity) for this round's box calibration
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 1500 python -m pytest tests/test_abi.py tests/test_conv_bf16x3_gpu.py tests/test_infer_gpu.py tests/test_two_ranks_gpu.py tests/test_bench_config_parity_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -s 2>&1 | F > gpurun_out/r6_c1_pytest_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/r6_c1_pytest_full.log > gpurun_out/r6_c1_pytest.log; tail -5 gpurun_out/r6_c1_pytest.log
echo "--- bench N=1 (full line)"
timeout 900 python bench.py > gpurun_out/r6_c1_bench_n1.json 2> gpurun_out/r6_c1_bench_n1.err; tail -c 600 gpurun_out/r6_c1_bench_n1.err | F
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_c1_bench_n1.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("N1", d["value"], d["ms_per_step"], r["frac"], r.get("sustained_peak"), r.get("frac_of_sustained"), r["avg_launch_ms"])
print("sustained", d.get("sustained_mfma"))
print("affinity", d.get("host_affinity"))
x=d.get("extras",{})
for k in ("stage1_f16_operands_fps","stage1_plus_stage2_f16_operands_fps","stage2_f16_fps","pipeline_frames_in_out_fps","emotion_driver_forward_fps","latency_b1_ms","bf16x3_split_fps"):
    print(k, x.get(k))
PY
echo "--- 2 ranks on one GPU (gloo): affinity record"
EMO_FORCE_DEVICE=0 EMO_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-sustained --strong-frames 21 > gpurun_out/r6_c1_bench_2ranks.json 2> gpurun_out/r6_c1_bench_2ranks.err; tail -c 400 gpurun_out/r6_c1_bench_2ranks.err | F
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r6_c1_bench_2ranks.json").read().strip().splitlines()[-1])
    print("2ranks", d["value"], d["host_affinity"])
except Exception as e:
    print("2ranks FAILED", e)
PY
lscpu | grep -i "numa\|model name\|^CPU(s)" | head; cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c | head
