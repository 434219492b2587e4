#!/bin/bash
# round 6, call 3: plain fp16 operands on the eight-wave two-tile kernel (emo_conv_igemm_f16w8, BASELINE configs[4]): parity, the
# driver-pass breakdown and the layer microbench with it on / off; the power evidence for the fp16 SPLIT (all-zero operands: four
# waves vs eight); one counter trace of the sampler on this round's binaries
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_stage2_gpu.py tests/test_abi.py -m gpu -q -s -k "fp16 or f16 or stage2 or resize or abi" 2>&1 | F > gpurun_out/r6_c3_pytest_full.log
grep -a "passed\|failed\|Error\|FAILED" gpurun_out/r6_c3_pytest_full.log | tail -6
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -s -k "f16 or fp16 or trained_like" 2>&1 | F > gpurun_out/r6_c3_pytest2_full.log
grep -a "PARITY.*f16\|PARITY.*fp16\|passed\|failed\|Error\|FAILED" gpurun_out/r6_c3_pytest2_full.log | cut -c1-250 | tail -8
echo "--- fp16 driver-pass breakdown, EMO_F16_W8 = 1 / 0"
EMO_F16_W8=1 timeout 300 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee gpurun_out/r6_c3_driver_f16_w8.jsonl | cut -c1-400
EMO_F16_W8=0 timeout 300 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee gpurun_out/r6_c3_driver_f16_old.jsonl | cut -c1-400
echo "--- fp16 layer microbench, EMO_F16_W8 = 1 / 0"
EMO_F16_W8=1 timeout 300 python tools/bench_conv.py 16 --f16-only --quick 2>&1 | F > gpurun_out/r6_c3_conv_f16_w8.jsonl
EMO_F16_W8=0 timeout 300 python tools/bench_conv.py 16 --f16-only --quick 2>&1 | F > gpurun_out/r6_c3_conv_f16_old.jsonl
python - <<'PY'
import json
def rows(f):
    out={}
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); out[(d.get("cin"),d.get("cout"),str(d.get("dims")),d.get("ups"))]=d
    return out
a,b=rows("gpurun_out/r6_c3_conv_f16_w8.jsonl"),rows("gpurun_out/r6_c3_conv_f16_old.jsonl")
for k in a:
    print(k, "f16w8", a[k].get("f16_cfg3_tflops"), "old", b.get(k,{}).get("f16_cfg3_tflops"))
PY
echo "--- fp16 SPLIT on all-zero operands: eight waves vs four"
EMO_CONV_W8=1 timeout 300 python tools/bench_conv.py 16 --bf16x3-only --f16x2 --quick --zeros 2>&1 | F > gpurun_out/r6_c3_conv_zeros_w8.jsonl
EMO_CONV_W8=0 timeout 300 python tools/bench_conv.py 16 --bf16x3-only --f16x2 --quick --zeros 2>&1 | F > gpurun_out/r6_c3_conv_zeros_ct2.jsonl
python - <<'PY'
import json
def rows(f):
    out={}
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); out[(d.get("cin"),d.get("cout"),str(d.get("dims")),d.get("ups"))]=d
    return out
a,b=rows("gpurun_out/r6_c3_conv_zeros_w8.jsonl"),rows("gpurun_out/r6_c3_conv_zeros_ct2.jsonl")
for k in a:
    print(k, "zeros: w8", a[k].get("f16x2_tflops"), "ct2", b.get(k,{}).get("f16x2_tflops"))
PY
echo "--- bench extras (f16 figures)"
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6_c3_bench.json 2> gpurun_out/r6_c3_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_c3_bench.json").read().strip().splitlines()[-1])
print("N1", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_sustained"))
x=d.get("extras",{})
for k in ("stage1_f16_operands_fps","stage1_plus_stage2_f16_operands_fps","stage2_f16_fps","stage2_f16x2_fps","pipeline_frames_in_out_fps","emotion_driver_forward_fps","latency_b1_ms"):
    print(k, x.get(k))
PY
echo "--- sampler counters (this round's binaries)"
timeout 600 bash tools/pmc_sampler.sh r6_ndhwc 16 0.05 ndhwc > gpurun_out/r6_c3_pmc_sampler.log 2>&1; tail -30 gpurun_out/r6_c3_pmc_sampler.log
