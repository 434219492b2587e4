#!/bin/bash
# round 6, call 6: emo_conv_igemm_f16w8 without scratch traffic in its epilogue (scale / shift entries no longer carried across items,
# second tile's residual issued behind the first tile): parity, layer microbench, driver breakdown, phase stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_bf16x3_gpu.py -m gpu -q -s -k "fp16_operands or two_tile" 2>&1 | F > gpurun_out/r6_c6_pytest_full.log
grep -a "passed\|failed\|Error\|FAILED" gpurun_out/r6_c6_pytest_full.log | tail -4
for mode in "1 0" "1 1" "0 0"; do set -- $mode
  EMO_F16_W8=$1 EMO_F16_W8_ODD=$2 timeout 300 python tools/bench_conv.py 16 --f16-only --quick 2>&1 | F > gpurun_out/r6_c6_conv_f16_$1$2.jsonl
done
python - <<'PY'
import json
def rows(f):
    out={}
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); out[(d.get("cin"),d.get("cout"),str(d.get("dims")),d.get("ups"))]=d
    return out
a,b,c=(rows(f"gpurun_out/r6_c6_conv_f16_{m}.jsonl") for m in ("10","11","00"))
for k in a:
    print(k, "w8 even-only", a[k].get("f16_cfg3_tflops"), "w8 all", b.get(k,{}).get("f16_cfg3_tflops"), "old", c.get(k,{}).get("f16_cfg3_tflops"))
PY
for mode in "1 0" "1 1" "0 0"; do set -- $mode
  echo "--- fp16 driver pass, EMO_F16_W8=$1 EMO_F16_W8_ODD=$2"
  EMO_F16_W8=$1 EMO_F16_W8_ODD=$2 timeout 300 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee gpurun_out/r6_c6_driver_f16_$1$2.jsonl | cut -c1-330
done
timeout 900 python -m emoportraits_amd.build --variant timing EMO_S_TIMING=1 > gpurun_out/r6_c6_build_timing.log 2>&1; tail -1 gpurun_out/r6_c6_build_timing.log
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so EMO_F16_W8_ODD=1 timeout 600 python tools/conv_phase_timing.py 16 --real --modes f16w8 --shapes 0,1,3,4 2>&1 | F > gpurun_out/r6_c6_phase_f16w8.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r6_c6_phase_f16w8.jsonl"):
    if not l.startswith("{"): print(l.strip()[:300]); continue
    d=json.loads(l)
    print(d["mode"], d["cin"], d["cout"], d["dims"], "ms", d["ms"], "TF", d["tflops"], "pro", d["prologue"]["med"], "kloop", d["kloop"]["med"], "epi", d["epilogue_issue"]["med"], "gap", d["gap_to_next_block"]["med"], "clk", d.get("eff_clock_ghz"), "stages", d["stages"])
PY
