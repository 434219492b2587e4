#!/bin/bash
# round 6, call 4: emo_conv_igemm_f16w8 with 32-channel stages (the split layout's planes as two k-blocks): parity, layer microbench
# (even tile counts only / odd ones too / off), fp16 driver-pass breakdown
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_stage2_gpu.py -m gpu -q -s -k "fp16 or f16 or stage2" 2>&1 | F > gpurun_out/r6_c4_pytest_full.log
grep -a "passed\|failed\|Error\|FAILED" gpurun_out/r6_c4_pytest_full.log | tail -6
timeout 900 python -m pytest tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -s -k "f16 or fp16 or trained_like" 2>&1 | F > gpurun_out/r6_c4_pytest2_full.log
grep -a "PARITY.*f16 \|PARITY.*fp16\|passed\|failed\|Error\|FAILED" gpurun_out/r6_c4_pytest2_full.log | cut -c1-250 | tail -6
for mode in "1 0" "1 1" "0 0"; do set -- $mode
  EMO_F16_W8=$1 EMO_F16_W8_ODD=$2 timeout 300 python tools/bench_conv.py 16 --f16-only --quick 2>&1 | F > gpurun_out/r6_c4_conv_f16_$1$2.jsonl
done
python - <<'PY'
import json
def rows(f):
    out={}
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); out[(d.get("cin"),d.get("cout"),str(d.get("dims")),d.get("ups"))]=d
    return out
a,b,c=(rows(f"gpurun_out/r6_c4_conv_f16_{m}.jsonl") for m in ("10","11","00"))
for k in a:
    print(k, "w8 even-only", a[k].get("f16_cfg3_tflops"), "w8 all", b.get(k,{}).get("f16_cfg3_tflops"), "old", c.get(k,{}).get("f16_cfg3_tflops"))
PY
for mode in "1 0" "1 1" "0 0"; do set -- $mode
  echo "--- fp16 driver pass, EMO_F16_W8=$1 EMO_F16_W8_ODD=$2"
  EMO_F16_W8=$1 EMO_F16_W8_ODD=$2 timeout 300 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee gpurun_out/r6_c4_driver_f16_$1$2.jsonl | cut -c1-330
done
