#!/bin/bash
# round 6, call 28: odd channel-tile counts of the plain-fp16 mode: pairs on the eight-wave kernel + the last tile on the older
# kernel (emo_conv_igemm_f16w8_rest, ABI 10; product) against EMO_F16_W8_REST=0 (half-empty pair from five tiles on), A B A B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests -m gpu -x -q -k "f16 or fp16 or abi" 2>&1 | tail -4
for i in 1 2; do for v in 1 0; do
  echo "--- EMO_F16_W8_REST=$v run $i"
  EMO_F16_W8_REST=$v timeout 400 python tools/bench_conv.py 16 --quick --f16 2>&1 | F | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        if d.get('k')==3 and 'f16_cfg3_tflops' in d: print(json.dumps(dict(rest=$v,cin=d['cin'],cout=d['cout'],dims=d['dims'],ups=d['ups'],tflops=d['f16_cfg3_tflops'])))" | tee -a gpurun_out/r6_c28_conv_f16_rest_ab.jsonl
  EMO_F16_W8_REST=$v timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee -a gpurun_out/r6_c28_driver_f16_rest_$v.jsonl | cut -c1-250
done; done
