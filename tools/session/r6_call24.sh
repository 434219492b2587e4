#!/bin/bash
# round 6, call 24: with the early patch loads, do odd channel-tile counts (320 / 192 outputs: a half-empty last pair) pay on the
# eight-wave kernel in the plain-fp16 mode?  EMO_F16_W8_ODD = 0 (planner default) / 1, A B A B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
for i in 1 2; do for v in 0 1; do
  echo "--- EMO_F16_W8_ODD=$v run $i"
  EMO_F16_W8_ODD=$v timeout 400 python tools/bench_conv.py 16 --quick --f16 2>&1 | F | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        if d.get('k')==3 and 'f16_cfg3_tflops' in d: print(d['cin'],d['cout'],d['dims'],d['ups'],d['f16_cfg3_tflops'])"
  EMO_F16_W8_ODD=$v timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | F | cut -c1-250
done; done
