#!/bin/bash
# A/B: non-temporal stores of the uv sampler output
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for lib in "" "emoportraits_amd/lib/libemoportraits_hip_ntout.so" "" "emoportraits_amd/lib/libemoportraits_hip_ntout.so"; do
  for c in 4 8 16; do EMO_HIP_LIB=$lib timeout 300 python tools/bench_sampler_pair.py 16 $c 0.03 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(json.dumps(dict(nt='$lib' != '', chunk=d['chunk'], case=d['case'][:40], pair=d['pair_us_per_frame'], uv=d['uv_us_per_frame'], rot=d['rot_us_per_frame'])))"; done
done > gpurun_out/r3c16_ntout.jsonl
cat gpurun_out/r3c16_ntout.jsonl
