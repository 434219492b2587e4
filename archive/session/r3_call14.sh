#!/bin/bash
# launch planner with / without the blocks-per-CU quantisation term: driver pass at batch 1, 2, 3, 4, 16
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for q in 0 1 0 1; do
  EMO_PLAN_QUANTISATION=$q timeout 600 python tools/bench_driver.py 512 1 2 3 4 16 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(json.dumps(dict(quantisation=$q, B=d['B'], total_ms=d['total_ms'], fps=d['fps'], warpgen_ms=d['warpgen_ms'], decoder_ms=d['decoder_ms'])))"
done > gpurun_out/r3c14_planner.jsonl
cat gpurun_out/r3c14_planner.jsonl
