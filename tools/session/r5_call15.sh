#!/bin/bash
# round 5, call 15: the bench line of the final code (metered pass sized from the host's enqueue time), default and bf16 split
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
EMO_CONV_PRECISION=bf16x3 timeout 600 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r5_bench_bf16x3.json 2>> gpurun_out/r5_bench.err
python - <<'PY'
import json
for f in ("r5_bench.json","r5_bench_bf16x3.json"):
    d=json.loads([l for l in open("gpurun_out/"+f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d["roofline"]["traffic"], d["config"]["metered_pass"][-110:])
PY
