#!/bin/bash
# full default bench line with extras and cpu baseline (what the driver runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python bench.py 2> gpurun_out/r4_c16_bench.err > gpurun_out/r4_c16_bench.json; tail -3 gpurun_out/r4_c16_bench.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4_c16_bench.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["achieved"], r["roofline_sampler"])
print(json.dumps(r.get("extras"), indent=0)[:3000])
print(r.get("cpu_baseline", {}).get("value"), r.get("source_pass_ms"))
PY
