#!/bin/bash
# why does the metered (eager) pass of the full default run see 2.8 ms per fp16-split launch pair when the quick run sees 1.7?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python bench.py --no-extras --no-cpu-baseline 2> gpurun_out/r4_c21.err > gpurun_out/r4_c21_bench_with_source.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4_c21_bench_with_source.json").read().strip().splitlines()[-1])
print(r["value"], r["roofline"]["avg_launch_ms"], r["roofline"]["achieved"], r["config"]["metered_pass"][-60:], r["config"]["f16x2_layers_recomputed_after_range_check"], r["source_pass_ms"])
PY
