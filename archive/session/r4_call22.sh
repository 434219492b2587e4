#!/bin/bash
# the full default bench line, alone on the box (the validation run's copy had its metered pass disturbed: 83.9 ms per eager
# step against 55.7 when run alone), then the same in the bf16 split
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python bench.py 2> gpurun_out/r4_bench_n1.err > gpurun_out/r4_bench_n1.json
tail -c 1500 gpurun_out/r4_bench_n1.json | head -c 1500
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4_bench_n1.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"], r["roofline_sampler"]["frac"], r["cpu_baseline"]["value"])
print({k: v for k, v in r["extras"].items() if not isinstance(v, (dict, str))})
PY
EMO_CONV_PRECISION=bf16x3 timeout 600 python bench.py --no-cpu-baseline 2> gpurun_out/r4_bench_n1_bf16x3.err > gpurun_out/r4_bench_n1_bf16x3.json
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r4_bench_n1_bf16x3.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["roofline"])
PY
