#!/bin/bash
# round 5, call 20: the 8-rank functional runs of bench.py (all ranks on the one GPU of the box, gloo) on the final code
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
EMO_FORCE_DEVICE=0 EMO_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --strong-frames 77 > gpurun_out/r5_bench_8ranks_weak.json 2> gpurun_out/r5_bench_8ranks_weak.err
EMO_FORCE_DEVICE=0 EMO_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --total-frames 77 > gpurun_out/r5_bench_8ranks_strong.json 2> gpurun_out/r5_bench_8ranks_strong.err
python - <<'PY'
import json
for f in ("weak","strong"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r5_bench_8ranks_{f}.json") if l.startswith("{")][-1])
        print(f, d["scaling"], d["n_gpus"], d["value"], d["ms_per_step"], d.get("strong_scaling"), d.get("broadcast_ms"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -c 400 gpurun_out/r5_bench_8ranks_strong.err
