#!/bin/bash
# round 6, call 8: the evidence set on the final kernel sources -- counter passes and kernel trace of the bench command FIRST (so
# that the bench line quotes HBM traffic measured on these sources: fp16 split, fp32 MFMA, pointwise and head kernels), then the
# whole bench half of tools/gpu_validate.sh (bench lines, microbenches, breakdowns, stage 2, sampler, pipeline, fp16 mode)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
bash tools/profile_bench.sh r6
python tools/collect_profiles.py r6 > gpurun_out/r6_collect_profiles.log 2>&1; tail -3 gpurun_out/r6_collect_profiles.log
ls profiles/r6_pmc_* 
bash tools/gpu_validate.sh r6 bench
EMO_DIST_BACKEND=gloo EMO_FORCE_DEVICE=0 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline --no-extras --no-sustained --strong-frames 77 > gpurun_out/r6_bench_8ranks_weak.json 2>> gpurun_out/r6_bench.err
python - <<'PY'
import json
for f in ("r6_bench","r6_bench_bf16x3","r6_bench_2ranks_1gpu","r6_bench_8ranks_weak"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "sust", r.get("frac_of_sustained"), "traffic", r.get("traffic"), "aff", str(d.get("host_affinity"))[:200])
        if f=="r6_bench":
            for k,v in d["roofline_other_convs"].items(): print("   other", k, v.get("achieved"), v.get("frac"), v.get("traffic"), v.get("share_of_step"))
    except Exception as e:
        print(f, "FAILED", e)
PY
