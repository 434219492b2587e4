#!/bin/bash
# round 6, call 21: plain-fp16 mode, the raw patch loads of stage cg + 2 issued in half-stage 0 (behind a conversion packed into its
# first four steps) instead of half-stage 1: product build against -DEMO_W8_EARLY_LOADS=0, A B A B on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests -m gpu -x -q -k "f16w8 or F16W8 or fp16 or f16" 2>&1 | tail -4
for i in 1 2; do
  for v in product late; do
    if [ $v = product ]; then unset EMO_HIP_LIB; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; fi
    timeout 400 python tools/bench_conv.py 16 --quick --f16 2>&1 | F > gpurun_out/r6_c21_conv_f16_${v}_$i.jsonl
    echo "--- $v run $i"
    timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee -a gpurun_out/r6_c21_driver_f16_$v.jsonl | cut -c1-260
    timeout 200 python tools/bench_stage2.py 8 f16 2>&1 | F | tee -a gpurun_out/r6_c21_stage2_f16_$v.jsonl
  done
done
unset EMO_HIP_LIB
python - <<'PY'
import json
rows = {}
for v in ("product", "late"):
    for i in (1, 2):
        for l in open(f"gpurun_out/r6_c21_conv_f16_{v}_{i}.jsonl"):
            if not l.startswith("{"): continue
            d = json.loads(l)
            if "f16_cfg3_tflops" not in d or d.get("k") != 3: continue
            key = (d["cin"], d["cout"], str(d["dims"]), d["ups"])
            rows.setdefault(key, {}).setdefault(v, []).append(d["f16_cfg3_tflops"])
for k, r in rows.items():
    print(json.dumps(dict(cin=k[0], cout=k[1], dims=k[2], ups=k[3], early_loads=r.get("product"), loads_in_half_stage_1=r.get("late"))))
PY
