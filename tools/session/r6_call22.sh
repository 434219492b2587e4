#!/bin/bash
# round 6, call 22: plain-fp16 mode, how early the patch loads can go: the conversion of stage cg + 1 in 4 (product), 2 or 1 steps of
# half-stage 0, the loads right behind it (-DEMO_W8_EARLY_STEPS=2 / 1); A B C A B C on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
for i in 1 2; do
  for v in product es2 es1; do
    if [ $v = product ]; then unset EMO_HIP_LIB; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; fi
    timeout 400 python tools/bench_conv.py 16 --quick --f16 2>&1 | F > gpurun_out/r6_c22_conv_f16_${v}_$i.jsonl
    echo "--- $v run $i"
    timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | F | tee -a gpurun_out/r6_c22_driver_f16_$v.jsonl | cut -c1-260
    timeout 200 python tools/bench_stage2.py 8 f16 2>&1 | F | tee -a gpurun_out/r6_c22_stage2_f16_$v.jsonl
  done
done
unset EMO_HIP_LIB
python - <<'PY'
import json
rows = {}
for v in ("product", "es2", "es1"):
    for i in (1, 2):
        for l in open(f"gpurun_out/r6_c22_conv_f16_{v}_{i}.jsonl"):
            if not l.startswith("{"): continue
            d = json.loads(l)
            if "f16_cfg3_tflops" not in d or d.get("k") != 3: continue
            key = (d["cin"], d["cout"], str(d["dims"]), d["ups"])
            rows.setdefault(key, {}).setdefault(v, []).append(d["f16_cfg3_tflops"])
for k, r in rows.items():
    print(json.dumps(dict(cin=k[0], cout=k[1], dims=k[2], ups=k[3], steps4=r.get("product"), steps2=r.get("es2"), steps1=r.get("es1"))))
PY
