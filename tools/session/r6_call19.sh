#!/bin/bash
# round 6, call 19: the operand supply of the PLAIN-fp16 mode on the eight-wave kernel (emo_conv_igemm_f16w8): the same measurement
# builds as call 16 (weights / patches from the L1, wrong results on purpose), per-layer rate, two rounds on one box; then the
# bench's counter passes on the final kernel sources (the measurement switches moved to the shared header)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
for i in 1 2; do
  for v in product wconst xconst wxconst; do
    if [ $v = product ]; then unset EMO_HIP_LIB; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; fi
    timeout 400 python tools/bench_conv.py 16 --quick --f16 2>&1 | F > gpurun_out/r6_c19_conv_f16_${v}_$i.jsonl
  done
done
unset EMO_HIP_LIB
python - <<'PY'
import json
rows = {}
for v in ("product", "wconst", "xconst", "wxconst"):
    for i in (1, 2):
        for l in open(f"gpurun_out/r6_c19_conv_f16_{v}_{i}.jsonl"):
            if not l.startswith("{"): continue
            d = json.loads(l)
            if "f16_cfg3_tflops" not in d or d.get("k") != 3: continue
            key = (d["cin"], d["cout"], str(d["dims"]), d["ups"])
            rows.setdefault(key, {}).setdefault(v, []).append(d["f16_cfg3_tflops"])
out = []
for k, r in rows.items():
    rec = dict(cin=k[0], cout=k[1], dims=k[2], ups=k[3], **{v: r.get(v) for v in ("product", "wconst", "xconst", "wxconst")})
    out.append(rec); print(json.dumps(rec))
json.dump(out, open("gpurun_out/r6_c19_f16w8_operand_stream_cost.json", "w"), indent=1)
PY
bash tools/profile_bench.sh r6
head -4 gpurun_out/r6_kernel_stats.csv | cut -c1-140
