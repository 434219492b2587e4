#!/bin/bash
# round 6, call 16: what the operand streams cost the (power-managed) two-tile split kernel.  Measurement builds whose results
# are WRONG on purpose: -DEMO_CT2_W_CONST=1 (every weight DMA piece re-reads the first KiB: served by the CU's L1 instead of
# the L2), -DEMO_CT2_X_CONST=1 (every stage re-reads input channel 0's patch), both; per-layer rate against the product build,
# same box, two rounds.  Same instruction stream, same LDS traffic, same MFMA work: only where the bytes come from changes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
for i in 1 2; do
  for v in product wconst xconst wxconst; do
    if [ $v = product ]; then unset EMO_HIP_LIB; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; fi
    timeout 400 python tools/bench_conv.py 16 --bf16x3-only --f16x2 --quick 2>&1 | F > gpurun_out/r6_c16_conv_${v}_$i.jsonl
  done
done
python - <<'PY'
import json
rows = {}
for v in ("product", "wconst", "xconst", "wxconst"):
    for i in (1, 2):
        for l in open(f"gpurun_out/r6_c16_conv_{v}_{i}.jsonl"):
            if not l.startswith("{"): continue
            d = json.loads(l)
            if "f16x2_tflops" not in d: continue
            key = (d["cin"], d["cout"], str(d["dims"]), d["ups"])
            rows.setdefault(key, {}).setdefault(v, []).append(d["f16x2_tflops"])
out = []
for k, r in rows.items():
    rec = dict(cin=k[0], cout=k[1], dims=k[2], ups=k[3], **{v: r.get(v) for v in ("product", "wconst", "xconst", "wxconst")})
    out.append(rec); print(json.dumps(rec))
json.dump(out, open("gpurun_out/r6_c16_operand_stream_cost.json", "w"), indent=1)
PY
# phase stamps with the clock: product against wxconst
for v in product wxconst; do
  if [ $v = product ]; then unset EMO_HIP_LIB; else export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$v.so; fi
  echo "--- bench step, $v (results of the variant are wrong: rate only)"
  timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-sustained --strong-frames 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launch_ms'))"
done
