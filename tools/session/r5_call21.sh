#!/bin/bash
# round 5, call 21: the fp16 split's operand conversion on pairs (one v_cvt_pk_f16_f32 per plane and two values, no separate
# v_cvt_f16_f32): the conv parity tests (same bits expected), A/B against a build with the scalar form on one box (A B A B)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -x -q > gpurun_out/r5_call21_pytest.log 2>&1
tail -3 gpurun_out/r5_call21_pytest.log
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-source-pass > gpurun_out/r5_call21_pair_$i.json 2>> gpurun_out/r5_call21.err
  EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_scalarcvt.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-source-pass > gpurun_out/r5_call21_scalar_$i.json 2>> gpurun_out/r5_call21.err
done
python - <<'PY'
import json
for f in ("pair_1","scalar_1","pair_2","scalar_2"):
    d=json.loads([l for l in open(f"gpurun_out/r5_call21_{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], {k:v["avg_launch_ms"] for k,v in d["roofline_other_convs"].items()})
PY
