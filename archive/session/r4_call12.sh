#!/bin/bash
# full GPU suite with the guarded f16x2 default; persistent blocks A/B on the layer shapes; bench default
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r4_c12_tests.txt; tail -5 gpurun_out/r4_c12_tests.txt
for pz in 0 1; do
EMO_CONV_BF16X3_PERSISTENT=$pz timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c12_convbench_p$pz.jsonl 2>> gpurun_out/r4_c12.err
python - <<PY
import json
print("persistent $pz")
for l in open("gpurun_out/r4_c12_convbench_p$pz.jsonl"):
    r = json.loads(l)
    print("  ", r["cin"], r["cout"], r["dims"], r["ups"], "bf16x3", r.get("bf16x3_tflops"), "f16x2", r.get("f16x2_tflops"))
PY
done
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c12_bench.err | tee gpurun_out/r4_c12_bench.json | cut -c1-200
EMO_CONV_BF16X3_PERSISTENT=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2>> gpurun_out/r4_c12_bench.err | tee gpurun_out/r4_c12_bench_persistent.json | cut -c1-200
EMO_F16X2_GUARD=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2>> gpurun_out/r4_c12_bench.err | tee gpurun_out/r4_c12_bench_noguard.json | cut -c1-200
