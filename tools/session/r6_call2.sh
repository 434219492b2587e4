#!/bin/bash
# round 6, call 2: the two-tile kernel with two waves per SIMD (conv_igemm_f16x2_w8.h) on the GPU: bit-identity with the single-tile
# kernel (and the four-wave two-tile kernel beside it), A B A B bench lines on one box (EMO_CONV_W8=1 / 0), the layer microbench, and
# the per-item phase stamps of both kernels from a measurement build
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q -x -s -k "two_tile" 2>&1 | F > gpurun_out/r6_c2_pytest_full.log
grep -a "passed\|failed\|Error\|FAILED\|assert" gpurun_out/r6_c2_pytest_full.log | tail -8
echo "--- A B A B bench (W8 = 1, 0, 1, 0)"
for i in 1 2; do
  for w in 1 0; do
    EMO_CONV_W8=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sustained --strong-frames 0 > gpurun_out/r6_c2_bench_w8_${w}_$i.json 2> gpurun_out/r6_c2_bench.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r6_c2_bench_w8_${w}_$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("W8=$w run $i:", d["value"], "fps", d["ms_per_step"], "ms  frac", r["frac"], "launch ms", r["avg_launch_ms"], "share", r["share_of_step"])
PY
  done
done
echo "--- layer microbench (16 frames), W8 = 1 / 0"
EMO_CONV_W8=1 timeout 400 python tools/bench_conv.py 16 --bf16x3-only --f16x2 --quick 2>&1 | F > gpurun_out/r6_c2_conv_w8.jsonl
EMO_CONV_W8=0 timeout 400 python tools/bench_conv.py 16 --bf16x3-only --f16x2 --quick 2>&1 | F > gpurun_out/r6_c2_conv_ct2.jsonl
python - <<'PY'
import json
def rows(f):
    out={}
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); out[(d.get("cin"),d.get("cout"),str(d.get("dims")),d.get("ups"))]=d
    return out
a,b=rows("gpurun_out/r6_c2_conv_w8.jsonl"),rows("gpurun_out/r6_c2_conv_ct2.jsonl")
for k in a:
    if k in b:
        print(k, "w8", a[k].get("f16x2_tflops"), "ct2", b[k].get("f16x2_tflops"))
PY
echo "--- phase stamps (measurement build)"
timeout 900 python -m emoportraits_amd.build --variant timing EMO_S_TIMING=1 > gpurun_out/r6_c2_build_timing.log 2>&1; tail -1 gpurun_out/r6_c2_build_timing.log
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 600 python tools/conv_phase_timing.py 16 --real --modes ct2,w8 --shapes 0,1,4 2>&1 | F > gpurun_out/r6_c2_phase.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r6_c2_phase.jsonl"):
    if not l.startswith("{"): print(l.strip()[:300]); continue
    d=json.loads(l)
    print(d["mode"], d["cin"], d["cout"], d["dims"], "ms", d["ms"], "TF", d["tflops"], "pro", d["prologue"]["med"], "kloop", d["kloop"]["med"], "epi", d["epilogue_issue"]["med"], "gap", d["gap_to_next_block"]["med"], "clk", d.get("eff_clock_ghz"))
PY
echo "--- the rest of call 1's list (stopped at its first failure) with the new kernel as the default"
timeout 1500 python -m pytest tests/test_two_ranks_gpu.py tests/test_bench_config_parity_gpu.py tests/test_nets_gpu.py -m gpu -q -s 2>&1 | F > gpurun_out/r6_c2_pytest2_full.log
grep -a "PARITY\|passed\|failed\|Error\|FAILED" gpurun_out/r6_c2_pytest2_full.log > gpurun_out/r6_c2_pytest2.log; tail -4 gpurun_out/r6_c2_pytest2.log
