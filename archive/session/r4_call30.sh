#!/bin/bash
# wait states in front of every asm vector-memory instruction that reads a scalar register (VALU-written SGPR -> VMEM hazard):
# the conv parity tests on the default, the unchained and the stamped build; then phases / microbenchmark / bench, chained vs not
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for V in "" _chain0 _timing; do
  echo "== tests on '$V'"
  EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$V.so timeout 400 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_kernels_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | grep -E 'passed|failed|FAILED|fault|rror' | head -8 | cut -c1-200
done
timeout 600 python -m pytest tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py tests/test_stage2_gpu.py tests/test_two_ranks_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_timing.so timeout 200 python tools/conv_phase_timing.py 16 --real > gpurun_out/r4_c30_phase.jsonl 2> gpurun_out/r4_c30_phase.err
tail -3 gpurun_out/r4_c30_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c30_phase.jsonl"):
    r = json.loads(l)
    print(r["cin"], r["cout"], r["dims"], r["ups"], r["mode"], "ms", r["ms"], "TF", r["tflops"], "pro", r["prologue"]["med"], "k", r["kloop"]["med"], "epi", r["epilogue_issue"]["med"], "gap", r["gap_to_next_block"]["med"],
          "| res", r["epi_res_issue"]["med"], "h0", r["epi_half0"]["med"], "h1", r["epi_half1"]["med"], "tail", r["epi_tail"]["med"])
PY
for V in "" _chain0; do
  export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$V.so
  echo "== perf '$V'"
  timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c30_convbench$V.jsonl 2> gpurun_out/r4_c30_convbench$V.err
  python - <<PY
import json
rows = [json.loads(l) for l in open("gpurun_out/r4_c30_convbench$V.jsonl")]
print("   f16x2 :", " ".join(str(r.get("f16x2_tflops")) for r in rows))
print("   bf16x3:", " ".join(str(r.get("bf16x3_tflops")) for r in rows))
PY
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c30_bench$V.err | tee gpurun_out/r4_c30_bench$V.json | cut -c1-160
done
unset EMO_HIP_LIB
EMO_CONV_PRECISION=bf16x3 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c30_bench_bf16x3.err | tee gpurun_out/r4_c30_bench_bf16x3.json | cut -c1-160
