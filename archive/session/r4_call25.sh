#!/bin/bash
# A/B on one box: next-item prefetch on / off x persistent blocks started in four groups 16 k cycles apart (de-phasing) on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for V in "" _pf0 _pf0_stag _pf1_stag; do
  export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip$V.so
  echo "== variant '$V'"
  timeout 200 python tools/bench_conv.py 16 --quick --bf16x3-only --f16x2 > gpurun_out/r4_c25_convbench$V.jsonl 2> gpurun_out/r4_c25_convbench$V.err
  python - <<PY
import json
print("   f16x2:", " ".join(str(json.loads(l).get("f16x2_tflops")) for l in open("gpurun_out/r4_c25_convbench$V.jsonl")))
PY
  timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c25_bench$V.err > gpurun_out/r4_c25_bench$V.json
  python - <<PY
import json
r = json.loads(open("gpurun_out/r4_c25_bench$V.json").read().strip().splitlines()[-1])
print("   bench", r["value"], "roofline", r["roofline"]["achieved"], r["roofline"]["avg_launch_ms"])
PY
done
export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_pf1_stag_t.so
timeout 200 python tools/conv_phase_timing.py 16 --real > gpurun_out/r4_c25_phase_pf1_stag.jsonl 2> gpurun_out/r4_c25_phase.err
python - <<'PY'
import json
for l in open("gpurun_out/r4_c25_phase_pf1_stag.jsonl"):
    r = json.loads(l)
    if r["mode"] == "f16x2":
        print(r["cin"], r["cout"], r["dims"], r["ups"], r["mode"], "ms", r["ms"], "TF", r["tflops"], "pro", r["prologue"]["med"], "k", r["kloop"]["med"], "epi", r["epilogue_issue"]["med"], "gap", r["gap_to_next_block"]["med"],
              "| res", r["epi_res_issue"]["med"], "h0", r["epi_half0"]["med"], "h1", r["epi_half1"]["med"], "tail", r["epi_tail"]["med"])
PY
