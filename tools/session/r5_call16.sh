#!/bin/bash
# round 5, call 16: the upsampling kernel without its one-output-at-a-time edge path, with the GroupNorm sums of its output
# (WarpGenerator); tests of the two, the networks' parity tests, A/B of the fused statistics, kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py -m gpu -x -q -k "upsample or groupnorm or warp or Warp or driver or pipeline or bench" > gpurun_out/r5_call16_pytest.log 2>&1
tail -5 gpurun_out/r5_call16_pytest.log
for f in 1 0; do
  EMO_FUSE_UPSAMPLE_STATS=$f timeout 600 python bench.py --no-extras --no-cpu-baseline --no-source-pass > gpurun_out/r5_call16_bench_fuse$f.json 2>> gpurun_out/r5_call16.err
done
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof16 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass --no-extras --no-graph > $R/gpurun_out/r5_call16_kt.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof16 gpurun_out/r5_call16_kernel_stats.csv; rm -rf gpurun_out/prof16
python - <<'PY'
import json
for f in ("r5_call16_bench_fuse1.json","r5_call16_bench_fuse0.json"):
    d=json.loads([l for l in open("gpurun_out/"+f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
grep -i "upsample\|gn_partial\|gn_finalize\|avgpool" gpurun_out/r5_call16_kernel_stats.csv | cut -c1-200
