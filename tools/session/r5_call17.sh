#!/bin/bash
# round 5, call 17: the image head as a stream (csrc/conv_head.hip): its tests, the networks' and the bench configuration's parity
# tests, A/B against the fp32 MFMA kernel on one box, kernel trace of the bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py tests/test_infer_gpu.py -m gpu -x -q  > gpurun_out/r5_call17_pytest.log 2>&1
tail -5 gpurun_out/r5_call17_pytest.log
for f in 1 0; do
  EMO_CONV_HEAD=$f timeout 600 python bench.py --no-extras --no-cpu-baseline --no-source-pass > gpurun_out/r5_call17_bench_head$f.json 2>> gpurun_out/r5_call17.err
done
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof17 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass --no-extras --no-graph > $R/gpurun_out/r5_call17_kt.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof17 gpurun_out/r5_call17_kernel_stats.csv; rm -rf gpurun_out/prof17
python - <<'PY'
import json
for f in ("r5_call17_bench_head1.json","r5_call17_bench_head0.json"):
    d=json.loads([l for l in open("gpurun_out/"+f) if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["achieved"],v["share_of_step"]) for k,v in d["roofline_other_convs"].items()})
PY
grep -i "conv_head\|1, 1, 16, 1, 1, 128" gpurun_out/r5_call17_kernel_stats.csv | cut -c1-200
tail -3 gpurun_out/r5_call17.err
