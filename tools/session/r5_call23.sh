#!/bin/bash
# round 5, call 23: the upsampling kernel with its column arithmetic out of the loop, the avg-pool with four outputs per thread
# (both bit for bit their generic kernels on the CPU: tests/test_stream_kernels_emul.py): their GPU tests, the networks' parity
# tests, kernel trace of the bench command, a short bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -x -q -k "upsample or avgpool or groupnorm or warp or Warp or driver or source or bench or parity" > gpurun_out/r5_call23_pytest.log 2>&1
tail -2 gpurun_out/r5_call23_pytest.log
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof23 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass --no-extras --no-graph > $R/gpurun_out/r5_call23_kt.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof23 gpurun_out/r5_call23_kernel_stats.csv; rm -rf gpurun_out/prof23
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r5_call23_bench.json 2> gpurun_out/r5_call23.err
grep -i "upsample\|avgpool" gpurun_out/r5_call23_kernel_stats.csv | cut -c1-60,150-260
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5_call23_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["source_pass_ms"])
PY
