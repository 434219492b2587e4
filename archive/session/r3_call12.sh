#!/bin/bash
# kernel trace of the driver pass at batch 1 (latency mode): where do the 9 ms go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -o b1 -- python $R/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-source-pass --no-extras > $R/gpurun_out/r3c12_b1.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_b1 gpurun_out/r3c12_b1_kernel_stats.csv
cp $(find gpurun_out/prof_b1 -name "*kernel_trace.csv" | head -1) gpurun_out/r3c12_b1_kernel_trace.csv 2>/dev/null
rm -rf gpurun_out/prof_b1
tail -2 gpurun_out/r3c12_b1.log | cut -c1-300
