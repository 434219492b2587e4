#!/bin/bash
# what the wrapper adds around the hot path: kernel trace of animate_frames (16 frames per batch, hipGraph replay of the driver
# pass) -- everything in it that is not in the bench step's kernel list is wrapper overhead
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pipe_kt -o pipe -- python $R/tools/bench_pipeline.py 512 16 > $R/gpurun_out/r6_c15_pipe_prof.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_pipe_kt gpurun_out/r6_c15_pipeline_kernel_stats.csv
rm -rf gpurun_out/prof_pipe_kt
head -60 gpurun_out/r6_c15_pipeline_kernel_stats.csv | cut -c1-160
grep -v amdgpu gpurun_out/r6_c15_pipe_prof.log | grep '^{' 
