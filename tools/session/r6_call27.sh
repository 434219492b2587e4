#!/bin/bash
# round 6, call 27: kernel trace of the fp16-operand driver pass (stage 1, 16 frames) on its own
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f16drv -o d -- python $R/tools/bench_driver.py 512 16 --f16 > $R/gpurun_out/r6_c27_prof.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_f16drv gpurun_out/r6_c27_f16_driver_kernel_stats.csv
rm -rf gpurun_out/prof_f16drv
head -30 gpurun_out/r6_c27_f16_driver_kernel_stats.csv | cut -c1-170
grep '^{' gpurun_out/r6_c27_prof.log | cut -c1-250
