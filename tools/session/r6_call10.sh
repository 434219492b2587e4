#!/bin/bash
# round 6, call 10: the kernel trace and counter passes of the bench command once more WITHOUT the sustained-MFMA diagnostic behind the
# timed region (call 8's trace holds its ~1 s per form: 92 % of the traced time)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
bash tools/profile_bench.sh r6
head -5 gpurun_out/r6_kernel_stats.csv | cut -c1-160
