#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box): kernel-trace stats of the driver-pass launches, and the
# HBM / MFMA counters in separate --pmc passes (never combined with tracing).  Summaries land in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r1}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-source-pass --no-extras --no-sustained --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- $CMD > $R/gpurun_out/${TAG}_prof_kt.log 2>&1
CMD1="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-source-pass --no-extras --no-sustained --no-graph"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o bench -- $CMD1 > $R/gpurun_out/${TAG}_prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o bench -- $CMD1 > $R/gpurun_out/${TAG}_prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_mfma -o bench -- $CMD1 > $R/gpurun_out/${TAG}_prof_mfma.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_kt gpurun_out/${TAG}_kernel_stats.csv
for p in fetch write mfma; do python tools/summarize_rocprof.py pmc gpurun_out/prof_$p gpurun_out/${TAG}_pmc_$p.json; done
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
