#!/bin/bash
# stage-2 refinement in the fp16-operand mode ALONE under the kernel trace and the MFMA / HBM counters (the round's earlier
# trace ran the fp32 passes of tools/bench_stage2.py in the same process: its percentages mixed the two modes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=r6
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/bench_stage2.py 8 f16"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f16_kt -o s2 -- $CMD > $R/gpurun_out/${TAG}_c13_f16_prof_kt.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_f16_mfma -o s2 -- $CMD > $R/gpurun_out/${TAG}_c13_f16_prof_mfma.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_f16_fetch -o s2 -- $CMD > $R/gpurun_out/${TAG}_c13_f16_prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_f16_write -o s2 -- $CMD > $R/gpurun_out/${TAG}_c13_f16_prof_write.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_f16_kt gpurun_out/${TAG}_c13_f16_stage2_kernel_stats.csv
for p in mfma fetch write; do python tools/summarize_rocprof.py pmc gpurun_out/prof_f16_$p gpurun_out/${TAG}_c13_f16_stage2_pmc_$p.json; done
rm -rf gpurun_out/prof_f16_kt gpurun_out/prof_f16_mfma gpurun_out/prof_f16_fetch gpurun_out/prof_f16_write
timeout 200 python tools/bench_stage2.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_c13_stage2.jsonl
head -30 gpurun_out/${TAG}_c13_f16_stage2_kernel_stats.csv
cat gpurun_out/${TAG}_c13_stage2.jsonl
# one frame per step (latency form): where the 4.3 ms go
timeout 200 python tools/bench_driver.py 512 1 1 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_c13_driver_b1.jsonl
timeout 200 python tools/bench_driver.py 512 4 4 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_c13_driver_b1.jsonl
cat gpurun_out/${TAG}_c13_driver_b1.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1_kt -o b1 -- python $R/bench.py --steps 20 --warmup 3 --batch 1 --no-cpu-baseline --no-source-pass --no-extras --no-sustained --no-graph > $R/gpurun_out/${TAG}_c13_b1_prof.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_b1_kt gpurun_out/${TAG}_c13_b1_kernel_stats.csv
rm -rf gpurun_out/prof_b1_kt
head -24 gpurun_out/${TAG}_c13_b1_kernel_stats.csv
tail -3 gpurun_out/${TAG}_c13_b1_prof.log
