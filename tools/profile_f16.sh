#!/bin/bash
# rocprofv3 evidence for the opt-in fp16-operand conv kernels (BASELINE configs[4]): kernel trace + MFMA / HBM counters of the
# stage-2 refinement at 512^2 (tools/bench_stage2.py) and of the stage-1 driver pass in f16 mode.  Separate --pmc passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/bench_stage2.py 8 f16"   # the fp16-operand mode alone (both norm variants)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f16_kt -o s2 -- $CMD > $R/gpurun_out/${TAG}_f16_prof_kt.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_f16_mfma -o s2 -- $CMD > $R/gpurun_out/${TAG}_f16_prof_mfma.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_f16_fetch -o s2 -- $CMD > $R/gpurun_out/${TAG}_f16_prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_f16_write -o s2 -- $CMD > $R/gpurun_out/${TAG}_f16_prof_write.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_f16_kt gpurun_out/${TAG}_f16_kernel_stats.csv
for p in mfma fetch write; do python tools/summarize_rocprof.py pmc gpurun_out/prof_f16_$p gpurun_out/${TAG}_f16_pmc_$p.json; done
rm -rf gpurun_out/prof_f16_kt gpurun_out/prof_f16_mfma gpurun_out/prof_f16_fetch gpurun_out/prof_f16_write
timeout 200 python tools/bench_conv.py 16 --quick --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_f16_conv.jsonl
timeout 200 python tools/bench_driver.py 512 16 --f16 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_f16_driver512.jsonl
# SQ issue / wait / LDS counters of two decoder-shape layers on the fp16-operand kernel, the per-block overhead fit of both conv
# kernels and the global-load instruction-rate microbenchmark that motivates the 16-byte staging loads
bash tools/pmc_conv.sh ${TAG}_f16_512c 512 512 64 64 0 f16 > /dev/null 2>&1
bash tools/pmc_conv.sh ${TAG}_f16_128c 128 128 512 512 0 f16 > /dev/null 2>&1
(timeout 100 python tools/fit_conv_overhead.py f16 512 128 4; timeout 100 python tools/fit_conv_overhead.py f32 512 128 4) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_conv_overhead_fit.jsonl
[ -x tools/microbench/vmem_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/vmem_rate.hip -o tools/microbench/vmem_rate 2>/dev/null
timeout 60 tools/microbench/vmem_rate > gpurun_out/${TAG}_vmem_rate.jsonl 2>&1
