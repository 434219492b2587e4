#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
rocprofv3 -L > gpurun_out/r1_rocprof_counters.txt 2>&1
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_one$i -o one -- python $R/tools/one_conv.py > $R/gpurun_out/r1_call8_prof$i.log 2>&1
  python $R/tools/summarize_rocprof.py pmc $R/gpurun_out/prof_one$i $R/gpurun_out/r1_call8_pmc_one$i.json
  rm -rf $R/gpurun_out/prof_one$i
done
cd $R; grep -c . gpurun_out/r1_rocprof_counters.txt; tail -3 gpurun_out/r1_call8_prof1.log
