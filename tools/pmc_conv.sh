#!/bin/bash
# SQ stall breakdown of ONE conv layer (two --pmc passes, no tracing):  bash tools/pmc_conv.sh <tag> [lib-variant-suffix] [B cfg cin cout hw]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; VAR=$2; shift 2
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
[ -n "$VAR" ] && export EMO_HIP_LIB=$R/emoportraits_amd/lib/libemoportraits_hip_$VAR.so
cd /tmp
CMD="python $R/tools/one_conv.py $@"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL --output-format csv -d $R/gpurun_out/pmc_a -o x -- $CMD > $R/gpurun_out/${TAG}_pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_b -o x -- $CMD > $R/gpurun_out/${TAG}_pmc_b.log 2>&1
cd $R
python tools/summarize_rocprof.py pmc gpurun_out/pmc_a gpurun_out/${TAG}_pmc_a.json
python tools/summarize_rocprof.py pmc gpurun_out/pmc_b gpurun_out/${TAG}_pmc_b.json
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
python - <<PY
import json
for p in ("a", "b"):
    d = json.load(open("gpurun_out/${TAG}_pmc_%s.json" % p))
    for k, v in d.items():
        if k.startswith("conv_igemm"):
            print("${TAG}", p, {c: round(x["sum"] / x["launches"]) for c, x in v.items()})
PY
