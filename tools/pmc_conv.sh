#!/bin/bash
# SQ issue / wait counters of one conv layer (separate --pmc passes, no tracing):  bash tools/pmc_conv.sh <tag> cin cout H W [ups] [f16|f32] [B]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/one_conv.py $@"
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_c$i -o x -- $CMD > $R/gpurun_out/${TAG}_pmc_c$i.log 2>&1
  (cd $R && python tools/summarize_rocprof.py pmc gpurun_out/pmc_c$i gpurun_out/${TAG}_pmc_c$i.json && rm -rf gpurun_out/pmc_c$i)
done
cd $R
python - <<PY
import json, glob
out = {}
for f in sorted(glob.glob("gpurun_out/${TAG}_pmc_c*.json")):
    for k, v in json.load(open(f)).items():
        if k.startswith("conv_igemm") or "conv_igemm" in k[:40]:
            out.setdefault(k[:90], {}).update({c: round(x["mean"]) for c, x in v.items()})
json.dump(out, open("gpurun_out/${TAG}_pmc_conv.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
