#!/bin/bash
# L2 / L1 counters of the two sampler kernels (separate --pmc passes, no tracing):  bash tools/pmc_sampler.sh <tag> [N] [delta_scale] [variant]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/tools/one_sampler.py $@"
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_s$i -o x -- $CMD > $R/gpurun_out/${TAG}_pmc_s$i.log 2>&1
  (cd $R && python tools/summarize_rocprof.py pmc gpurun_out/pmc_s$i gpurun_out/${TAG}_pmc_s$i.json && rm -rf gpurun_out/pmc_s$i)
done
cd $R
python - <<PY
import json, glob
out = {}
for f in sorted(glob.glob("gpurun_out/${TAG}_pmc_s*.json")):
    for k, v in json.load(open(f)).items():
        if k.startswith("gs3d") or "tile_kernel" in k:
            out.setdefault(k[:48], {}).update({c: round(x["mean"]) for c, x in v.items()})
json.dump(out, open("gpurun_out/${TAG}_pmc_sampler.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
