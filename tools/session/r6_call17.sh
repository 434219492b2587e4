#!/bin/bash
# round 6, call 17: what a bare MFMA stream sustains by instruction shape and by operand data (tools/microbench/mfma_power.hip)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_power.hip -o /tmp/mfma_power 2>/dev/null
timeout 300 /tmp/mfma_power | tee gpurun_out/r6_c17_mfma_power.jsonl
timeout 300 /tmp/mfma_power | tee -a gpurun_out/r6_c17_mfma_power.jsonl
