#!/bin/bash
# round 3, call 2: counters of the first tile kernels (16 frames per launch; uv 4x8x8 upb24 + rot 4x8x8 upb24, then upb3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
V24=$(python -c "from emoportraits_amd import ops; print(ops.tile_variant((8,8,4),24))")
V3=$(python -c "from emoportraits_amd import ops; print(ops.tile_variant((8,8,4),3))")
bash tools/pmc_sampler.sh r3c2_upb24 16 0.03 p4 $V24 $V24 > gpurun_out/r3c2_upb24.log 2>&1
bash tools/pmc_sampler.sh r3c2_upb3 16 0.03 p4 $V3 $V3 > gpurun_out/r3c2_upb3.log 2>&1
tail -5 gpurun_out/r3c2_upb24.log
