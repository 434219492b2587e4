#!/bin/bash
# round 5, call 11: kernel trace of the stage-2 refinement in the default conv mode (where do its 41 ms per 8 frames go?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/s2_once.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from emoportraits_amd import stage2
cfg = stage2.stage2_config(overrides=dict(output_size_s2=512))
s2 = stage2.Stage2(stage2.random_state_dict(cfg, seed=0), cfg, "cuda:0")
g = torch.Generator().manual_seed(1)
img = torch.rand(8, 3, 512, 512, generator=g).to("cuda:0")
mask = (torch.rand(8, 1, 512, 512, generator=g) > 0.1).float().to("cuda:0")
face = (torch.rand(8, 1, 512, 512, generator=g) > 0.3).float().to("cuda:0")
for _ in range(4):
    s2.refine(img, mask, face)
torch.cuda.synchronize()
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s2 -o s2 -- python /tmp/s2_once.py > $R/gpurun_out/r5_s2_prof.log 2>&1
cd $R
python tools/summarize_rocprof.py stats gpurun_out/prof_s2 gpurun_out/r5_stage2_f16x2_kernel_stats.csv
rm -rf gpurun_out/prof_s2
head -30 gpurun_out/r5_stage2_f16x2_kernel_stats.csv | cut -c1-160
