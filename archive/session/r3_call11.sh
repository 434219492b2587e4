#!/bin/bash
# ablation of the shared-volume kernel: where do its 10-12 us per frame go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for ab in 0 1 2 4 8 3 6 7 14 15; do
EMO_GS3D_SHARED_ABLATE=$ab python - <<PY 2>&1 | grep -v amdgpu
import json, os, sys, torch
sys.path.insert(0, ".")
from emoportraits_amd import ops
C, D, S, N = 96, 16, 64, 16
g = torch.Generator().manual_seed(1)
vcl = ops.volume_to_channels_last(torch.randn(1, C, D, S, S, generator=g).cuda())
delta = (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.01).cuda()
out = torch.empty(N, D, S, S, C, device="cuda")
for chunk in (4, 16):
    def run():
        for a in range(0, N, chunk):
            ops.grid_sample3d(vcl, delta=delta[a:a + chunk], in_layout="ndhwc", out_layout="ndhwc", out=out[a:a + chunk])
    for _ in range(3): run()
    torch.cuda.synchronize()
    a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(10): run()
    b0.record(); torch.cuda.synchronize()
    print(json.dumps(dict(ablate=int(os.environ["EMO_GS3D_SHARED_ABLATE"]), chunk=chunk, us_per_frame=round(a0.elapsed_time(b0) / 10 / N * 1e3, 2))))
PY
done > gpurun_out/r3c11_ablate.jsonl
cat gpurun_out/r3c11_ablate.jsonl
