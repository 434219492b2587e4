#!/bin/bash
# ablation of the PRODUCT sampler kernels (channels-last rows): where do the uv call's 12.5 us and the rotation call's 12.6-15 us go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
for v in 0 1 2 3 4 5; do
lib=""; [ $v != 0 ] && lib=emoportraits_amd/lib/libemoportraits_hip_abl$v.so
EMO_HIP_LIB=$lib python - <<PY 2>&1 | grep -v amdgpu
import json, os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
from emoportraits_amd import ops
import restate as O
C, D, S, N = 96, 16, 64, 16
g = torch.Generator().manual_seed(1)
vcl = ops.volume_to_channels_last(torch.randn(1, C, D, S, S, generator=g).cuda())
theta = O.get_transform_matrix(1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g), 0.05 * torch.randn(N, 3, generator=g))[:, :3].contiguous().cuda()
delta = (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.03).cuda()
mid = torch.randn(N, D, S, S, C, device="cuda"); out = torch.empty(N, C, D, S, S, device="cuda")
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / it / N * 1e3, 2)
rec = dict(ablate=$v)
for chunk in (4, 16):
    def uv():
        for a in range(0, N, chunk): ops.grid_sample3d(vcl, delta=delta[a:a + chunk], in_layout="ndhwc", out_layout="ndhwc", out=mid[a:a + chunk])
    def rot():
        for a in range(0, N, chunk): ops.grid_sample3d(mid[a:a + chunk], theta=theta[a:a + chunk], in_layout="ndhwc", out_layout="ncdhw", out=out[a:a + chunk])
    def pair():
        for a in range(0, N, chunk):
            ops.grid_sample3d(vcl, delta=delta[a:a + chunk], in_layout="ndhwc", out_layout="ndhwc", out=mid[a:a + chunk])
            ops.grid_sample3d(mid[a:a + chunk], theta=theta[a:a + chunk], in_layout="ndhwc", out_layout="ncdhw", out=out[a:a + chunk])
    rec[f"uv_chunk{chunk}"] = t(uv); rec[f"rot_chunk{chunk}"] = t(rot); rec[f"pair_chunk{chunk}"] = t(pair)
print(json.dumps(rec))
PY
done > gpurun_out/r3c19_ablation.jsonl
cat gpurun_out/r3c19_ablation.jsonl
