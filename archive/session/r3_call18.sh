#!/bin/bash
# LDS-staged tile kernels: raw s_barrier (no vmcnt wait for the stores) vs __syncthreads() after the gather
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_sampler_tile_gpu.py -m gpu -x -q --timeout=600 2>&1 | grep -v "amdgpu.ids" | tail -3
for lib in "" "emoportraits_amd/lib/libemoportraits_hip_syncbar.so"; do
EMO_HIP_LIB=$lib python - <<PY 2>&1 | grep -v amdgpu
import json, os, sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
from emoportraits_amd import ops
import restate as O
C, D, S, N = 96, 16, 64, 16
g = torch.Generator().manual_seed(1)
vol = torch.randn(1, C, D, S, S, generator=g).cuda()
vp4 = ops.volume_to_p4(vol)
theta = O.get_transform_matrix(1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g), 0.05 * torch.randn(N, 3, generator=g))[:, :3].contiguous().cuda()
delta = (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.03).cuda()
in_p4 = torch.randn(N, C // 4, D, S, S, 4, device="cuda")
mid = torch.empty(N, C // 4, D, S, S, 4, device="cuda"); out = torch.empty(N, C, D, S, S, device="cuda")
tv = ops.tile_variant
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it / N * 1e3
lib = os.environ.get("EMO_HIP_LIB") or "raw barrier"
for name, var in (("4x8x8 u24", tv((8, 8, 4), 24)), ("4x8x8 u6", tv((8, 8, 4), 6)), ("4x8x8 u3", tv((8, 8, 4), 3)), ("4x8x16 u24", tv((16, 8, 4), 24)), ("4x16x16 t512 u24", tv((16, 16, 4), 24, threads=512))):
    r = t(lambda: ops.grid_sample3d(in_p4, theta=theta, in_layout="p4", out_layout="ncdhw", out=out, variant=var))
    r2 = t(lambda: ops.grid_sample3d(in_p4, theta=theta, in_layout="p4", out_layout="p4", out=mid, variant=var))
    u = t(lambda: ops.grid_sample3d(vp4, delta=delta, in_layout="p4", out_layout="p4", out=mid, variant=var))
    print(json.dumps(dict(lib=lib[-20:], tuning=name, rot_p4_to_ncdhw=round(r, 2), rot_p4_to_p4=round(r2, 2), uv_p4_to_p4=round(u, 2))))
PY
done > gpurun_out/r3c18_rawbarrier.jsonl
cat gpurun_out/r3c18_rawbarrier.jsonl
