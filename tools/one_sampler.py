"""launch the two sampler calls of the driver pass a few times (PMC target):  python tools/one_sampler.py [N] [delta_scale] [ndhwc|p4] [uv tuning] [rot tuning]
   uv call: shared channels-last canonical volume + planar deltas (identity + delta_scale * tanh(randn)); rotation call:
   per-sample channels-last volumes + analytic theta (SURVEY.md section 8d config 2 distributions)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dscale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
layout = sys.argv[3] if len(sys.argv) > 3 else "ndhwc"     # ndhwc (the driver pass's kernels) | p4 (LDS-staged tile kernels)
uv_variant = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0      # tuning words of the tile kernels (ops.tile_variant)
rot_variant = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
DEV = "cuda:0"
C, D, S = 96, 16, 64
g = torch.Generator().manual_seed(1)
vol = torch.randn(1, C, D, S, S, generator=g).to(DEV)
vcl = {"p4": ops.volume_to_p4}.get(layout, ops.volume_to_channels_last)(vol)
delta = (dscale * torch.tanh(torch.randn(N, 3, D, S, S, generator=g))).to(DEV)
theta = ops.pose_theta((0.9 + 0.2 * torch.rand(N, 3, generator=g)).to(DEV), (torch.rand(N, 3, generator=g) * 0.6 - 0.3).to(DEV),
                       (torch.rand(N, 3, generator=g) * 0.1 - 0.05).to(DEV))
warped = torch.empty({"p4": (N, C // 4, D, S, S, 4)}.get(layout, (N, D, S, S, C)), device=DEV)
out = torch.empty(N, C, D, S, S, device=DEV)
for _ in range(4):
    ops.grid_sample3d(vcl, delta=delta, in_layout=layout, out_layout=layout, out=warped, variant=uv_variant)
    ops.grid_sample3d(warped, theta=theta, in_layout=layout, out_layout="ncdhw", out=out, variant=rot_variant)
torch.cuda.synchronize()
