"""Micro-benchmark of the HIP 3-D grid_sample kernels (in-process A/B, HIP events on the launch stream).
Prints one JSON line per (case, variant).  Algorithmic bytes per SURVEY.md section 8(d):
  read volume C*D*H*W*4 (once when shared by the batch) + read grid Do*Ho*Wo*12 (0 for analytic) + write C*Do*Ho*Wo*4.
The LDS-staged tile kernels have their own sweeps: tools/bench_sampler_tile.py, tools/bench_sampler_pair.py."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import ops  # noqa: E402

DEV = "cuda:0"
C, D, S = 96, 16, 64


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def ident():
    gs, gz = torch.linspace(-1, 1, S), torch.linspace(-1, 1, D)
    w, v, u = torch.meshgrid(gz, gs, gs, indexing="ij")
    return torch.stack([u, v, w], -1)[None]


def main():
    Ns = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
    g = torch.Generator().manual_seed(1)
    vol = torch.randn(1, C, D, S, S, generator=g).to(DEV)
    vcl = ops.volume_to_channels_last(vol)
    for N in Ns:
        warp = (ident() + 0.05 * torch.tanh(torch.randn(N, D, S, S, 3, generator=g))).to(DEV)
        theta = ops.pose_theta((0.9 + 0.2 * torch.rand(N, 3, generator=g)).to(DEV), (torch.rand(N, 3, generator=g) * 0.6 - 0.3).to(DEV),
                               (torch.rand(N, 3, generator=g) * 0.1 - 0.05).to(DEV))[:, :3].contiguous()
        delta = (warp - ident().to(DEV)).permute(0, 4, 1, 2, 3).contiguous()
        out_nc = torch.empty(N, C, D, S, S, device=DEV)
        out_cl = torch.empty(N, D, S, S, C, device=DEV)
        inN_cl = torch.randn(N, D, S, S, C, device=DEV)
        inN_nc = torch.randn(N, C, D, S, S, device=DEV)
        vol_bytes, grid_bytes = C * D * S * S * 4, D * S * S * 12
        uv_b, rot_b = vol_bytes / N + grid_bytes + vol_bytes, 2 * vol_bytes
        cases = [
            ("uv_delta/cl rows (default)", lambda: ops.grid_sample3d(vcl, delta=delta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl), uv_b),
            ("uv_delta/cl bricks", lambda: ops.grid_sample3d(vcl, delta=delta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl, variant=1), uv_b),
            ("uv_grid/cl rows", lambda: ops.grid_sample3d(vcl, warp, in_layout="ndhwc", out_layout="ndhwc", out=out_cl), uv_b),
            ("rot_theta_unshared/cl2ncdhw", lambda: ops.grid_sample3d(inN_cl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc), rot_b),
            ("rot_theta_unshared/cl2ncdhw non-temporal out", lambda: ops.grid_sample3d(inN_cl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc, variant=2), rot_b),
            ("rot_theta_unshared/cl bricks", lambda: ops.grid_sample3d(inN_cl, theta=theta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl, variant=1), rot_b),
            # the reference's call shape: NCDHW in, explicit grid, NCDHW out
            ("seam ncdhw->ncdhw/direct gather cpb8", lambda: ops.grid_sample3d(inN_nc, warp, variant=8, out=out_nc), rot_b + grid_bytes),
            ("seam ncdhw->ncdhw/default (repack + channels-last gather)", lambda: ops.grid_sample3d(inN_nc, warp, out=out_nc), rot_b + grid_bytes),
            ("seam ncdhw->ncdhw/LDS-staged planar tiles", lambda: ops.grid_sample3d(inN_nc, warp, variant=ops.TILE, out=out_nc), rot_b + grid_bytes),
            ("seam repack alone (NCDHW -> NDHWC)", lambda: ops.volume_to_channels_last(inN_nc), rot_b),
            ("torch/F.grid_sample(uv, expanded vol)", lambda: torch.nn.functional.grid_sample(inN_nc, warp, align_corners=False), rot_b + grid_bytes),
            ("copy/out_nc.copy_(inN_nc)", lambda: out_nc.copy_(inN_nc), rot_b),
        ]

        def pair(chunk, uv_variant=0, rot_variant=0):
            def run():
                for a in range(0, N, chunk):
                    b = min(N, a + chunk)
                    ops.grid_sample3d(vcl, delta=delta[a:b], in_layout="ndhwc", out_layout="ndhwc", out=out_cl[a:b], variant=uv_variant)
                    ops.grid_sample3d(out_cl[a:b], theta=theta[a:b], in_layout="ndhwc", out_layout="ncdhw", out=out_nc[a:b], variant=rot_variant)
            return run
        for chunk in sorted({N, min(N, 8), min(N, 4), min(N, 2)}, reverse=True):
            cases.append((f"pair/rows + rows (driver pass default)/chunk{chunk}", pair(chunk), uv_b + rot_b))
            cases.append((f"pair/rows + rows, non-temporal out/chunk{chunk}", pair(chunk, 0, 2), uv_b + rot_b))
            cases.append((f"pair/bricks + rows/chunk{chunk}", pair(chunk, 1, 0), uv_b + rot_b))
        for name, fn, bytes_per_sample in cases:
            med, best = timeit(fn)
            print(json.dumps(dict(N=N, case=name, ms_med=round(med, 4), ms_min=round(best, 4),
                                  us_per_sample=round(med * 1e3 / N, 2),
                                  alg_GBps=round(bytes_per_sample * N / (med * 1e-3) / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
