"""Micro-benchmark of the HIP 3-D grid_sample variants (in-process A/B, HIP events on the launch stream).
Prints one JSON line per (case, variant).  Algorithmic bytes per SURVEY.md section 8(d):
  read volume C*D*H*W*4 (once when shared by the batch) + read grid Do*Ho*Wo*12 (0 for analytic) + write C*Do*Ho*Wo*4."""
import json
import sys
import os

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
        yaw = (torch.rand(N, 3, generator=g) * 0.6 - 0.3)
        theta = ops.pose_theta((0.9 + 0.2 * torch.rand(N, 3, generator=g)).to(DEV), yaw.to(DEV),
                               (torch.rand(N, 3, generator=g) * 0.1 - 0.05).to(DEV))[:, :3].contiguous()
        lat = torch.cat([ident().view(1, -1, 3), torch.ones(1, D * S * S, 1)], -1).to(DEV)
        rot_warp = lat.expand(N, -1, -1).bmm(theta.transpose(1, 2)).view(N, D, S, S, 3).contiguous()
        volN = vol.expand(N, -1, -1, -1, -1).contiguous() if N <= 8 else None
        out_nc = torch.empty(N, C, D, S, S, device=DEV)
        out_cl = torch.empty(N, D, S, S, C, device=DEV)
        vol_bytes, grid_bytes = C * D * S * S * 4, D * S * S * 12
        cases = []
        for cpb in (4, 8, 12, 16, 24, 32, 48, 96):
            cases.append((f"uv/ncdhw/cpb{cpb}", lambda cpb=cpb: ops.grid_sample3d(vol, warp, variant=cpb, out=out_nc), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("uv/cl", lambda: ops.grid_sample3d(vcl, warp, in_layout="ndhwc", out_layout="ndhwc", out=out_cl), vol_bytes / N + grid_bytes + vol_bytes))
        for var in (9, 10, 11):
            cases.append((f"uv/cl_var{var}", lambda var=var: ops.grid_sample3d(vcl, warp, in_layout="ndhwc", out_layout="ndhwc", out=out_cl, variant=var), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("uv/cl_v1", lambda: ops.grid_sample3d(vcl, warp, in_layout="ndhwc", out_layout="ndhwc", out=out_cl, variant=1), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("uv/cl2ncdhw", lambda: ops.grid_sample3d(vcl, warp, in_layout="ndhwc", out_layout="ncdhw", out=out_nc), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("rot_explicit/ncdhw/cpb8", lambda: ops.grid_sample3d(vol, rot_warp, variant=8, out=out_nc), vol_bytes / N + grid_bytes + vol_bytes))
        for cpb in (8, 16, 32):
            cases.append((f"rot_theta/ncdhw/cpb{cpb}", lambda cpb=cpb: ops.grid_sample3d(vol, theta=theta, variant=cpb, out=out_nc), vol_bytes / N + vol_bytes))
        cases.append(("rot_theta/cl", lambda: ops.grid_sample3d(vcl, theta=theta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl), vol_bytes / N + vol_bytes))
        cases.append(("rot_theta/cl2ncdhw", lambda: ops.grid_sample3d(vcl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc), vol_bytes / N + vol_bytes))
        # per-sample (unshared) input volumes: the 2nd sampler of the driver pass reads a different volume per frame
        inN_cl = torch.randn(N, D, S, S, C, device=DEV)
        inN_nc = torch.randn(N, C, D, S, S, device=DEV)
        cases.append(("rot_theta_unshared/ncdhw/cpb8", lambda: ops.grid_sample3d(inN_nc, theta=theta, variant=8, out=out_nc), 2 * vol_bytes))
        for var in (2, 10):
            cases.append((f"rot_theta_unshared/cl2ncdhw_var{var}", lambda var=var: ops.grid_sample3d(inN_cl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc, variant=var), 2 * vol_bytes))
        cases.append(("rot_theta_unshared/cl2ncdhw_v1", lambda: ops.grid_sample3d(inN_cl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc, variant=1), 2 * vol_bytes))
        cases.append(("rot_theta_unshared/cl2ncdhw", lambda: ops.grid_sample3d(inN_cl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc), 2 * vol_bytes))
        cases.append(("uv_unshared/cl", lambda: ops.grid_sample3d(inN_cl, warp, in_layout="ndhwc", out_layout="ndhwc", out=out_cl), 2 * vol_bytes + grid_bytes))
        # channel-group-per-XCD layout (EMO_LAYOUT_CG8) and the driver-pass PAIR of calls (uv warp of the shared canonical volume,
        # then the rotation of the per-sample result), whole batch per launch or in chunks small enough for the intermediate
        # to stay in the 256 MiB Infinity Cache
        vcg = ops.volume_to_cg8(vol)
        inN_cg = torch.randn(N, 8, D, S, S, C // 8, device=DEV)
        out_cg = torch.empty(N, 8, D, S, S, C // 8, device=DEV)
        delta = (warp - ident().to(DEV)).permute(0, 4, 1, 2, 3).contiguous()
        cases.append(("uv/cg8", lambda: ops.grid_sample3d(vcg, warp, in_layout="cg8", out_layout="cg8", out=out_cg), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("uv_delta/cg8", lambda: ops.grid_sample3d(vcg, delta=delta, in_layout="cg8", out_layout="cg8", out=out_cg), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("uv_delta/cl_brick", lambda: ops.grid_sample3d(vcl, delta=delta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl, variant=12), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("uv_unshared_delta/cl_brick", lambda: ops.grid_sample3d(inN_cl, delta=delta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl, variant=12), 2 * vol_bytes + grid_bytes))
        cases.append(("uv_delta/cl", lambda: ops.grid_sample3d(vcl, delta=delta, in_layout="ndhwc", out_layout="ndhwc", out=out_cl), vol_bytes / N + grid_bytes + vol_bytes))
        cases.append(("rot_theta_unshared/cg8_to_ncdhw", lambda: ops.grid_sample3d(inN_cg, theta=theta, in_layout="cg8", out_layout="ncdhw", out=out_nc), 2 * vol_bytes))
        pair_bytes = vol_bytes / N + grid_bytes + vol_bytes + 2 * vol_bytes

        def pair(lay, vshared, mid, chunk, uv_variant=0):
            def run():
                for a in range(0, N, chunk):
                    b = min(N, a + chunk)
                    ops.grid_sample3d(vshared, delta=delta[a:b], in_layout=lay, out_layout=lay, out=mid[a:b], variant=uv_variant)
                    ops.grid_sample3d(mid[a:b], theta=theta[a:b], in_layout=lay, out_layout="ncdhw", out=out_nc[a:b])
            return run
        for chunk in sorted({N, min(N, 8), min(N, 4), min(N, 2)}, reverse=True):
            cases.append((f"pair/cl/chunk{chunk}", pair("ndhwc", vcl, out_cl, chunk), pair_bytes))
            cases.append((f"pair/cg8/chunk{chunk}", pair("cg8", vcg, out_cg, chunk), pair_bytes))
            cases.append((f"pair/cl_brick_uv/chunk{chunk}", pair("ndhwc", vcl, out_cl, chunk, 12), pair_bytes))
        cases.append(("torch/F.grid_sample(uv, expanded vol)", lambda: torch.nn.functional.grid_sample(inN_nc, warp, align_corners=False), 2 * vol_bytes + grid_bytes))
        cases.append(("copy/out_nc.copy_(inN_nc)", lambda: out_nc.copy_(inN_nc), 2 * vol_bytes))
        for name, fn, bytes_per_sample in cases:
            med, best = timeit(fn)
            print(json.dumps(dict(N=N, case=name, ms_med=round(med, 4), ms_min=round(best, 4),
                                  us_per_sample=round(med * 1e3 / N, 2),
                                  alg_GBps=round(bytes_per_sample * N / (med * 1e-3) / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
