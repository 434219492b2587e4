"""In-process A/B of the LDS-staged tile sampler tunings against the direct-gather kernels, on the driver pass's call pair
at the released shape [96,16,64,64] (HIP events on the launch stream; one JSON line per case).  Algorithmic bytes per
SURVEY.md section 8(d).  Usage: python tools/bench_sampler_tile.py [N=16] [chunk=4] [--quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
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


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    quick = "--quick" in sys.argv
    N = int(args[0]) if args else 16
    chunk = int(args[1]) if len(args) > 1 else 4
    import restate as O
    g = torch.Generator().manual_seed(1)
    vol = torch.randn(1, C, D, S, S, generator=g).to(DEV)
    vol_bytes, grid_bytes = C * D * S * S * 4, D * S * S * 12
    theta_b = O.get_transform_matrix(1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g),
                                     0.05 * torch.randn(N, 3, generator=g))[:, :3].contiguous().to(DEV)       # bench.py's distribution
    yaw = torch.rand(N, 3, generator=g) * 0.6 - 0.3
    theta_s = ops.pose_theta((0.9 + 0.2 * torch.rand(N, 3, generator=g)).to(DEV), yaw.to(DEV),
                             (torch.rand(N, 3, generator=g) * 0.1 - 0.05).to(DEV))[:, :3].contiguous()
    deltas = {"d0.03": (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.03).to(DEV),     # |delta| < 1 voxel in x, y
              "d0.05": (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.05).to(DEV),
              "smooth0.1": None, "wild0.6": (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * 0.6).to(DEV)}
    # a smooth field of +-0.1 (3 voxels): low-pass noise, what a conv head produces
    sm = torch.nn.functional.interpolate(torch.randn(N, 3, 4, 8, 8, generator=g), size=(D, S, S), mode="trilinear", align_corners=True)
    deltas["smooth0.1"] = (0.1 * torch.tanh(sm)).contiguous().to(DEV)
    vcl = ops.volume_to_channels_last(vol)
    vp4 = ops.volume_to_p4(vol)
    mid_cl = torch.empty(N, D, S, S, C, device=DEV)
    mid_p4 = torch.empty(N, C // 4, D, S, S, 4, device=DEV)
    mid_nc = torch.empty(N, C, D, S, S, device=DEV)
    out_nc = torch.empty(N, C, D, S, S, device=DEV)
    in_p4 = torch.randn(N, C // 4, D, S, S, 4, device=DEV)
    in_cl = torch.randn(N, D, S, S, C, device=DEV)
    in_nc = torch.randn(N, C, D, S, S, device=DEV)
    tv = ops.tile_variant
    recs = []

    def rec(name, fn, bytes_total):
        med, best = timeit(fn, iters=10 if quick else 20)
        r = dict(N=N, case=name, ms_med=round(med, 4), ms_min=round(best, 4), us_per_sample=round(med * 1e3 / N, 2),
                 alg_GBps=round(bytes_total / (med * 1e-3) / 1e9, 1), frac_of_8TBps=round(bytes_total / (med * 1e-3) / 8e12, 3))
        print(json.dumps(r), flush=True)
        recs.append(r)

    uv_bytes = vol_bytes + N * (grid_bytes + vol_bytes)
    rot_bytes = N * 2 * vol_bytes
    # ---- rotation call alone (per-sample input), bench theta distribution and the survey's ----
    for tname, theta in (("bench", theta_b), ("survey", theta_s)):
        rec(f"rot[{tname}]/direct cl2ncdhw", lambda theta=theta: ops.grid_sample3d(in_cl, theta=theta, in_layout="ndhwc", out_layout="ncdhw", out=out_nc), rot_bytes)
        rot_tunings = [("4x8x8 upb6", tv((8, 8, 4), 6)), ("4x8x8 upb24", tv((8, 8, 4), 24)), ("4x8x8 upb12", tv((8, 8, 4), 12)),
                       ("4x8x8 upb3", tv((8, 8, 4), 3)), ("4x8x16 upb24", tv((16, 8, 4), 24)), ("4x8x16 upb6", tv((16, 8, 4), 6)),
                       ("4x16x16 t512 upb24", tv((16, 16, 4), 24, threads=512)), ("4x16x16 t512 upb6", tv((16, 16, 4), 6, threads=512)),
                       ("4x8x8 upb24 lds32", tv((8, 8, 4), 24, lds_kib=32)), ("4x8x8 upb24 lds20", tv((8, 8, 4), 24, lds_kib=20)),
                       ("2x8x16 upb24", tv((16, 8, 2), 24)), ("4x4x16 upb24", tv((16, 4, 4), 24)), ("8x8x8 upb24", tv((8, 8, 8), 24))]
        if quick:
            rot_tunings = rot_tunings[:4]
        for tag, var in rot_tunings:
            rec(f"rot[{tname}]/tile p4->ncdhw {tag}", lambda theta=theta, var=var: ops.grid_sample3d(in_p4, theta=theta, in_layout="p4", out_layout="ncdhw", out=out_nc, variant=var), rot_bytes)
        rec(f"rot[{tname}]/tile p4->p4 4x8x8 upb24", lambda theta=theta: ops.grid_sample3d(in_p4, theta=theta, in_layout="p4", out_layout="p4", out=mid_p4, variant=tv((8, 8, 4), 24)), rot_bytes)
        rec(f"rot[{tname}]/direct ncdhw", lambda theta=theta: ops.grid_sample3d(in_nc, theta=theta, out=out_nc), rot_bytes)
        for tag, var in (("default", 0), ("4x8x16 upb16", tv((16, 8, 4), 16)), ("4x4x64 upb16", tv((64, 4, 2), 16)), ("2x8x32 upb8", tv((32, 8, 2), 8)),
                         ("4x8x16 upb96", tv((16, 8, 4), 31))):
            rec(f"rot[{tname}]/tile ncdhw->ncdhw {tag}", lambda theta=theta, var=var: ops.grid_sample3d(in_nc, theta=theta, out=out_nc, variant=ops.TILE | var), rot_bytes)
    # ---- uv call alone (shared canonical volume + planar deltas) ----
    for dname, delta in deltas.items():
        rec(f"uv[{dname}]/direct cl", lambda delta=delta: ops.grid_sample3d(vcl, delta=delta, in_layout="ndhwc", out_layout="ndhwc", out=mid_cl), uv_bytes)
        uv_tunings = [("16x4x8 upb3", tv((8, 4, 16), 3)), ("16x4x8 upb6", tv((8, 4, 16), 6)), ("16x4x8 upb24", tv((8, 4, 16), 24)),
                      ("16x8x8 t512 upb3", tv((8, 8, 16), 3, threads=512)), ("16x8x8 t512 upb24", tv((8, 8, 16), 24, threads=512)),
                      ("16x4x4 upb3", tv((4, 4, 16), 3)), ("16x4x4 upb24", tv((4, 4, 16), 24)), ("4x8x8 upb24", tv((8, 8, 4), 24)),
                      ("8x8x8 upb24", tv((8, 8, 8), 24)), ("8x8x8 upb3", tv((8, 8, 8), 3)), ("16x2x16 upb24", tv((16, 2, 16), 24))]
        if quick:
            uv_tunings = uv_tunings[:4]
        for tag, var in uv_tunings:
            rec(f"uv[{dname}]/tile p4->p4 {tag}", lambda delta=delta, var=var: ops.grid_sample3d(vp4, delta=delta, in_layout="p4", out_layout="p4", out=mid_p4, variant=var), uv_bytes)
        rec(f"uv[{dname}]/direct ncdhw", lambda delta=delta: ops.grid_sample3d(vol, delta=delta, out=mid_nc), uv_bytes)
        rec(f"uv[{dname}]/tile ncdhw->ncdhw default", lambda delta=delta: ops.grid_sample3d(vol, delta=delta, out=mid_nc, variant=ops.TILE), uv_bytes)
        rec(f"uv[{dname}]/tile ncdhw->ncdhw 4x4x64", lambda delta=delta: ops.grid_sample3d(vol, delta=delta, out=mid_nc, variant=ops.TILE | tv((64, 4, 2), 16)), uv_bytes)
    # ---- the pair, chunked ----
    pair_bytes = uv_bytes + rot_bytes

    def pair(kind, delta, theta, ch, uvv=0, rotv=0):
        def run():
            for a in range(0, N, ch):
                b = min(N, a + ch)
                if kind == "cl":
                    ops.grid_sample3d(vcl, delta=delta[a:b], in_layout="ndhwc", out_layout="ndhwc", out=mid_cl[a:b])
                    ops.grid_sample3d(mid_cl[a:b], theta=theta[a:b], in_layout="ndhwc", out_layout="ncdhw", out=out_nc[a:b])
                else:
                    ops.grid_sample3d(vp4, delta=delta[a:b], in_layout="p4", out_layout="p4", out=mid_p4[a:b], variant=uvv)
                    ops.grid_sample3d(mid_p4[a:b], theta=theta[a:b], in_layout="p4", out_layout="ncdhw", out=out_nc[a:b], variant=rotv)
        return run
    for dname in ("d0.03", "smooth0.1", "wild0.6"):
        for ch in sorted({N, chunk, 2, 8} & set(range(1, N + 1)), reverse=True):
            rec(f"pair[{dname},bench theta]/direct cl chunk{ch}", pair("cl", deltas[dname], theta_b, ch), pair_bytes)
            rec(f"pair[{dname},bench theta]/tile default chunk{ch}", pair("p4", deltas[dname], theta_b, ch), pair_bytes)
            rec(f"pair[{dname},bench theta]/tile uv 16x4x8 upb24, rot 4x8x8 upb24 chunk{ch}",
                pair("p4", deltas[dname], theta_b, ch, tv((8, 4, 16), 24), tv((8, 8, 4), 24)), pair_bytes)
    rec("copy/out_nc.copy_(in_nc)", lambda: out_nc.copy_(in_nc), rot_bytes)


if __name__ == "__main__":
    main()
