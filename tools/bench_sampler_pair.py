"""The driver pass's sampler pair as the hot path runs it -- chunks of `chunk` frames, uv call (shared canonical volume,
planar deltas) then rotation call (analytic theta, NCDHW out) on the chunk's intermediate -- for a list of tile tunings,
against the direct-gather pair.  HIP events on the launch stream around the whole 16-frame loop and around each call kind.
   python tools/bench_sampler_pair.py [N=16] [chunk=4] [delta_amp=0.03]"""
import itertools
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from emoportraits_amd import ops  # noqa: E402

DEV = "cuda:0"
C, D, S = 96, 16, 64


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    amp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
    import restate as O
    g = torch.Generator().manual_seed(1)
    vol = torch.randn(1, C, D, S, S, generator=g).to(DEV)
    theta = O.get_transform_matrix(1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g),
                                   0.05 * torch.randn(N, 3, generator=g))[:, :3].contiguous().to(DEV)
    delta = (torch.tanh(torch.randn(N, 3, D, S, S, generator=g)) * amp).to(DEV)
    vcl, vp4 = ops.volume_to_channels_last(vol), ops.volume_to_p4(vol)
    mid_cl = torch.empty(N, D, S, S, C, device=DEV)
    mid_p4 = torch.empty(N, C // 4, D, S, S, 4, device=DEV)
    out = torch.empty(N, C, D, S, S, device=DEV)
    vol_bytes, grid_bytes = C * D * S * S * 4, D * S * S * 12
    uv_bytes = vol_bytes + N * (grid_bytes + vol_bytes)
    rot_bytes = N * 2 * vol_bytes
    tv = ops.tile_variant

    def run(kind, uvv, rotv, ev=None):
        cl_uv_variant = 1 if kind == "cl_bricks" else 0
        for a in range(0, N, chunk):
            b = min(N, a + chunk)
            if ev is not None:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
            if kind in ("cl", "cl_bricks"):
                ops.grid_sample3d(vcl, delta=delta[a:b], in_layout="ndhwc", out_layout="ndhwc", out=mid_cl[a:b], variant=cl_uv_variant)
            else:
                ops.grid_sample3d(vp4, delta=delta[a:b], in_layout="p4", out_layout="p4", out=mid_p4[a:b], variant=uvv)
            if ev is not None:
                e1.record()
            if kind in ("cl", "cl_bricks"):
                ops.grid_sample3d(mid_cl[a:b], theta=theta[a:b], in_layout="ndhwc", out_layout="ncdhw", out=out[a:b])
            else:
                ops.grid_sample3d(mid_p4[a:b], theta=theta[a:b], in_layout="p4", out_layout="ncdhw", out=out[a:b], variant=rotv)
            if ev is not None:
                e2.record()
                ev.append((e0, e1, e2))

    def measure(name, kind, uvv=0, rotv=0, iters=12):
        for _ in range(3):
            run(kind, uvv, rotv)
        torch.cuda.synchronize()
        tot, uv_t, rot_t = [], [], []
        for _ in range(iters):
            ev = []
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run(kind, uvv, rotv, ev)
            e.record()
            torch.cuda.synchronize()
            tot.append(s.elapsed_time(e))
            uv_t.append(sum(a.elapsed_time(b) for a, b, _ in ev))
            rot_t.append(sum(b.elapsed_time(c) for _, b, c in ev))
        med = lambda x: sorted(x)[len(x) // 2]
        t, u, r = med(tot), med(uv_t), med(rot_t)
        print(json.dumps(dict(case=name, N=N, chunk=chunk, delta_amp=amp, pair_us_per_frame=round(t * 1e3 / N, 2),
                              uv_us_per_frame=round(u * 1e3 / N, 2), rot_us_per_frame=round(r * 1e3 / N, 2),
                              pair_frac_of_8TBps=round((uv_bytes + rot_bytes) / (t * 1e-3) / 8e12, 3),
                              uv_GBps=round(uv_bytes / (u * 1e-3) / 1e9), rot_GBps=round(rot_bytes / (r * 1e-3) / 1e9))), flush=True)

    measure("channels-last pair: uv rows + rot (driver pass default)", "cl")
    measure("channels-last pair: uv 4x4x4 bricks + rot", "cl_bricks")
    if "--tiles" not in sys.argv:
        return
    T = ops.TILE
    tv = lambda *a, **k: T | ops.tile_variant(*a, **k)
    uvs = [("16x4x8 u24", tv((8, 4, 16), 24)), ("16x8x8 t512 u24", tv((8, 8, 16), 24, threads=512)), ("4x8x8 u24", tv((8, 8, 4), 24)),
           ("16x4x8 u12", tv((8, 4, 16), 12)), ("16x4x8 u6", tv((8, 4, 16), 6)), ("16x4x8 u3", tv((8, 4, 16), 3)),
           ("16x8x8 t512 u12", tv((8, 8, 16), 12, threads=512)), ("16x8x8 t512 u6", tv((8, 8, 16), 6, threads=512)),
           ("16x8x8 t512 u3", tv((8, 8, 16), 3, threads=512)), ("8x8x8 u24", tv((8, 8, 8), 24)), ("16x2x16 u24", tv((16, 2, 16), 24)),
           ("8x4x16 u24", tv((16, 4, 8), 24))]
    rots = [("4x8x16 u24", tv((16, 8, 4), 24)), ("4x16x16 t512 u24", tv((16, 16, 4), 24, threads=512)), ("4x8x8 u24", tv((8, 8, 4), 24)),
            ("4x8x16 u12", tv((16, 8, 4), 12)), ("4x8x16 u6", tv((16, 8, 4), 6)), ("4x16x16 t512 u12", tv((16, 16, 4), 12, threads=512)),
            ("2x8x16 u24", tv((16, 8, 2), 24)), ("4x4x16 u24", tv((16, 4, 4), 24)), ("2x8x32 u24", tv((32, 8, 2), 24)),
            ("4x4x32 u24", tv((32, 4, 4), 24)), ("2x16x16 u24", tv((16, 16, 2), 24)), ("4x8x32 t512 u24", tv((32, 8, 4), 24, threads=512)),
            ("8x8x16 t512 u24", tv((16, 8, 8), 24, threads=512))]
    for (un, uvv) in uvs:
        measure(f"tile uv {un} | rot {rots[0][0]}", "p4", uvv, rots[0][1])
    for (rn, rotv) in rots[1:]:
        measure(f"tile uv {uvs[0][0]} | rot {rn}", "p4", uvs[0][1], rotv)


if __name__ == "__main__":
    main()
