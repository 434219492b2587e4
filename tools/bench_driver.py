"""Stage breakdown of the HIP driver pass (per batch of B frames) at the released architecture.  JSON lines."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import config, graphs, nets, ops, random_init  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=7, warmup=2):
    """median of per-call HIP-event times (a mean lets one hiccup -- a lazy weight packing, a clock ramp -- leak into a stage)"""
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
    return ts[len(ts) // 2]


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    # --f16: opt-in fp16-operand convs (reduced precision); --bf16x3 / --f32: the two fp32 modes (default: nets.DEFAULT_PRECISION)
    precision = ("f16" if "--f16" in sys.argv else "bf16x3" if "--bf16x3" in sys.argv else "f32" if "--f32" in sys.argv else
                 "f16x2" if "--f16x2" in sys.argv else None)
    Bs = [int(a) for a in sys.argv[2:] if not a.startswith("--")] or [1, 4, 8, 16]
    cfg = config.hot_path_config(overrides={"image_size": S})
    sd = random_init.random_state_dict(cfg, seed=0, with_source=False)
    hp = nets.HotPath(sd, cfg, DEV, with_source=False, precision=precision)
    g = torch.Generator().manual_seed(1)
    canonical = torch.randn(1, 96, 16, 64, 64, generator=g).to(DEV)
    ccl = hp.prepare_canonical(canonical)
    idt = torch.randn(1, 512, 4, 4, generator=g).to(DEV)
    for B in Bs:
        pose = torch.randn(B, 128, generator=g).to(DEV)
        srt = [t.to(DEV) for t in (1 + 0.05 * torch.randn(B, 3, generator=g), 0.3 * torch.randn(B, 3, generator=g),
                                  0.05 * torch.randn(B, 3, generator=g))]
        theta = ops.pose_theta(*srt)
        emb = hp.embed(pose, idt)
        delta = hp.uv_generator(emb)
        lay = "ndhwc"
        warped = ops.grid_sample3d(ccl, delta=delta, in_layout=lay, out_layout=lay)
        aligned = ops.grid_sample3d(warped, theta=theta, in_layout=lay, out_layout="ncdhw")
        feat = aligned.view(B, 96 * 16, 64, 64)
        rec = dict(S=S, B=B, conv_operands=hp.precision)
        rec["embed_ms"] = timeit(lambda: hp.embed(pose, idt))
        rec["warpgen_ms"] = timeit(lambda: hp.uv_generator(emb))
        rec["sampler_uv_ms"] = timeit(lambda: ops.grid_sample3d(ccl, delta=delta, in_layout=lay, out_layout=lay))
        rec["sampler_rot_ms"] = timeit(lambda: ops.grid_sample3d(warped, theta=theta, in_layout=lay, out_layout="ncdhw"))
        rec["decoder_ms"] = timeit(lambda: hp.decoder(feat))
        rec["total_ms"] = timeit(lambda: hp.driver_pass(ccl, idt, pose, theta))
        rec["fps"] = B / rec["total_ms"] * 1e3
        graphed = graphs.Graphed(lambda p, t: hp.driver_pass(ccl, idt, p, t), clone_outputs=False)
        rec["graph_total_ms"] = timeit(lambda: graphed(pose, theta))
        rec["graph_fps"] = B / rec["graph_total_ms"] * 1e3
        rec = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rec.items()}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
