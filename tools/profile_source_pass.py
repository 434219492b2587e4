"""Per-launch conv timing of the once-per-identity source pass (HIP events around every conv_igemm launch), per conv mode.
    python tools/profile_source_pass.py [512]   ->   JSON lines: the slowest launches and the totals of each mode"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import config, nets, ops, random_init  # noqa: E402

DEV = "cuda:0"


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    cfg = config.hot_path_config(overrides={"image_size": S})
    sd = random_init.trained_like_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(1, 3, S, S, generator=g).to(DEV)
    idt = torch.randn(1, cfg["gen_max_channels"], 4, 4, generator=g).to(DEV)
    pose = torch.randn(1, cfg["lpe_output_channels_expression"], generator=g).to(DEV)
    th = ops.pose_theta(*[t.to(DEV) for t in (1 + 0.05 * torch.randn(1, 3, generator=g), 0.3 * torch.randn(1, 3, generator=g),
                                              0.05 * torch.randn(1, 3, generator=g))])
    orig = ops.conv_igemm
    for prec in ("f32", "bf16x3"):
        hp = nets.HotPath(sd, cfg, DEV, precision=prec)
        for _ in range(2):
            hp.source_pass(img, idt, pose, th)
        torch.cuda.synchronize()
        recs = []

        def wrapped(x, layer, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ret = orig(x, layer, *a, **kw)
            e1.record()
            recs.append((e0, e1, layer.name, tuple(x.shape), layer.cout, layer.last_plan))
            return ret

        ops.conv_igemm = wrapped
        nets.ops.conv_igemm = wrapped
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hp.source_pass(img, idt, pose, th)
        e1.record()
        torch.cuda.synchronize()
        ops.conv_igemm = orig
        nets.ops.conv_igemm = orig
        rows = sorted(((a.elapsed_time(b), n, sh, co, pl) for a, b, n, sh, co, pl in recs), reverse=True)
        print(json.dumps(dict(mode=prec, source_pass_ms=round(e0.elapsed_time(e1), 2), conv_ms=round(sum(r[0] for r in rows), 2),
                              launches=len(rows))), flush=True)
        for ms, n, sh, co, pl in rows[:12]:
            print(json.dumps(dict(mode=prec, ms=round(ms, 3), layer=n, x=sh, cout=co, cfg=pl[0], ksplit=pl[1], kernel=pl[2])), flush=True)
        del hp


if __name__ == "__main__":
    main()
