"""Micro-benchmark: hand-written implicit-GEMM conv (fp32 MFMA) vs stock PyTorch-ROCm (MIOpen) on the released
decoder / WarpGenerator layer shapes.  In-process A/B, HIP events on the launch stream.  JSON lines."""
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import ops, pack  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def add_best(rec):
    """best hand-written kernel of the row against stock torch on the SAME work: the pointwise rows compare on the source grid (how
    the network runs them), every other row on the row's own launch"""
    pw = rec.get("pointwise_on_the_source_grid")
    if pw:
        rec["best_hip_tflops"], rec["torch_same_work_tflops"] = max(pw["f16x2_tflops"], pw["fp32_mfma_tflops"]), pw["torch_tflops"]
    else:
        hip = [v for k, v in rec.items() if k.endswith("_tflops") and not k.startswith("torch") and v]
        hip += [rec["planner"]["tflops"]] if isinstance(rec.get("planner"), dict) else []
        rec["best_hip_tflops"], rec["torch_same_work_tflops"] = (max(hip) if hip else None), rec.get("torch_tflops")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    quick = "--quick" in sys.argv
    f16 = "--f16" in sys.argv or "--f16-only" in sys.argv   # also time the opt-in fp16-operand kernels (emo_conv_igemm_f16acc32)
    f16_only = "--f16-only" in sys.argv
    split = "--bf16x3" in sys.argv or "--bf16x3-only" in sys.argv   # also time emo_conv_igemm_bf16x3 (fp32 on the bf16 pipes)
    split_only = "--bf16x3-only" in sys.argv
    f16_only = f16_only or split_only
    torch.backends.cudnn.benchmark = True
    # (Cin, Cout, dims, k, ups)
    shapes = [(1536, 512, (64, 64), 1, False), (512, 512, (64, 64), 3, False),
              (512, 320, (64, 64), 3, True), (320, 320, (128, 128), 3, False), (512, 320, (64, 64), 1, True),
              (320, 192, (128, 128), 3, True), (192, 192, (256, 256), 3, False),
              (192, 128, (256, 256), 3, True), (128, 128, (512, 512), 3, False), (192, 128, (256, 256), 1, True),
              (128, 3, (512, 512), 1, False),
              (512, 256, (8, 8, 8), 3, False), (256, 128, (16, 16, 16), 3, False), (128, 64, (32, 32, 32), 3, False),
              (64, 32, (32, 64, 64), 3, False), (32, 32, (32, 64, 64), 3, False), (32, 3, (16, 64, 64), 3, False)]
    if quick:
        shapes = [sh for sh in shapes if sh[3] == 3 and sh[1] >= 64 and sh[2][-1] >= 32]
    for cin, cout, dims, k, ups in shapes:
        three_d = len(dims) == 3
        x = torch.randn(B, cin, *dims, device=DEV)
        kd = k if three_d else 1
        w = torch.randn(cout, cin, *([k] * len(dims))) / math.sqrt(cin * k * k * kd)
        if "--zeros" in sys.argv:       # all-zero operands: no toggling in the matrix pipes -- what the SCHEDULE gives when power does not bind
            x.zero_()
            w.zero_()
        scale = torch.rand(B, cin, device=DEV) + 0.5
        shift = torch.randn(B, cin, device=DEV) * 0.1
        odims = tuple(d * 2 for d in dims) if ups else dims
        flops = 2.0 * B * cout * cin * (k ** len(dims)) * math.prod(odims)
        wd = w.to(DEV)
        xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
        conv = F.conv3d if three_d else F.conv2d
        ms_t = 0.0 if quick else timeit(lambda: conv(xin, wd, padding=k // 2))
        rec = dict(B=B, cin=cin, cout=cout, dims=dims, k=k, ups=ups, torch_ms=round(ms_t, 3),
                   torch_tflops=round(flops / ms_t / 1e9, 1) if ms_t else None)
        for cfg in (() if f16_only else (0, 1, 2, 3, 4, 5)):
            bm = {0: 128, 1: 64, 2: 32, 3: 64, 4: 64, 5: 32}[cfg]
            if cfg == 2 and cout > 320:
                continue
            if cfg in (3, 4, 5) and (k != 3 or dims[-1] < 32 or (three_d and cfg == 4)):
                continue
            if (cfg in (3, 4) and cout < 64) or (cfg == 5 and cout > 96):
                continue
            if -(-cout // bm) * bm > 1.5 * cout and cfg != pack.choose_cfg(cout):
                continue
            layer = pack.PackedConv("b", w, None, DEV, cfg=cfg)
            out = ops.conv_igemm(x, layer, scale, shift, relu_in=True, ups=ups)
            ms = timeit(lambda: ops.conv_igemm(x, layer, scale, shift, relu_in=True, ups=ups, out=out))
            rec[f"hip_cfg{cfg}_ms"] = round(ms, 3)
            rec[f"hip_cfg{cfg}_tflops"] = round(flops / ms / 1e9, 1)
        if f16 and k in (1, 3) and cin % 8 == 0 and cout >= 32 and pack.f16_launch_fits(odims[-2], odims[-1]):
            lh = pack.PackedConv("b16", w, None, DEV, precision="f16")       # fp16 operands: 64 x 256 tile (cfg 3)
            out = ops.conv_igemm(x, lh, scale, shift, relu_in=True, ups=ups)
            msh = timeit(lambda: ops.conv_igemm(x, lh, scale, shift, relu_in=True, ups=ups, out=out))
            rec["f16_cfg3_ms"] = round(msh, 3)
            rec["f16_cfg3_tflops"] = round(flops / msh / 1e9, 1)
        fits = split and k == 3 and pack.bf16x3_launch_fits(odims[-2], odims[-1], ups)
        if fits and pack.supports_bf16x3(cout, cin, kd, k, k):
            ls = pack.PackedConv("b3", w, None, DEV, precision="bf16x3")
            out = ops.conv_igemm(x, ls, scale, shift, relu_in=True, ups=ups)
            mss = timeit(lambda: ops.conv_igemm(x, ls, scale, shift, relu_in=True, ups=ups, out=out))
            rec["bf16x3_ms"] = round(mss, 3)
            rec["bf16x3_tflops"] = round(flops / mss / 1e9, 1)      # fp32-equivalent (algorithmic) FLOPs
        if fits and "--f16x2" in sys.argv and pack.supports_bf16x3(cout, cin, kd, k, k, "f16x2"):   # two-term fp16 split (SPLIT = 2)
            l2 = pack.PackedConv("h2", w, None, DEV, precision="f16x2")
            out = ops.conv_igemm(x, l2, scale, shift, relu_in=True, ups=ups)
            ms2 = timeit(lambda: ops.conv_igemm(x, l2, scale, shift, relu_in=True, ups=ups, out=out))
            rec["f16x2_ms"] = round(ms2, 3)
            rec["f16x2_tflops"] = round(flops / ms2 / 1e9, 1)
        if "--f16x2" in sys.argv and k == 1 and not three_d and pack.supports_f16x2_pointwise(cout, cin, 1, 1, 1):
            # pointwise layers of the default mode (csrc/conv_igemm_f16x2_p1.h).  A 1x1 skip commutes with the nearest upsample, so
            # the network runs it on the PRE-upsample tensor (nets.ResBlock): a quarter of the positions -- timed that way, with
            # stock torch on the same tensor beside it; no input affine (these layers have none: decoder.py:66-70, utils.py:764-781)
            lp = pack.PackedConv("p1", w, None, DEV, precision="f16x2")
            pos_src = B * math.prod(dims)
            if lp.plan_for(max(1, -(-pos_src // 128)), dims[-2], dims[-1])[2] == "f16x2":
                fl_src = 2.0 * B * cout * cin * math.prod(dims)
                out = ops.conv_igemm(x, lp)
                msp = timeit(lambda: ops.conv_igemm(x, lp, out=out))
                ms_ts = timeit(lambda: conv(x, wd))
                l32 = pack.PackedConv("p32", w, None, DEV)
                out32 = ops.conv_igemm(x, l32)
                ms32 = timeit(lambda: ops.conv_igemm(x, l32, out=out32))
                rec["pointwise_on_the_source_grid"] = dict(f16x2_ms=round(msp, 3), f16x2_tflops=round(fl_src / msp / 1e9, 1),
                                                          fp32_mfma_ms=round(ms32, 3), fp32_mfma_tflops=round(fl_src / ms32 / 1e9, 1),
                                                          torch_ms=round(ms_ts, 3), torch_tflops=round(fl_src / ms_ts / 1e9, 1))
        if not f16_only:       # what the planner picks for this launch (block config + K split), as the networks run it
            la = pack.PackedConv("auto", w, None, DEV)
            Hl_, Wl_ = odims[-2], odims[-1]
            pos = B * math.prod(odims)
            cfg_a, ks_a, _ = la.plan_for(max(1, -(-pos // 128)), Hl_, Wl_, ups, affine=True)
            out = ops.conv_igemm(x, la, scale, shift, relu_in=True, ups=ups)
            ms = timeit(lambda: ops.conv_igemm(x, la, scale, shift, relu_in=True, ups=ups, out=out))
            rec["planner"] = dict(cfg=cfg_a, ksplit=ks_a, ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1))
        rec["auto_cfg"] = pack.choose_cfg(cout)
        add_best(rec)
        print(json.dumps(rec), flush=True)
    # GroupNorm statistics kernel vs torch group_norm+relu (which the conv staging makes unnecessary)
    for c, h in ([] if quick else [(512, 64), (320, 128), (192, 256), (128, 512)]):
        x = torch.randn(B, c, h, h, device=DEV)
        g, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        ms = timeit(lambda: ops.groupnorm_affine(x, g, b))
        ms_t = timeit(lambda: F.relu(F.group_norm(x, 32, g, b)))
        print(json.dumps(dict(op="gn_stats", B=B, c=c, hw=h, hip_ms=round(ms, 4), read_GBps=round(x.numel() * 4 / ms / 1e6, 1),
                              torch_gn_relu_ms=round(ms_t, 4))), flush=True)


if __name__ == "__main__":
    main()
