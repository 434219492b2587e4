"""diagnostic: per-stage error of the fp16-operand mode vs the fp32 HIP path (same weights, same inputs) at R256"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from emoportraits_amd import config, nets, random_init
from test_nets_gpu import _full_size
DEV = "cuda:0"
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
for seed, B, gain in ((17, 1, None), (31, 2, 0.2), (31, 1, 0.2), (17, 2, None)):
    cfg = config.hot_path_config(overrides={"image_size": 256})
    sd = random_init.random_state_dict(cfg, seed=seed, image_head_gain=gain)
    _, _, x = _full_size(256, B, seed=seed)
    d = lambda t: t.to(DEV)
    outs = {}
    for prec in ("f32", "f16"):
        hp = nets.HotPath(sd, cfg, DEV, with_source=False, precision=prec)
        outs[prec] = hp.driver_pass(hp.prepare_canonical(d(x["canonical"])), d(x["idt"]), d(x["pose_t"]), d(x["th_t"]), keep=True)
    rec = dict(seed=seed, B=B, gain=gain)
    for k in ("warp_embed", "delta_uv", "aligned", "deep_f", "img_f", "img"):
        rec[k] = f"{rel(outs['f16'][k], outs['f32'][k]):.2e}"
    rec["delta_uv_abs"] = f"{(outs['f16']['delta_uv'] - outs['f32']['delta_uv']).abs().max().item():.2e}"
    # decoder alone on the fp32 aligned tensor
    hp16 = nets.HotPath(sd, cfg, DEV, with_source=False, precision="f16")
    img, deep_f, img_f = hp16.decoder(outs["f32"]["aligned"].view(B, -1, 64, 64))
    rec["decoder_only_deep_f"] = f"{rel(deep_f, outs['f32']['deep_f']):.2e}"
    rec["decoder_only_img_f"] = f"{rel(img_f, outs['f32']['img_f']):.2e}"
    print(json.dumps(rec), flush=True)
