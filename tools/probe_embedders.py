"""Prints the GPU-vs-oracle errors and timings of the embedder path (used to set the test tolerances)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emoportraits_amd import embedders as E  # noqa: E402

dev = torch.device("cuda:0")
blob = torch.load(os.path.join(ROOT, "tests", "golden", "embedders.pt"), weights_only=False)
cfg, seeds = blob["cfg"], blob["seeds"]
sds = dict(idt=E.random_state_dict(E.idt_schema(cfg), seeds["idt"]),
           expression=E.random_state_dict(E.expression_schema(cfg), seeds["expression"]),
           head_pose=E.random_state_dict(E.head_pose_schema(), seeds["head_pose"]))
crops = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(seeds["inputs"])).to(dev)
rel = lambda a, b: ((a.cpu().double() - b.double()).abs().max() / b.double().abs().max()).item()
idt, hp, ex = E.IdtEmbed(sds["idt"], cfg, dev), E.HeadPoseRegressor(sds["head_pose"], dev), E.ExpressionEmbed(sds["expression"], cfg, dev)
print("idt", rel(idt(crops[:1]), blob["idt_embed"]))
th = hp.forward(crops, True)
for n, g in zip(("theta", "scale", "rotation", "translation"), th):
    print("hp." + n, rel(g, blob["head_pose"][n]))
pose, al, warp = ex.forward(crops, blob["theta"].to(dev), want_aligned=True)
print("ex.warp", (warp.cpu()[:, ::8, ::8] - blob["align_warp_sub"]).abs().max().item())
print("ex.aligned", (al.cpu()[:, :, ::8, ::8] - blob["img_align_sub"]).abs().max().item())
print("ex.pose", rel(pose, blob["pose_embed"]))
print("ex.chain", rel(ex(crops, th[0]), blob["pose_embed_chain"]))
for B in (1, 16):
    x = torch.rand(B, 3, 512, 512, device=dev)
    for name, fn in (("head_pose", lambda: hp(x)), ("expression", lambda: ex(x, blob["theta"][:1].expand(B, -1, -1).to(dev))),
                     ("idt(B=1)", lambda: idt(x[:1]))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"B={B} {name}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms/call")
