"""TEST INFRASTRUCTURE ONLY (build container) -- pins oracle/restate.py against the reference's own
nn.Modules (imported from /root/reference through oracle/ref_harness.py).

usage:  python oracle/validate_restatement.py [256|512] [--out oracle/VALIDATION_<S>.json]

Seeded random-init weights (the released checkpoint is not in the repo: README.md:125-139 points to
Google Drive), 1-D parameters perturbed so norm affines/biases are exercised, synthetic inputs as in
SURVEY.md section 8(d) config 1 plus the "realistic warp" variant.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as R  # noqa: E402
import restate as O      # noqa: E402


def stats(name, a, b):
    a, b = a.float(), b.float()
    d = (a - b).abs()
    denom = b.abs().max().item() + 1e-30
    r = dict(name=name, shape=list(a.shape), max_abs=d.max().item(), rel_to_max=d.max().item() / denom,
             ref_absmax=b.abs().max().item())
    print(f"  {name:28s} max_abs={r['max_abs']:.3e} rel={r['rel_to_max']:.3e} |ref|max={r['ref_absmax']:.3e}")
    return r


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    out = None
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    torch.set_num_threads(os.cpu_count())
    args = R.released_args(S)
    h = R.build_holder(args, seed=0)
    R.randomize_affines(h, seed=123, scale=0.2)
    sd = {k: v.detach().clone() for k, v in h.state_dict().items()}
    cfg = O.cfg_from_args(args)
    g = torch.Generator().manual_seed(7)
    img = torch.rand(1, 3, S, S, generator=g)
    idt = torch.randn(1, 512, 4, 4, generator=g)
    pose_s = torch.randn(1, 128, generator=g)
    pose_t = torch.randn(1, 128, generator=g)
    install_pt = R.install  # noqa: F841
    from utils import point_transforms
    srt = lambda: (1 + 0.05 * torch.randn(1, 3, generator=g), 0.3 * torch.randn(1, 3, generator=g),
                   0.05 * torch.randn(1, 3, generator=g))
    a, b = srt(), srt()
    th_s_ref = point_transforms.get_transform_matrix(*a)
    th_t_ref = point_transforms.get_transform_matrix(*b)
    results = []
    print("get_transform_matrix")
    results.append(stats("theta_src", O.get_transform_matrix(*a), th_s_ref))
    results.append(stats("theta_drv", O.get_transform_matrix(*b), th_t_ref))

    print(f"source pass R{S}")
    t = time.time()
    ref_s = R.reference_source_pass(h, img, idt, pose_s, th_s_ref)
    t_ref = time.time() - t
    t = time.time()
    with torch.no_grad():
        our_s = O.source_pass(sd, cfg, img, idt, pose_s, th_s_ref)
    t_our = time.time() - t
    for k in ("warp_embed", "latents", "source_rotation_warp", "xy_warp", "source_volume", "pre_canonical", "canonical"):
        results.append(stats("src." + k, our_s[k], ref_s[k]))
    print(f"  reference {t_ref:.2f}s  restatement {t_our:.2f}s")

    print(f"driver pass R{S}")
    t = time.time()
    ref_d = R.reference_driver_pass(h, ref_s["canonical"], idt, pose_t, th_t_ref, img)
    t_ref_d = time.time() - t
    t = time.time()
    with torch.no_grad():
        our_d = O.driver_pass(sd, cfg, ref_s["canonical"], idt, pose_t, th_t_ref)
    t_our_d = time.time() - t
    for k in ("warp_embed", "target_rotation_warp", "uv_warp", "delta_uv", "aligned", "deep_f", "img_f", "img"):
        results.append(stats("drv." + k, our_d[k], ref_d[k]))
    print(f"  reference {t_ref_d:.2f}s  restatement {t_our_d:.2f}s")

    print("sampler restatement (numpy, from first principles) vs torch CPU F.grid_sample")
    import numpy as np
    import torch.nn.functional as F
    gg = torch.Generator().manual_seed(11)
    vol = torch.randn(2, 5, 4, 6, 7, generator=gg)
    grid = torch.rand(2, 3, 5, 9, 3, generator=gg) * 2.6 - 1.3
    for pm in ("zeros", "border", "reflection"):
        ref = F.grid_sample(vol, grid, padding_mode=pm, align_corners=False).numpy()
        ours = O.grid_sample3d_restated(vol.numpy(), grid.numpy(), pm)
        d = np.abs(ref - ours).max()
        print(f"  {pm:10s} max_abs={d:.3e}")
        results.append(dict(name="sampler." + pm, max_abs=float(d)))

    print("stage 2 (infer_s2.py:351-376), default flags (BatchNorm) and GroupNorm+WS variant")
    for ov in (dict(output_size_s2=S), dict(output_size_s2=S, norm_layer_type="gn", use_ws=True)):
        a2 = R.stage2_args(ov)
        h2 = R.build_stage2_holder(a2, seed=1)
        R.make_trained_like(h2)
        R.randomize_affines(h2, seed=5, scale=0.1)
        R.randomize_bn_stats(h2)
        sd2 = {k: v.detach().clone() for k, v in h2.state_dict().items()}
        gg2 = torch.Generator().manual_seed(3)
        im = torch.rand(1, 3, S, S, generator=gg2)
        mk = (torch.rand(1, 1, S, S, generator=gg2) > 0.2).float()
        fm = (torch.rand(1, 1, S, S, generator=gg2) > 0.3).float()
        ref2 = R.reference_stage2(h2, im, mk, fm)
        with torch.no_grad():
            our2 = O.stage2_forward(sd2, O.stage2_cfg_from_args(a2), im, mk, fm)
        for k in ("latents", "add", "out"):
            results.append(stats(f"s2[{a2.norm_layer_type}].{k}", our2[k], ref2[k]))

    print("embedders (section 8f-1): reference IdtEmbed / HeadPoseRegressor / ExpressionEmbed on the restated torchvision ResNets")
    he = R.build_embedder_holder(args, seed=2)
    R.make_trained_like(he)
    R.randomize_affines(he, seed=9, scale=0.1)
    R.randomize_bn_stats(he.head_pose_regressor.net)
    sde = {k: v.detach().clone() for k, v in he.state_dict().items()}
    sdh = {k: v.detach().clone() for k, v in he.head_pose_regressor.net.state_dict().items()}
    ge = torch.Generator().manual_seed(3)
    crops = torch.rand(2, 3, S, S, generator=ge)
    with torch.no_grad():
        results.append(stats("emb.idt_embed", O.idt_embed(sde, "idt_embedder_nw", crops[:1]), R.reference_idt_embed(he, crops[:1])))
        rp, op = R.reference_head_pose(he, crops), O.head_pose(sdh, crops)
        for k in rp:
            results.append(stats("emb.head_pose." + k, op[k], rp[k]))
        rex = R.reference_expression(he, crops, rp["theta"])
        oex = O.expression_embed(sde, "expression_embedder_nw", crops, rp["theta"])
        results.append(stats("emb.img_align", oex["img_align"], rex["img_align"]))
        results.append(stats("emb.align_warp", oex["align_warp"], rex["align_warp"][:2]))
        results.append(stats("emb.pose_embed", oex["pose_embed"], rex["pose_embed"]))   # reference batch is cat(src, tgt)

    summary = dict(image_size=S, torch=torch.__version__, threads=torch.get_num_threads(),
                   reference_source_s=t_ref, reference_driver_s=t_ref_d, results=results)
    if out:
        with open(out, "w") as f:
            json.dump(summary, f, indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
