"""TEST INFRASTRUCTURE ONLY (build container) -- generates the committed fixtures under tests/golden/ by running
the REAL reference (imported from /root/reference via oracle/ref_harness.py) and torch's CPU ops.

    python oracle/make_golden.py

The reference ships no golden vectors for this path, so these fixtures are "outputs of the
reference itself run here".  They are small on purpose (they travel in git); full-size parity is checked
on the GPU box against oracle/restate.py, which oracle/validate_restatement.py pins bit-exactly to the
reference at R256/R512 in this container.

Fixtures
  sampler_kat.npz    known-answer tests of F.grid_sample (5-D) incl. out-of-range / half-integer / exact-edge
                     coordinates, three padding modes.
  pose_theta.npz     utils/point_transforms.py:188-242 get_transform_matrix + the rotation warp of
                     va.py:101-105 / notebooks/infer.py:441-444,583-588 (incl. rotation clamp edge cases).
  hostglue.pt        crop windows / crops of InferenceWrapper.crop_image and poses of get_mixing_theta (notebooks/infer.py).
  embedders.pt       seeds + outputs of the reference's IdtEmbed / HeadPoseRegressor / ExpressionEmbed at full width.
  tiny_hotpath.pt    reduced-width released architecture (same code paths: SN, WS, ada-GN, up/down sampling):
                     raw state_dict + synthetic inputs + per-stage outputs of the reference source and driver passes.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TINY = dict(image_size=64, aug_warp_size=64,
            latent_volume_channels=32, latent_volume_depth=4, latent_volume_size=16,
            gen_latent_texture_channels=32, gen_latent_texture_depth=4, gen_latent_texture_size=16,
            gen_num_channels=32, gen_max_channels=64, gen_max_channels_unet3d=64, enc_channel_mult=1.0,
            gen_embed_size=4, gen_dummy_input_size=4, warp_output_size=16,
            source_volume_num_blocks=2, dec_num_blocks=2, dec_channel_mult=1.0, dec_max_channels=64,
            lpe_output_channels_expression=16)


def sampler_kat():
    g = torch.Generator().manual_seed(2024)
    vol = torch.randn(2, 4, 3, 5, 7, generator=g)
    # crafted coordinates: far out of range, exactly -1/+1 (image edges at -0.5 / size-0.5), pixel centres,
    # half-integer source indices, tiny offsets around cell boundaries
    special = torch.tensor([-3.0, -1.5, -1.0, -1.0 + 1e-6, -0.999999, -0.5, -1e-7, 0.0, 1e-7, 0.3333333, 0.5,
                            1.0 - 1e-6, 1.0, 1.0 + 1e-6, 1.25, 2.9])
    W, H, D = 7, 5, 3
    centres_x = (2 * torch.arange(W) + 1) / W - 1       # unnormalise to exact integers
    halves_x = (2 * torch.arange(W + 1)) / W - 1        # unnormalise to k - 0.5
    xs = torch.cat([special, centres_x, halves_x])
    ys = torch.cat([special, (2 * torch.arange(H) + 1) / H - 1, (2 * torch.arange(H + 1)) / H - 1])
    zs = torch.cat([special, (2 * torch.arange(D) + 1) / D - 1, (2 * torch.arange(D + 1)) / D - 1])
    n = 4 * 6 * 9
    pick = lambda v, seed: v[torch.randint(0, v.numel(), (2 * n,), generator=torch.Generator().manual_seed(seed))]
    grid = torch.stack([pick(xs, 1), pick(ys, 2), pick(zs, 3)], dim=-1).view(2, 4, 6, 9, 3)
    # half of the points: smooth random in [-1.2, 1.2]
    rnd = torch.rand(2, 4, 6, 9, 3, generator=g) * 2.4 - 1.2
    mask = (torch.rand(2, 4, 6, 9, 1, generator=g) < 0.5)
    grid = torch.where(mask, rnd, grid).contiguous()
    out = {"vol": vol.numpy(), "grid": grid.numpy()}
    for pm in ("zeros", "border", "reflection"):
        out["out_" + pm] = F.grid_sample(vol, grid, padding_mode=pm, align_corners=False).numpy()
        # shared-volume form: one volume, 2 grids
        out["out_shared_" + pm] = F.grid_sample(vol[:1].expand(2, -1, -1, -1, -1), grid, padding_mode=pm,
                                                align_corners=False).numpy()
    np.savez_compressed(os.path.join(OUT, "sampler_kat.npz"), **out)
    print("sampler_kat.npz", {k: v.shape for k, v in out.items()})


def pose_theta():
    R.install()
    from utils import point_transforms
    g = torch.Generator().manual_seed(5)
    scale = 1 + 0.1 * torch.randn(8, 3, generator=g)
    rot = 0.6 * torch.randn(8, 3, generator=g)
    rot[0] = torch.tensor([-2.0, 3.5, 0.1])              # clamp(-pi/2, pi) edge cases (point_transforms.py:211)
    rot[1] = torch.tensor([-math.pi / 2, math.pi, 0.0])
    trans = 0.1 * torch.randn(8, 3, generator=g)
    theta = point_transforms.get_transform_matrix(scale, rot, trans)
    scale1 = scale[:, :1].contiguous()                    # scale.shape[1] == 1 branch (:203-206)
    theta1 = point_transforms.get_transform_matrix(scale1, rot, trans)
    d, s = 16, 64
    grid_s = torch.linspace(-1, 1, s)
    grid_z = torch.linspace(-1, 1, d)
    w, v, u = torch.meshgrid(grid_z, grid_s, grid_s, indexing="ij")      # va.py:101-105
    ident = torch.stack([u, v, w, torch.ones_like(u)], dim=3).view(1, -1, 4)
    # driver form (infer.py:583-586), batch 1 per call as the reference does
    warps = torch.cat([ident.bmm(theta[i:i + 1, :3].transpose(1, 2)).view(1, d, s, s, 3) for i in range(3)])
    inv = theta[:3].float().inverse()                                    # source form (infer.py:443-444)
    warps_inv = torch.cat([ident.bmm(inv[i:i + 1, :3].transpose(1, 2)).view(1, d, s, s, 3) for i in range(3)])
    # keep the fixture small: a strided subset of lattice points
    sub = (slice(None), slice(0, d, 5), slice(0, s, 9), slice(0, s, 7))
    out = dict(scale=scale.numpy(), rotation=rot.numpy(), translation=trans.numpy(), theta=theta.numpy(),
               theta_scalar_scale=theta1.numpy(), theta_inv=inv.numpy(),
               warp_sub=warps[sub].numpy(), warp_inv_sub=warps_inv[sub].numpy(),
               lin_s=grid_s.numpy(), lin_z=grid_z.numpy())
    np.savez_compressed(os.path.join(OUT, "pose_theta.npz"), **out)
    print("pose_theta.npz", {k: v.shape for k, v in out.items()})


def tiny_hotpath():
    args = R.released_args(TINY["image_size"], overrides=TINY)
    h = R.build_holder(args, seed=3)
    R.randomize_affines(h, seed=321, scale=0.2)
    sd = {k: v.detach().clone() for k, v in h.state_dict().items()}
    g = torch.Generator().manual_seed(17)
    S = TINY["image_size"]
    img = torch.rand(1, 3, S, S, generator=g)
    C = TINY["gen_max_channels"]
    E = TINY["lpe_output_channels_expression"]
    idt = torch.randn(1, C, 4, 4, generator=g)
    pose_s = torch.randn(1, E, generator=g)
    pose_t = torch.randn(2, E, generator=g)              # two driver frames
    from utils import point_transforms
    srt = lambda n: (1 + 0.05 * torch.randn(n, 3, generator=g), 0.3 * torch.randn(n, 3, generator=g),
                     0.05 * torch.randn(n, 3, generator=g))
    th_s = point_transforms.get_transform_matrix(*srt(1))
    th_t = point_transforms.get_transform_matrix(*srt(2))
    src = R.reference_source_pass(h, img, idt, pose_s, th_s)
    drv = [R.reference_driver_pass(h, src["canonical"], idt, pose_t[i:i + 1], th_t[i:i + 1], img) for i in range(2)]
    keep_src = ("latents", "xy_warp", "source_volume", "pre_canonical", "canonical", "warp_embed")
    keep_drv = ("uv_warp", "aligned", "deep_f", "img_f", "img", "warp_embed")
    blob = dict(cfg={k: getattr(args, k) for k in __import__("restate").RELEASED_CFG},
                state_dict=sd, img=img, idt_embed=idt, source_pose_embed=pose_s, target_pose_embed=pose_t,
                theta_src=th_s, theta_drv=th_t,
                source={k: src[k] for k in keep_src},
                driver=[{k: d[k] for k in keep_drv} for d in drv])
    path = os.path.join(OUT, "tiny_hotpath.pt")
    torch.save(blob, path)
    print("tiny_hotpath.pt", os.path.getsize(path) / 1e6, "MB;", len(sd), "tensors,",
          sum(v.numel() for v in sd.values()) / 1e6, "M params")


TINY_S2 = dict(output_size_s2=64, gen_latent_texture_size2=16, gen_latent_texture_channels2=8, gen_latent_texture_depth=4,
               gen_num_channels=32, gen_max_channels=64, enc_channel_mult_stage2=1.0, dec_channel_mult_stage2=1.0,
               dec_num_blocks_stage2=2, dec_max_channels2=64)


def tiny_stage2():
    """reduced-width stage-2 model with its default flags (BatchNorm, no WS), trained-like spectral-norm vectors and
    non-trivial BN statistics; outputs of the reference's LocalEncoderOld + Decoder_stage2Old"""
    import restate as O
    args = R.stage2_args(TINY_S2)
    h = R.build_stage2_holder(args, seed=4)
    R.make_trained_like(h)
    R.randomize_affines(h, seed=6, scale=0.1)
    R.randomize_bn_stats(h, seed=8)
    sd = {k: v.detach().clone() for k, v in h.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    S = TINY_S2["output_size_s2"]
    img = torch.rand(2, 3, S, S, generator=g)
    mask = (torch.rand(2, 1, S, S, generator=g) > 0.15).float()
    face = (torch.rand(2, 1, S, S, generator=g) > 0.3).float()
    ref = R.reference_stage2(h, img, mask, face)
    blob = dict(cfg=O.stage2_cfg_from_args(args), state_dict=sd, img=img, mask=mask, face_mask=face, **ref)
    path = os.path.join(OUT, "tiny_stage2.pt")
    torch.save(blob, path)
    print("tiny_stage2.pt", os.path.getsize(path) / 1e6, "MB;", len(sd), "tensors")


def embedders_golden():
    """SURVEY.md section 8f-1.  The backbones are full-width by construction (the reference hard-codes 512 * expansion
    channels, identity_embedder.py:37 / expression_embedder.py:381), ~150 MB of weights -- too large for git.  The fixture
    therefore stores SEEDS + the reference's outputs: weights and inputs are regenerated on the test side with
    emoportraits_amd.embedders.random_state_dict (a CPU torch.Generator stream), and `checksums` guards that regeneration.
    Loading those state_dicts into the reference's own modules with strict=True also pins the checkpoint key schema."""
    sys.path.insert(0, os.path.dirname(HERE))
    from emoportraits_amd import embedders as E
    args = R.released_args(512)
    h = R.build_embedder_holder(args, seed=0)
    cfg = E.embedder_config(vars(args), released=False)
    seeds = dict(idt=11, expression=12, head_pose=13, inputs=14)
    sd_i = E.random_state_dict(E.idt_schema(cfg), seeds["idt"])
    sd_e = E.random_state_dict(E.expression_schema(cfg), seeds["expression"])
    sd_h = E.random_state_dict(E.head_pose_schema(), seeds["head_pose"])
    buffers = {k: v for k, v in h.state_dict().items() if k.endswith(E._BUFFERS)}
    h.load_state_dict({**sd_i, **sd_e, **buffers}, strict=True)
    h.head_pose_regressor.net.load_state_dict(sd_h, strict=True)
    g = torch.Generator().manual_seed(seeds["inputs"])
    crops = torch.rand(2, 3, 512, 512, generator=g)
    from utils import point_transforms
    theta = point_transforms.get_transform_matrix(1 + 0.05 * torch.randn(2, 3, generator=g),
                                                  0.3 * torch.randn(2, 3, generator=g), 0.05 * torch.randn(2, 3, generator=g))
    idt = R.reference_idt_embed(h, crops[:1])
    hp = R.reference_head_pose(h, crops)
    ex = R.reference_expression(h, crops, theta)
    ex_own = R.reference_expression(h, crops, hp["theta"])        # the wrapper's real chain: pose net -> alignment
    checks = {n: float(sum(v.double().sum() for v in sd.values())) for n, sd in (("idt", sd_i), ("expression", sd_e), ("head_pose", sd_h))}
    checks["crops"] = float(crops.double().sum())
    blob = dict(cfg=cfg, seeds=seeds, checksums=checks, theta=theta, idt_embed=idt, head_pose=hp,
                pose_embed=ex["pose_embed"], img_align_sub=ex["img_align"][:, :, ::8, ::8].clone(),
                align_warp_sub=ex["align_warp"][:2, ::8, ::8].clone(), pose_embed_chain=ex_own["pose_embed"])
    path = os.path.join(OUT, "embedders.pt")
    torch.save(blob, path)
    print("embedders.pt", os.path.getsize(path) / 1e3, "KB", checks)


def hostglue_golden():
    """notebooks/infer.py crop_image (:301-352, with remove_overflow :243-261 and the smoothed-crop state :317-327) and
    get_mixing_theta (:686-736), called unbound on a stub `self`.  The crop windows are recovered from coordinate-coded
    images by intercepting the module's F.interpolate (the crop the reference hands to the bicubic resize)."""
    m, _ = R.reference_wrapper_stub(32)
    rng = np.random.RandomState(5)
    seen = []
    real_interp = m.F.interpolate

    def spy(x, *a, **kw):
        seen.append((int(x[0, 0, 0, 0]), int(x[0, 1, 0, 0]), int(x.shape[-1]), int(x.shape[-2])))
        return real_interp(x, *a, **kw)

    def coded(h, w):
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        return torch.stack([xx, yy, torch.zeros_like(xx)])

    cases = []
    m.F.interpolate = spy
    try:
        for name, kw, frames in (
            ("independent", dict(), 8), ("scaled", dict(scale=1.3), 6),
            ("smoothed", dict(use_smoothed_crop=True), 8), ("smoothed_fast", dict(use_smoothed_crop=True, momentum=0.5), 8),
            ("fixed", dict(use_smoothed_crop=True, fixed_bounding_box=True), 5),
        ):
            _, stub = R.reference_wrapper_stub(32, momentum=kw.pop("momentum", 0.01),
                                               fixed_bounding_box=kw.pop("fixed_bounding_box", False))
            sizes, faces = [], []
            for i in range(frames):
                h, w = int(rng.randint(90, 260)), int(rng.randint(90, 260))
                if name != "independent":
                    h, w = 200, 240
                cx, cy, half = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(15, 90)     # boxes that overflow the image
                faces.append(None if (name == "independent" and i == 3) else
                             np.array([cx - half, cy - half * 1.2, cx + half, cy + half * 1.1]))
                sizes.append((h, w))
            seen.clear()
            imgs = [coded(h, w) for h, w in sizes]
            crops, check, scales = m.InferenceWrapper.crop_image(stub, imgs, faces, **kw)
            windows, it = [], iter(seen)
            for f in faces:
                windows.append(None if f is None else next(it))
            cases.append(dict(name=name, kwargs=kw, momentum=stub.momentum, fixed=stub.fixed_bounding_box, sizes=sizes,
                              faces=faces, windows=windows, face_check=check, face_scale=scales))
        # one pixel-valued case for the whole crop (window + bicubic resize + clip) at image_size 32
        seen.clear()
        _, stub = R.reference_wrapper_stub(32)
        g = torch.Generator().manual_seed(9)
        img = torch.rand(3, 150, 190, generator=g)
        faces = [np.array([40.0, 30.0, 130.0, 140.0]), np.array([120.0, -20.0, 230.0, 95.0])]
        crops, _, _ = m.InferenceWrapper.crop_image(stub, [img, img], faces)
    finally:
        m.F.interpolate = real_interp
    pixel = dict(image=img, faces=faces, crops=crops)

    from utils import point_transforms
    mix = []
    gm = torch.Generator().manual_seed(12)
    for B, T in ((1, 1), (1, 3), (2, 2)):
        srt = lambda n: (1 + 0.1 * torch.randn(n, 3, generator=gm), 0.4 * torch.randn(n, 3, generator=gm),
                         0.1 * torch.randn(n, 3, generator=gm))
        ths, tht = point_transforms.get_transform_matrix(*srt(B)), point_transforms.get_transform_matrix(*srt(B * T))
        for mix_old in (True, False):
            _, stub = R.reference_wrapper_stub(32, mix_old=mix_old)
            out = m.InferenceWrapper.get_mixing_theta(stub, ths, tht)
            mix.append(dict(source=ths, target=tht, mix_old=mix_old, out=out))
    det = []
    for _ in range(4):      # the bbox arithmetic of infer.py:385-391 evaluated literally
        xmin, ymin, wd, ht, W, H = rng.uniform(0.1, 0.5), rng.uniform(0.05, 0.5), rng.uniform(0.2, 0.5), rng.uniform(0.2, 0.6), 640, 480
        det.append(dict(rel=(xmin, ymin, wd, ht), size=(W, H),
                        face=np.array([W * xmin, H * ymin * 0.9, W * (xmin + wd), min(H * (ymin + ht * 1.2), H - 1)])))
    path = os.path.join(OUT, "hostglue.pt")
    torch.save(dict(crop_cases=cases, pixel=pixel, mixing=mix, detections=det), path)
    print("hostglue.pt", os.path.getsize(path) / 1e3, "KB;", sum(len(c["faces"]) for c in cases), "crop windows,", len(mix), "mixing cases")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    sampler_kat()
    pose_theta()
    tiny_hotpath()
    tiny_stage2()
    embedders_golden()
    hostglue_golden()
