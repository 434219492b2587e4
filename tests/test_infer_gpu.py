"""GPU tests of the drop-in boundary: InferenceWrapper with the reference's constructor / forward
signature, fed from an args.txt + checkpoint on disk, checked against the golden outputs of the real reference."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(golden_dir):
    return torch.load(os.path.join(golden_dir, "tiny_hotpath.pt"), weights_only=False)


@pytest.fixture(scope="module")
def project(tmp_path_factory, tiny):
    """<project>/<folder>/<exp>/args.txt (`key: value` lines as train.py:80-83 dumps them) + checkpoints/<file>"""
    from emoportraits_amd import config
    root = tmp_path_factory.mktemp("proj")
    exp = root / "logs" / "exp"
    (exp / "checkpoints").mkdir(parents=True)
    cfg = config.hot_path_config(overrides=tiny["cfg"])
    with open(exp / "args.txt", "wt") as f:
        for k, v in cfg.items():
            f.write(f"{k}: {v}\n")
        f.write("experiment_name: exp\nuse_seg: True\n")
    torch.save(tiny["state_dict"], exp / "checkpoints" / "model.pth")
    return root


def _wrapper(project, **kw):
    from notebooks.infer import InferenceWrapper
    return InferenceWrapper(experiment_name="exp", model_file_name="model.pth", project_dir=str(project), folder="logs",
                            print_params=False, **kw)


def test_source_then_driver_calls_match_reference_golden(project, tiny):
    w = _wrapper(project)
    S = tiny["cfg"]["image_size"]
    # source call: returns None (driver_image is None), caches the canonical volume like the reference
    r = w.forward(source_image=tiny["img"], crop=False, source_mask=torch.ones(1, 1, S, S),
                  custome_idt_embed=tiny["idt_embed"], custome_source_pose_embed=tiny["source_pose_embed"],
                  custome_source_theta_embed=tiny["theta_src"])
    assert r is None
    ref_c = tiny["source"]["canonical"]
    assert w.target_latent_volume.shape == ref_c.shape
    assert (w.target_latent_volume.cpu() - ref_c).abs().max().item() <= 1e-3 * ref_c.abs().max().item()
    for attr in ("idt_embed", "source_latent_volume", "target_latent_volume", "pred_source_theta",
                 "source_rotation_warp", "source_xy_warp_resize", "pred_source_pose_embed"):
        assert getattr(w, attr) is not None
    # driver-only call with the "emotion-driver" hooks (infer.py:565-566,603-604), one frame as the reference does
    w.target_latent_volume = ref_c.to(w.device)                 # continue from the reference's volume
    w._canonical_cl = w.hot_path.prepare_canonical(w.target_latent_volume)
    for i in range(2):
        imgs, t = w.forward(driver_image=None, crop=False, custome_target_pose_embed=tiny["target_pose_embed"][i:i + 1],
                            custome_target_theta_embed=tiny["theta_drv"][i:i + 1])
        assert len(imgs) == 1 and imgs[0].size == (S, S) and imgs[0].mode == "RGB"
        ref = tiny["driver"][i]["img"]
        assert t.shape == ref.shape
        assert (t.cpu() - ref).abs().max().item() <= 5e-3
        import numpy as np
        want = ref[0].clamp(0, 1).mul(255).byte().permute(1, 2, 0).numpy()
        assert np.abs(np.asarray(imgs[0]).astype(int) - want.astype(int)).max() <= 2
        assert w.pred_target_theta.shape == (1, 4, 4) and w.target_pose_embed.shape[0] == 1
    # batched extension: both frames in one call
    imgs, t = w.forward(crop=False, custome_target_pose_embed=tiny["target_pose_embed"],
                        custome_target_theta_embed=tiny["theta_drv"])
    assert len(imgs) == 2 and t.shape[0] == 2


def test_animate_streams_all_frames_in_order(project, tiny):
    w = _wrapper(project)
    S = tiny["cfg"]["image_size"]
    w.forward(source_image=tiny["img"], crop=False, source_mask=torch.ones(1, 1, S, S),
              custome_idt_embed=tiny["idt_embed"], custome_source_pose_embed=tiny["source_pose_embed"],
              custome_source_theta_embed=tiny["theta_src"])
    g = torch.Generator().manual_seed(0)
    N = 7
    pose = torch.randn(N, tiny["cfg"]["lpe_output_channels_expression"], generator=g)
    srt = (1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g), 0.05 * torch.randn(N, 3, generator=g))
    seen = []
    frames = {}
    for b0, u8 in w.animate(pose, srt, batch_size=3):
        assert u8.dtype == torch.uint8 and u8.shape[1:] == (S, S, 3)
        seen.append((b0, u8.shape[0]))
        for j in range(u8.shape[0]):
            frames[b0 + j] = u8[j].cpu()
    assert seen == [(0, 3), (3, 3), (6, 1)]
    # frame 4 alone equals frame 4 of the stream
    imgs, _ = w.forward(crop=False, custome_target_pose_embed=pose[4:5], custome_target_theta_embed=tuple(t[4:5] for t in srt))
    import numpy as np
    assert np.abs(np.asarray(imgs[0]).astype(int) - frames[4].numpy().astype(int)).max() <= 1


def test_strict_checkpoint_loading_and_loud_failures(project, tiny):
    bad = dict(tiny["state_dict"])
    bad.pop("decoder_nw.res_decoder.1.block.0.weight_u")
    with pytest.raises(KeyError, match="missing"):
        _wrapper(project, state_dict=bad)
    bad = dict(tiny["state_dict"])
    bad["decoder_nw.res_decoder.0.weight_orig"] = torch.zeros(3, 3, 1, 1)
    with pytest.raises(KeyError, match="shape mismatch"):
        _wrapper(project, state_dict=bad)
    with pytest.raises(RuntimeError, match="no CPU path"):
        _wrapper(project, use_gpu=False)
    w = _wrapper(project)
    with pytest.raises(RuntimeError, match="source_image first"):
        w.forward(crop=False, custome_target_pose_embed=tiny["target_pose_embed"], custome_target_theta_embed=tiny["theta_drv"])
    with pytest.raises(RuntimeError, match="face_detector"):
        w.forward(source_image=tiny["img"], crop=True)
    with pytest.raises(RuntimeError, match="idt_embedder"):
        w.forward(source_image=tiny["img"], crop=False, source_mask=torch.ones(1, 1, 64, 64))


def test_model_attribute_seam(project, tiny):
    """b2/b3: the reference reaches into self.model.<net>; the operator seam grid_sample(inputs, grid) is NCDHW"""
    import torch.nn.functional as F
    w = _wrapper(project)
    m = w.model
    c, d, s = (tiny["cfg"][k] for k in ("latent_volume_channels", "latent_volume_depth", "latent_volume_size"))
    g = torch.Generator().manual_seed(2)
    vol = torch.randn(1, c, d, s, s, generator=g)
    grid = torch.rand(1, d, s, s, 3, generator=g) * 2.2 - 1.1
    got = m.grid_sample(vol.to(w.device), grid.to(w.device))
    assert torch.equal(got.cpu(), F.grid_sample(vol, grid, padding_mode="zeros", align_corners=False))
    assert m.identity_grid_3d.shape == (1, d * s * s, 4)
    dd = {"idt_embed": tiny["idt_embed"].to(w.device), "source_pose_embed": tiny["source_pose_embed"].to(w.device),
          "target_pose_embed": tiny["target_pose_embed"][:1].to(w.device)}
    src_e, tgt_e, _, embed_dict = m.predict_embed(dd)
    assert embed_dict == {} and tgt_e["orig"].shape == tiny["driver"][0]["warp_embed"].shape
    warp, delta = m.uv_generator_nw(tgt_e)
    assert warp.shape == (1, d, s, s, 3) and delta.shape == (1, 3, d, s, s)
    assert (warp.cpu() - tiny["driver"][0]["uv_warp"]).abs().max().item() <= 2e-4
    feat = torch.randn(1, c * d, s, s, generator=g).to(w.device)
    img, seg, deep_f, img_f = m.decoder_nw({}, {}, feat, False, stage_two=True)
    assert seg is None and img.shape[1] == 3 and deep_f is not None and img_f is not None


def test_images_in_images_out_with_native_embedders(tmp_path, tiny):
    """with the embedder weights present (checkpoint keys + head_pose_regressor_path) the wrapper
    takes raw crops, as the reference does with crop=False (notebooks/infer.py:395-507, :546-644).  Checked against the
    oracle's composition of the same stages."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate as O
    from emoportraits_amd import config
    from emoportraits_amd import embedders as E
    ecfg = E.embedder_config(overrides=dict(idt_output_channels=tiny["cfg"]["gen_max_channels"],
                                            lpe_output_channels_expression=tiny["cfg"]["lpe_output_channels_expression"]))
    sd = {**tiny["state_dict"], **E.random_state_dict(E.idt_schema(ecfg), 1),
          **E.random_state_dict(E.expression_schema(ecfg), 2)}
    hp_sd = E.random_state_dict(E.head_pose_schema(), 3)
    hp_sd["fc.weight"] *= 0.05                                 # keep the random pose net near its bias ...
    hp_sd["fc.bias"] = torch.tensor([1.0, 1.0, 1.0, 0.1, -0.2, 0.05, 0.02, -0.03, 0.01])   # ... = a plausible head pose
    exp = tmp_path / "logs" / "exp"
    (exp / "checkpoints").mkdir(parents=True)
    torch.save(sd, exp / "checkpoints" / "model.pth")
    torch.save(hp_sd, tmp_path / "head_pose_regressor.pth")
    with open(exp / "args.txt", "wt") as f:
        for k, v in {**config.hot_path_config(overrides=tiny["cfg"]), **ecfg}.items():
            f.write(f"{k}: {v}\n")
        f.write(f"head_pose_regressor_path: {tmp_path / 'head_pose_regressor.pth'}\n")
    w = _wrapper(tmp_path)
    assert set(w.embedders) == {"idt_embedder", "expression_embedder", "head_pose_regressor"}
    S = tiny["cfg"]["image_size"]
    yy, xx = torch.meshgrid(torch.linspace(0, 1, S), torch.linspace(0, 1, S), indexing="ij")
    smooth = lambda a, b: torch.stack([0.5 + 0.4 * torch.sin(a * xx + b * yy), 0.5 + 0.4 * torch.cos(b * xx - a * yy),
                                       0.3 + 0.5 * xx * yy])[None]
    src, drv = smooth(5.0, 3.0), torch.cat([smooth(4.0, 6.0), smooth(2.0, 7.0)])
    mask = torch.ones(1, 1, S, S)
    assert w.forward(source_image=src, crop=False, source_mask=mask) is None
    imgs, t = w.forward(driver_image=drv, crop=False)
    assert len(imgs) == 2 and t.shape == (2, 3, S, S)
    cfg = tiny["cfg"]
    with torch.no_grad():
        idt = O.idt_embed(sd, "idt_embedder_nw", src)
        th_s = O.head_pose(hp_sd, src)["theta"]
        pe_s = O.expression_embed(sd, "expression_embedder_nw", src, th_s)["pose_embed"]
        canon = O.source_pass(sd, cfg, src, idt, pe_s, th_s)["canonical"]
        th_t = O.head_pose(hp_sd, drv)["theta"]
        pe_t = O.expression_embed(sd, "expression_embedder_nw", drv, th_t)["pose_embed"]
        ref = torch.cat([O.driver_pass(sd, cfg, canon, idt, pe_t[i:i + 1], th_t[i:i + 1])["img"] for i in range(2)])
    assert (w.idt_embed.cpu() - idt).abs().max().item() <= 2e-4 * idt.abs().max().item()
    assert (w.pred_target_theta.cpu() - th_t).abs().max().item() <= 2e-4 * th_t.abs().max().item()
    assert (w.target_pose_embed.cpu() - pe_t).abs().max().item() <= 1e-3 * pe_t.abs().max().item()
    assert (t.cpu() - ref).abs().max().item() <= 1e-2


def test_graph_replay_equals_eager_and_follows_a_new_identity(project, tiny):
    """hipGraph replay (emoportraits_amd/graphs.py) must be the same arithmetic as the eager launches, and a second source
    call must reach the captured sequence (the per-identity cache is updated in place)"""
    S = tiny["cfg"]["image_size"]
    src_kw = dict(crop=False, source_mask=torch.ones(1, 1, S, S), custome_idt_embed=tiny["idt_embed"],
                  custome_source_pose_embed=tiny["source_pose_embed"], custome_source_theta_embed=tiny["theta_src"])
    drv_kw = dict(crop=False, custome_target_pose_embed=tiny["target_pose_embed"], custome_target_theta_embed=tiny["theta_drv"])
    # (graphs are on by default -- first call of a signature eager, second captured, then replay; `eager` turns them off)
    eager, graphed = _wrapper(project, use_graphs=False), _wrapper(project)
    assert graphed.use_graphs and not eager.use_graphs and not eager._graphed
    outs = {}
    for name, w in (("eager", eager), ("graphed", graphed)):
        w.forward(source_image=tiny["img"], **src_kw)
        a = w.forward(**drv_kw)[1].clone()
        b = w.forward(**drv_kw)[1].clone()                       # second call: captured + replayed
        assert torch.equal(w.forward(**drv_kw)[1], b)            # third call: pure replay
        w.forward(source_image=tiny["img"].flip(-1), **src_kw)   # new identity
        c = w.forward(**drv_kw)[1].clone()
        outs[name] = (a, b, c)
    assert len(graphed._graphed["driver"].signatures()) == 1
    for x, y in zip(outs["eager"], outs["graphed"]):
        assert torch.equal(x, y)
    assert torch.equal(outs["graphed"][0], outs["graphed"][1])
    assert not torch.equal(outs["graphed"][0], outs["graphed"][2])


def test_crop_image_matches_reference_golden(project, golden_dir):
    """InferenceWrapper.crop_image (notebooks/infer.py:301-352): window arithmetic + in-place bicubic resize + clip against
    the crops the reference's own method produced (tests/golden/hostglue.pt)"""
    glue = torch.load(os.path.join(golden_dir, "hostglue.pt"), weights_only=False)["pixel"]
    w = _wrapper(project)
    w.cfg["image_size"] = glue["crops"].shape[-1]
    crops, check, scales = w.crop_image([glue["image"], glue["image"]], glue["faces"])
    assert check.all() and crops.shape == glue["crops"].shape
    assert (crops.cpu() - glue["crops"]).abs().max().item() <= 1e-5
    # a missing face gives a zero crop and a False flag, like the reference
    crops, check, scales = w.crop_image([glue["image"], glue["image"]], [None, glue["faces"][0]])
    assert list(check) == [False, True] and crops[0].abs().max().item() == 0 and scales[0] == 0


def test_crop_true_and_mix_true_paths(project, tiny):
    """crop=True with a face-detector callable (mediapipe's relative box in the reference) and mix=True (source stretch
    + driver rotation/translation, infer.py:568-569, 686-736) run end to end and equal the explicit formulation"""
    from emoportraits_amd import hostglue
    S = tiny["cfg"]["image_size"]
    kw = dict(custome_idt_embed=tiny["idt_embed"], custome_source_pose_embed=tiny["source_pose_embed"],
              custome_source_theta_embed=tiny["theta_src"])
    big = torch.rand(3, 2 * S, 3 * S, generator=torch.Generator().manual_seed(4))
    rel = (0.2, 0.25, 0.4, 0.45)
    w = _wrapper(project, embedders={"face_detector": lambda img: rel})
    w.forward(source_image=big, crop=True, source_mask=torch.ones(1, 1, S, S), **kw)
    face = hostglue.detection_to_face(*rel, 3 * S, 2 * S)
    want, _, _ = w.crop_image([big], [face])
    assert torch.equal(w.source_image_crop, want)
    drv_kw = dict(crop=False, custome_target_pose_embed=tiny["target_pose_embed"])
    # mix=True == passing the mixed pose explicitly
    th_t = tiny["theta_drv"].to(w.device)
    w.embedders["head_pose_regressor"] = lambda crop, srt=False: (th_t, None, None, None)
    dummy = torch.zeros(2, 3, S, S)
    _, mixed_img = w.forward(driver_image=dummy, mix=True, **drv_kw)
    mixed = torch.from_numpy(hostglue.mixing_theta(tiny["theta_src"].numpy(), tiny["theta_drv"].numpy(), True)).float()
    assert (w.pred_target_theta.cpu() - mixed).abs().max().item() <= 1e-6
    full = torch.cat([mixed, torch.tensor([[[0.0, 0, 0, 1]]]).expand(2, -1, -1)], dim=1)
    _, explicit_img = w.forward(custome_target_theta_embed=full, **drv_kw)
    assert torch.equal(mixed_img, explicit_img)


def _toy_embedders(tiny, device):
    """stand-ins for the driver-side networks (the tiny fixture has no embedder weights): deterministic functions of the
    crop, with the call signatures the wrapper uses for user-supplied callables"""
    from emoportraits_amd import ops
    E = tiny["cfg"]["lpe_output_channels_expression"]
    g = torch.Generator().manual_seed(9)
    proj = (torch.randn(3 * 16, E, generator=g) * 0.5).to(device)

    def head_pose(crop, return_srt=False):
        # (frame by frame: a batched reduction may split its work differently for another batch size, and the multi-rank tests
        # compare thetas of the same frame computed in batches of different sizes bit for bit)
        m = torch.stack([crop[i].mean(dim=(1, 2)) for i in range(crop.shape[0])])   # [B,3]
        scale, rot, trans = 1 + 0.1 * (m - 0.5), 0.6 * (m - 0.5), 0.1 * (m.flip(1) - 0.5)
        theta = ops.pose_theta(scale.contiguous(), rot.contiguous(), trans.contiguous())
        return (theta, scale, rot, trans) if return_srt else theta

    def expression(crop, theta):
        small = torch.nn.functional.adaptive_avg_pool2d(crop, 4).reshape(crop.shape[0], -1)
        return small @ proj, crop[:, :, ::2, ::2].contiguous()                # (pose embedding, "aligned" crop)

    return {"head_pose_regressor": head_pose, "expression_embedder": expression}


def test_animate_frames_is_device_resident_and_equals_forward(project, tiny):
    import numpy as np
    w = _wrapper(project)
    w.embedders.update(_toy_embedders(tiny, w.device))
    S = tiny["cfg"]["image_size"]
    w.forward(source_image=tiny["img"], crop=False, source_mask=torch.ones(1, 1, S, S),
              custome_idt_embed=tiny["idt_embed"], custome_source_pose_embed=tiny["source_pose_embed"],
              custome_source_theta_embed=tiny["theta_src"])
    g = torch.Generator().manual_seed(3)
    N = 8
    frames = (torch.rand(N, S, S, 3, generator=g) * 255).to(torch.uint8)
    got = {}
    order = []
    for b0, u8 in w.animate_frames(frames.pin_memory(), batch_size=3, ring=2):
        assert not u8.is_cuda and u8.dtype == torch.uint8 and u8.shape[1:] == (S, S, 3)
        order.append((b0, u8.shape[0]))
        for j in range(u8.shape[0]):
            got[b0 + j] = u8[j].clone()
    assert order == [(0, 3), (3, 3), (6, 2)]
    # the reference's per-call path on the same frames (float tensors in [0,1] = byte / 255)
    imgs, _ = w.forward(driver_image=frames.permute(0, 3, 1, 2).float() / 255.0, crop=False)
    assert w.target_img_align is not None and w.target_img_align.shape == (N, 3, S // 2, S // 2)    # infer.py:608
    for i in range(N):
        assert np.abs(np.asarray(imgs[i]).astype(int) - got[i].numpy().astype(int)).max() <= 1
    # crop windows are read in place from larger frames and resized on the device
    big = (torch.rand(2, 96, 128, 3, generator=g) * 255).to(torch.uint8)
    wins = [(10, 5, 80), (40, 16, 64)]
    dev_out = [u8 for _, u8 in w.animate_frames(big, batch_size=2, windows=wins, to_host=False)]
    assert dev_out[0].is_cuda and dev_out[0].shape == (2, S, S, 3)
    x = big.permute(0, 3, 1, 2).float() / 255.0
    crops = torch.cat([torch.nn.functional.interpolate(x[i:i + 1, :, y:y + s, xx:xx + s], size=(S, S), mode="bicubic").clamp(0, 1)
                       for i, (xx, y, s) in enumerate(wins)])
    imgs, _ = w.forward(driver_image=crops, crop=False)
    for i in range(2):
        assert np.abs(np.asarray(imgs[i]).astype(int) - dev_out[0][i].cpu().numpy().astype(int)).max() <= 2


def test_source_mask_semantics_follow_reference(project, tiny):
    """notebooks/infer.py:408-420: the face-parsing mask (> 0.6) always multiplies the crop; source_mask only replaces
    source_img_mask"""
    w = _wrapper(project)
    S = tiny["cfg"]["image_size"]
    g = torch.Generator().manual_seed(4)
    soft = torch.rand(1, 1, S, S, generator=g)
    w.embedders["face_parsing"] = lambda crop: soft.to(crop.device)
    matte = torch.rand(1, 1, S, S, generator=g)
    w.forward(source_image=tiny["img"], crop=False, source_mask=matte, custome_idt_embed=tiny["idt_embed"],
              custome_source_pose_embed=tiny["source_pose_embed"], custome_source_theta_embed=tiny["theta_src"])
    hard = (soft > 0.6).float()
    assert torch.equal(w.source_img_crop_m.cpu(), tiny["img"] * hard)
    assert torch.equal(w.source_img_mask.cpu(), matte)
    d, s = tiny["cfg"]["latent_volume_depth"], tiny["cfg"]["latent_volume_size"]
    assert w.source_rotation_warp.shape == (1, d, s, s, 3)                  # the cached rotation warp is a grid (infer.py:441-444)


def test_driver_call_without_a_frame_needs_the_theta_hook(project, tiny):
    w = _wrapper(project)
    S = tiny["cfg"]["image_size"]
    w.forward(source_image=tiny["img"], crop=False, source_mask=torch.ones(1, 1, S, S),
              custome_idt_embed=tiny["idt_embed"], custome_source_pose_embed=tiny["source_pose_embed"],
              custome_source_theta_embed=tiny["theta_src"])
    with pytest.raises(RuntimeError, match="custome_target_theta_embed"):
        w.forward(driver_image=None, crop=False, custome_target_pose_embed=tiny["target_pose_embed"][:1])
