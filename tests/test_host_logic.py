"""CPU tests of the host-side logic around the kernels: config contract, checkpoint schema, SN/WS folding, weight packing."""
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import restate as O  # noqa: E402

from emoportraits_amd import config, pack, random_init, schema  # noqa: E402


def test_released_config_matches_the_oracle_and_is_validated():
    cfg = config.hot_path_config()
    assert {k: cfg[k] for k in O.RELEASED_CFG} == O.RELEASED_CFG
    with pytest.raises(ValueError, match="norm_layer_type"):
        config.hot_path_config(overrides={"norm_layer_type": "bn"})
    with pytest.raises(ValueError, match="use_back"):
        config.hot_path_config(overrides={"use_back": True})
    with pytest.raises(ValueError, match="resize_warp"):
        config.hot_path_config(overrides={"warp_output_size": 128})
    for bad in (384, 768, 96):          # widths the conv kernels do not tile: refused at configuration time
        with pytest.raises(ValueError, match="image_size"):
            config.hot_path_config(overrides={"image_size": bad})
    for ok in (64, 128, 256, 512, 1024):
        assert config.hot_path_config(overrides={"image_size": ok})["image_size"] == ok


def test_args_txt_both_formats(tmp_path):
    # `key: value` dump (train.py:80-83 / utils/args.py:34-65 type sniffing)
    p = tmp_path / "args.txt"
    p.write_text("image_size: 256\nnorm_layer_type: gn\nuse_ws: True\nuse_sn: True\nenc_channel_mult: 4.0\n"
                 "experiment_name: foo: bar\nuse_back: False\n")
    found = config.parse_args_txt(p)
    assert found["image_size"] == 256 and found["use_ws"] is True and found["enc_channel_mult"] == 4.0
    assert found["experiment_name"] == "foo: bar" and found["use_back"] is False
    # launch-command form (experiments/args.txt of the reference)
    q = tmp_path / "cmd.txt"
    q.write_text("python3 -m torch.distributed.launch --nproc_per_node=8 ../train.py --image_size 512 --norm_layer_type gn "
                 "--use_ws True --dec_channel_mult 2 --im_dec_ch_div_factor 1.5 --use_back False")
    found = config.parse_args_txt(q)
    assert found["image_size"] == 512 and found["use_ws"] is True and found["im_dec_ch_div_factor"] == 1.5
    cfg = config.hot_path_config(found)
    assert cfg["image_size"] == 512 and cfg["dec_channel_mult"] == 2.0


def test_schema_matches_golden_reference_state_dict(golden_dir):
    tiny = torch.load(os.path.join(golden_dir, "tiny_hotpath.pt"), weights_only=False)
    cfg = config.hot_path_config(overrides=tiny["cfg"])
    want = schema.hot_path_schema(cfg)
    got = {k: tuple(v.shape) for k, v in tiny["state_dict"].items()}
    assert want == got
    sd = dict(tiny["state_dict"])
    sd["decoder_nw.extra.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="unexpected"):
        schema.check_state_dict(sd, cfg)
    # a 512-px checkpoint does not load into a 256-px model (layer names embed the resolution)
    cfg256 = config.hot_path_config(overrides={"image_size": 256})
    cfg512 = config.hot_path_config(overrides={"image_size": 512})
    k512 = set(schema.hot_path_schema(cfg512))
    assert "local_encoder_nw.from_rgb_512px.weight_orig" in k512
    with pytest.raises(KeyError):
        schema.check_state_dict({k: torch.zeros(s) for k, s in schema.hot_path_schema(cfg512).items()}, cfg256)


def test_random_state_dict_is_schema_complete_and_trained_like():
    cfg = config.hot_path_config(overrides={"image_size": 256})
    sd = random_init.random_state_dict(cfg, seed=1, with_source=False)
    assert schema.check_state_dict(sd, cfg, with_source=False)
    # u, v at the dominant singular pair => folded weight has spectral norm ~1
    w = pack.fold_sn(sd["decoder_nw.res_decoder.1.block.0.weight_orig"], sd["decoder_nw.res_decoder.1.block.0.weight_u"],
                     sd["decoder_nw.res_decoder.1.block.0.weight_v"])
    s = torch.linalg.svdvals(w.reshape(w.shape[0], -1))[0].item()
    assert 0.9 < s < 1.2


def test_folding_equals_the_oracle_restatement(golden_dir):
    tiny = torch.load(os.path.join(golden_dir, "tiny_hotpath.pt"), weights_only=False)
    sd = tiny["state_dict"]
    p = "decoder_nw.res_decoder.1.block.0"
    assert torch.equal(pack.folded_conv(sd, p, "sn")[0], O.sn_weight(sd, p))
    p = "decoder_nw.res_decoder.1.block_feats.2"
    assert torch.equal(pack.folded_conv(sd, p, "ws")[0], O.ws_weight(sd[p + ".weight"]))


@pytest.mark.parametrize("cfg_id", [0, 1, 2])
@pytest.mark.parametrize("shape", [(5, 6, 3, 3), (40, 9, 3, 3, 3), (33, 20, 1, 1), (130, 3, 7, 1, 7)])
def test_pack_weight_layout(cfg_id, shape):
    """every weight element lands at [co_tile][chunk][kd][pair][tap][half][BM] and padding is zero"""
    if shape[-1] == 7 and cfg_id == 2:
        pytest.skip("1x7 taps have no 32-row config")
    w = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    flat = pack.pack_weight(w, cfg_id)
    w5 = w if w.dim() == 5 else w.unsqueeze(2)
    cout, cin, kd, kh, kw = w5.shape
    bm, kc = pack.conv_pack_info(kh, kw, cfg_id)
    ncc = -(-cin // kc)
    taps = kh * kw
    assert flat.numel() == (-(-cout // bm)) * bm * ncc * kc * kd * taps
    assert abs(flat.abs().sum().item() - w.abs().sum().item()) < 1e-3 * w.abs().sum().item()
    g = torch.Generator().manual_seed(1)
    for _ in range(50):
        co, ci = int(torch.randint(cout, (1,), generator=g)), int(torch.randint(cin, (1,), generator=g))
        t, tap = int(torch.randint(kd, (1,), generator=g)), int(torch.randint(taps, (1,), generator=g))
        cot, i = divmod(co, bm)
        cc, cl = divmod(ci, kc)
        pair, half = divmod(cl, 2)
        idx = (((((cot * ncc + cc) * kd + t) * (kc // 2) + pair) * taps + tap) * 2 + half) * bm + i
        assert flat[idx].item() == w5[co, ci, t, tap // kw, tap % kw].item()


def test_pack_weight_layout_of_the_pointwise_fp16_split():
    """pack.pack_weight_f16x2_1x1 (csrc/conv_igemm_f16x2_p1.h): [channel tile, even count][Cin chunk of 32][plane][k-step of 16]
    [half][BM][8]; the two planes are the fp16 pair of w * w_scale; padding (channels, the tile that fills an odd count) is zero"""
    g = torch.Generator().manual_seed(7)
    cout, cin = 320, 96
    w = torch.randn(cout, cin, 1, 1, generator=g)
    flat, ws = pack.pack_weight_f16x2_1x1(w)
    ncot, ncc = 6, 3
    assert flat.dtype == torch.float16 and flat.numel() == ncot * ncc * 2 * 2 * 2 * 64 * 8 and 512 <= w.abs().max().item() * ws < 1024
    v = flat.view(ncot, ncc, 2, 2, 2, 64, 8)
    wp = torch.zeros(ncot * 64, ncc * 32)
    wp[:cout, :cin] = w.view(cout, cin) * ws
    w1 = wp.to(torch.float16)
    planes = (w1, (wp - w1.float()).to(torch.float16))
    assert torch.equal((planes[0].float() + planes[1].float())[:cout, :cin], (w.view(cout, cin) * ws)) or \
        ((planes[0].float() + planes[1].float())[:cout, :cin] - w.view(cout, cin) * ws).abs().max().item() <= 2.0 ** -13
    for _ in range(500):
        co, ci, pl = (int(torch.randint(n, (1,), generator=g)) for n in (ncot * 64, ncc * 32, 2))
        cot, m = divmod(co, 64)
        cc, r = divmod(ci, 32)
        ks, r2 = divmod(r, 16)
        half, k8 = divmod(r2, 8)
        assert v[cot, cc, pl, ks, half, m, k8] == planes[pl][co, ci]
    assert v[5].abs().sum() == 0                                         # the padding tile of the odd count
    # which layers: whole channel tiles, at least one pair, an even number of 32-channel stages
    ok = pack.supports_f16x2_pointwise
    assert ok(512, 1536, 1, 1, 1) and ok(320, 512, 1, 1, 1) and ok(192, 320, 1, 1, 1) and ok(128, 192, 1, 1, 1)
    assert not ok(64, 128, 1, 1, 1) and not ok(128, 96, 1, 1, 1) and not ok(128, 128, 1, 3, 3) and not ok(130, 128, 1, 1, 1)


def test_launch_config_heuristic():
    assert pack.choose_cfg(512) == pack.CFG_B and pack.choose_cfg(320) == pack.CFG_B and pack.choose_cfg(3) == pack.CFG_C
    # few position tiles: prefer more (smaller) blocks; many: least padding wins
    assert pack.choose_cfg_for_launch(256, 16) == pack.CFG_C
    assert pack.choose_cfg_for_launch(320, 4096) == pack.CFG_B
    # equal padding: the 64-row tile (5 blocks per CU) is preferred to the 128-row one (3 per CU), measured faster or equal
    assert pack.choose_cfg_for_launch(128, 32768) == pack.CFG_B
    # K split: big launches stay single-pass; a 64x64 map at batch 1 (32 position tiles) splits
    assert pack.plan_launch(128, 128, 1, 3, 3, 32768) == (pack.CFG_B, 1)
    cfg, ks = pack.plan_launch(512, 512, 1, 3, 3, 32)
    assert ks > 1 and (-(-512 // pack._BM[cfg])) * 32 * ks >= 512
    assert pack.plan_launch(256, 512, 3, 3, 3, 4)[1] > 1                 # 8^3 WarpGenerator layer, batch 1
    assert pack.ksplit_for(4, 7) == 1                                   # never fewer than 8 stages per split


# ---- f1: embedders ---------------------------------------------------------------------------------------------------
def test_embedder_schema_follows_the_reference_wrapping_rules():
    from emoportraits_amd import embedders as E
    cfg = E.embedder_config()
    idt = E.idt_schema(cfg)
    # Bottleneck: conv1 keeps spectral norm, conv2 / conv3 are weight-standardised (utils.py:1061-1096)
    assert "idt_embedder_nw.net.layer1.0.conv1.weight_orig" in idt
    assert "idt_embedder_nw.net.layer1.0.conv2.bias" in idt and "idt_embedder_nw.net.layer1.0.conv3.bias" in idt
    assert "idt_embedder_nw.net.layer1.0.downsample.0.weight_u" in idt
    assert idt["idt_embedder_nw.net.fc.weight_orig"] == (512, 2048, 1, 1)
    assert not any("running_mean" in k for k in idt)          # norm_layer_type gn
    ex = E.expression_schema(cfg)
    assert "expression_embedder_nw.net_face.net.layer2.0.conv1.weight_orig" in ex          # BasicBlock.conv1: SN
    assert "expression_embedder_nw.net_face.net.layer2.0.conv2.bias" in ex                 # BasicBlock.conv2: WS
    assert ex["expression_embedder_nw.net_face.pose_head.weight_orig"] == (128, 128 * 16)
    hp = E.head_pose_schema()
    assert hp["fc.weight"] == (9, 512) and "bn1.running_var" in hp and "conv1.weight" in hp and "conv1.bias" not in hp
    # bn variant: no WS replacement (the rule keys on GroupNorm siblings)
    bn = E.idt_schema(E.embedder_config(overrides=dict(norm_layer_type="bn")))
    assert "idt_embedder_nw.net.layer1.0.conv2.weight_orig" in bn and "idt_embedder_nw.net.bn1.running_mean" in bn
    with pytest.raises(ValueError):
        E.embedder_config(overrides=dict(lpe_final_pooling_type="transformer"))
    sd = E.random_state_dict(hp, 0)
    assert E.check_state_dict(sd, hp, "")
    sd.pop("fc.bias")
    with pytest.raises(KeyError):
        E.check_state_dict(sd, hp, "")


def test_pack_generic_layout():
    w = torch.arange(5 * 3 * 2 * 2, dtype=torch.float32).reshape(5, 3, 2, 2)
    wt = pack.pack_generic(w)
    assert wt.shape == (12, 64)
    assert torch.equal(wt[:, :5], w.reshape(5, 12).t()) and wt[:, 5:].abs().sum() == 0


# ---- f4: wrapper host glue pinned to the reference's own methods ----------------------------------------------------
@pytest.fixture(scope="module")
def glue(golden_dir):
    return torch.load(os.path.join(golden_dir, "hostglue.pt"), weights_only=False)


def test_crop_windows_match_reference_crop_image(glue):
    """every window InferenceWrapper.crop_image (notebooks/infer.py:301-352) cut -- overflowing boxes, missing faces,
    the `scale` argument, the smoothed / fixed bounding-box state -- recovered from coordinate-coded images"""
    from emoportraits_amd import hostglue as H
    n = 0
    for case in glue["crop_cases"]:
        kw = case["kwargs"]
        tracker = H.CropTracker(case["momentum"], case["fixed"]) if kw.get("use_smoothed_crop") else None
        for (h, w), face, want, scale_ref, ok in zip(case["sizes"], case["faces"], case["windows"], case["face_scale"],
                                                     case["face_check"]):
            got = H.crop_window(face, w, h, tracker, kw.get("scale", 1))
            if want is None:
                assert got is None and not ok and scale_ref == 0
                continue
            x_lo, y_lo, side, face_scale = got
            assert (x_lo, y_lo, side, side) == tuple(want), (case["name"], got, want)
            assert face_scale == scale_ref
            n += 1
    assert n == 34


def test_detection_box_and_mixing_theta_match_reference(glue):
    import numpy as np
    from emoportraits_amd import hostglue as H
    for d in glue["detections"]:
        assert np.array_equal(H.detection_to_face(*d["rel"], *d["size"]), d["face"])
    for m in glue["mixing"]:
        got = H.mixing_theta(m["source"].numpy(), m["target"].numpy(), m["mix_old"])
        assert got.shape == tuple(m["out"].shape)
        assert np.abs(got.astype(np.float32) - m["out"].numpy()).max() <= 1e-6


def test_fp16_plan_falls_back_to_fp32_where_the_fp16_kernel_cannot_run():
    """PackedConv(precision="f16").plan_for: the fp16-operand kernel needs a 64 x 256 output tile, a 16-byte aligned input and --
    with a fused scale / shift -- at most 1024 input channels (its LDS tables); everything else takes the exact-fp32 kernel"""
    import torch
    from emoportraits_amd import pack
    lay = pack.PackedConv("p", torch.zeros(64, 64, 3, 3), None, "cpu", precision="f16")
    assert lay.plan_for(32, 64, 64)[2] == "f16"
    assert lay.plan_for(32, 64, 64, aligned16=False)[2] == "f32"
    assert lay.plan_for(2, 16, 16)[2] == "f32"                      # no 256-pixel tile shape fits a 16 x 16 plane
    assert lay.plan_for(32, 64, 64, affine=True)[2] == "f16"
    wide = pack.PackedConv("w", torch.zeros(64, 1536, 1, 1), None, "cpu", precision="f16")
    assert wide.plan_for(32, 64, 64)[2] == "f16"                    # identity tables wrap: no limit without an affine
    assert wide.plan_for(32, 64, 64, affine=True)[2] == "f32"
    assert pack.F16_AFFINE_MAX_CIN == 1024
    assert pack.PackedConv("f", torch.zeros(64, 64, 3, 3), None, "cpu").plan_for(32, 64, 64)[2] == "f32"
    # round 6: in the decoders' launch form (3x3, whole 64-channel tiles, 4 x 64 position tiles, no activation, aligned out / res,
    # two pair items per CU) the layer runs the eight-wave two-tile kernel with plain fp16 operands; every other form as before
    big = 1 << 14                                                    # 128-position tiles: 8192 pair items per channel-tile pair
    two = pack.PackedConv("p2", torch.zeros(128, 64, 3, 3), None, "cpu", precision="f16")
    assert two.plan_for(big, 512, 512) == (pack.CFG_D, 1, "f16w8")
    assert two.plan_for(big, 512, 512, act="tanh")[2] == "f16" and two.plan_for(big, 512, 512, io_aligned16=False)[2] == "f16"
    assert two.plan_for(big, 32, 32)[2] == "f16"                     # 8 x 32 position tiles: the older kernel
    assert lay.plan_for(big, 512, 512)[2] == "f16"                   # an odd number of channel tiles (one): the older kernel
    assert pack.PackedConv("c", torch.zeros(96, 64, 3, 3), None, "cpu", precision="f16").plan_for(big, 512, 512)[2] == "f16"
    # ... an odd count >= 3: the pairs on the eight-wave kernel, the last tile on the older one (emo_conv_igemm_f16w8_rest, ABI 10) where
    # the plane is in BOTH kernels' launch form; elsewhere (planner logic only: the C entry points admit widths that are multiples of
    # 128, or 64 / 32 / 16 / 8) a half-empty last pair from five tiles on
    t3 = pack.PackedConv("t3", torch.zeros(192, 64, 3, 3), None, "cpu", precision="f16")
    t5 = pack.PackedConv("t5", torch.zeros(320, 64, 3, 3), None, "cpu", precision="f16")
    assert t3.plan_for(big, 512, 512)[2] == "f16w8" and pack.f16w8_rest_fits(192, 512, 512)
    assert t5.plan_for(big, 512, 512)[2] == "f16w8" and pack.f16w8_rest_fits(320, 512, 512)
    assert not pack.f16w8_rest_fits(192, 192, 192) and not pack.f16w8_rest_fits(128, 512, 512) and not pack.f16w8_rest_fits(64, 512, 512)
    assert t3.plan_for(big, 192, 192)[2] == "f32" and t5.plan_for(big, 192, 192)[2] == "f16w8"
    flat, ws = pack.pack_weight_f16w8(torch.randn(70, 24, 3, 3))
    assert flat.dtype == torch.float16 and flat.numel() == 2 * 2 * 9 * 2 * 64 * 8 and ws == 2.0 ** math.floor(math.log2(ws))


# ---- round-3 host logic ----------------------------------------------------------------------------------------------
def test_trained_like_checkpoint_scales_only_the_head_tensors():
    """trained_like_state_dict = random_state_dict with the image-head and warp-head gains; every other tensor identical
    (the gains keep the predicted warp within a voxel and the sigmoid unsaturated -- the regime of a trained network)"""
    cfg = config.hot_path_config(overrides={"image_size": 256})
    raw = random_init.random_state_dict(cfg, seed=3, with_source=False)
    tl = random_init.trained_like_state_dict(cfg, seed=3, with_source=False)
    assert schema.check_state_dict(tl, cfg, with_source=False) and set(raw) == set(tl)
    changed = sorted(k for k in raw if not torch.equal(raw[k], tl[k]))
    assert changed, "gains had no effect"
    assert all((".pre_head." in k or ".head." in k or "dec_img_head" in k) for k in changed), changed
    warp = [k for k in changed if k.startswith(("xy_generator_nw", "uv_generator_nw"))]
    assert warp and all(tl[k].abs().max() <= raw[k].abs().max() for k in warp)


def test_tile_tuning_word_round_trips_the_header_fields():
    """ops.tile_variant packs what gs3d_tile_launch.h unpacks: log2 tile extents, channel units per block, LDS KiB, 512 flag"""
    from emoportraits_amd import ops
    v = ops.tile_variant(tile=(4, 8, 8), units_per_block=24, lds_kib=96, threads=512)
    assert (v & 15, (v >> 4) & 15, (v >> 8) & 15) == (2, 3, 3)
    assert (v >> 12) & 31 == 24 and (v >> 17) & 255 == 96 and (v >> 25) & 1 == 1
    assert v < ops.TILE and ops.tile_variant() == 0
    assert (ops.tile_variant(tile=(16, 4, 4)) >> 25) & 1 == 0


def test_planner_quantisation_term():
    """blocks / CU rounding: 640 blocks occupy 2.5 of 3 rounds; exact multiples and sub-machine launches are not penalised"""
    assert pack._quantisation(640) == pytest.approx(2.5 / 3)
    assert pack._quantisation(1280) == 1.0 and pack._quantisation(256) == 1.0 and pack._quantisation(100) == 1.0
    assert pack._quantisation(257) == pytest.approx(257 / 512)
    # batch-2 320 -> 320 at 128^2: the planner leaves the 640-block tiling (measured 113 TF) for the 1280-block one (130 TF)
    allowed = (pack.CFG_A, pack.CFG_B, pack.CFG_C, pack.CFG_D)
    cfg, ks = pack.plan_launch(320, 320, 1, 3, 3, 2 * 128 * 128 // 128, allowed)
    assert (-(-320 // pack._BM[cfg])) * (2 * 128 * 128 // pack._BP[cfg]) * ks % 256 == 0


def test_bf16x3_split_is_exact_and_the_packed_layout_is_the_documented_one():
    """pack.split_bf16x3: h + m + l == w bit for bit; pack.pack_weight_bf16x3: element (co, ci, kd, r, s) of plane p lands at
    [co_tile][ci/16][kd][r][p][s][half][co%64][ci%8] (include/emo_hip.h, emo_conv_igemm_bf16x3)"""
    g = torch.Generator().manual_seed(2)
    w = torch.randn(70, 24, 3, 3, 3, generator=g) * torch.logspace(-6, 3, 70).view(70, 1, 1, 1, 1)
    h, m, l = pack.split_bf16x3(w)
    assert torch.equal(h.float() + m.float() + l.float(), w)
    assert (m.float().abs() <= h.float().abs() * 2.0 ** -8 + 1e-45).all() and (l.float().abs() <= h.float().abs() * 2.0 ** -16 + 1e-45).all()
    flat = pack.pack_weight_bf16x3(w)
    cout, cin, kd = 70, 24, 3
    n_cot, n_cc = 2, 2
    assert flat.dtype == torch.bfloat16 and flat.numel() == n_cot * n_cc * kd * 3 * 3 * 3 * 2 * 64 * 8
    planes = (h, m, l)
    for _ in range(200):
        co, ci, t, r, s, p = (int(torch.randint(n, (1,), generator=g)) for n in (cout, cin, kd, 3, 3, 3))
        cot, i = divmod(co, 64)
        cc, cl = divmod(ci, 16)
        half, k8 = divmod(cl, 8)
        idx = ((((((((cot * n_cc + cc) * kd + t) * 3 + r) * 3 + p) * 3 + s) * 2 + half) * 64 + i) * 8 + k8)
        assert flat[idx].item() == planes[p][co, ci, t, r, s].item()
    # padding (channels 70..127, 24..31) is zero
    assert flat.float().abs().sum().item() == pytest.approx(sum(t.float().abs().sum().item() for t in planes), rel=1e-3)
    import ctypes
    from emoportraits_amd import hip
    lib = hip.load()
    bm, kc = ctypes.c_int(), ctypes.c_int()
    assert lib.emo_conv_pack_info_bf16x3(3, 3, 3, ctypes.byref(bm), ctypes.byref(kc)) == 0
    assert (bm.value, kc.value) == (pack.BF16X3_BM, pack.BF16X3_KC)
    assert lib.emo_conv_pack_info_bf16x3(1, 1, 3, ctypes.byref(bm), ctypes.byref(kc)) != 0


def test_f16x2_weight_split_and_scale():
    """pack.pack_weight_f16x2: w * w_scale = w1 + w2 to 2^-24 relative (2^-25 absolute below 0.25), max|w| * w_scale in
    [512, 1024), planes laid out like the bf16x3 tensor with two planes"""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(70, 24, 3, 3, generator=g) * 0.03
    flat, sc = pack.pack_weight_f16x2(w)
    assert flat.dtype == torch.float16 and flat.numel() == 2 * 2 * 1 * 3 * 2 * 3 * 2 * 64 * 8
    assert 512 <= w.abs().max().item() * sc < 1024 and math.log2(sc) == int(math.log2(sc))
    planes = flat.view(2, 2, 1, 3, 2, 3, 2, 64, 8).float()      # [cot, cc, kd, r, plane, s, half, co, k8]
    back = (planes[:, :, :, :, 0] + planes[:, :, :, :, 1])      # [cot, cc, kd, r, s, half, co, k8]
    full = back.permute(0, 6, 1, 5, 7, 2, 3, 4).reshape(128, 32, 1, 3, 3)[:70, :24, 0]
    err = (full / sc - w).abs()
    assert (err <= w.abs() * 2.0 ** -23 + 2.0 ** -25 / sc).all()


def test_split_convolution_arithmetic_on_the_cpu():
    """the claims csrc/conv_igemm_bf16x3.h makes about its arithmetic, checked without a GPU (tools/split_accuracy.py): products
    of bf16 terms are exact in fp32; the six-product bf16 split is closer to an fp64 convolution than an fp32 convolution is;
    three bf16 products are not (2^-16); the three-product fp16 split of the scaled operands is at the fp32 convolution's level"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(1, 64, 24, 24, generator=g) * 3 + 0.5)
    w = torch.randn(32, 64, 3, 3, generator=g) / (64 * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    scale = ref.abs().mean()
    err = lambda y: ((y.double() - ref).abs().mean() / scale).item()
    xs = [t.float() for t in pack.split_bf16x3(x)]
    ws = [t.float() for t in pack.split_bf16x3(w)]
    a, b = xs[1][0, 0, 0, :8], ws[0][0, 0, 0, :3]
    assert torch.equal((a[:3] * b).double(), a[:3].double() * b.double())                 # 8 x 8 significand bits fit fp32
    c64 = lambda p, q: F.conv2d(p.double(), q.double(), padding=1)
    six = sum(c64(xs[i], ws[j]) for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)))
    three = sum(c64(xs[i], ws[j]) for i, j in ((1, 0), (0, 1), (0, 0)))
    e32 = err(F.conv2d(x, w, padding=1))
    assert err(six) < 0.2 * e32 and err(three) > 5 * e32
    sx = pack.F16X2_IN_SCALE
    flat, sw = pack.pack_weight_f16x2(w)
    x1 = (x * sx).to(torch.float16).float()
    x2 = (x * sx - x1).to(torch.float16).float()
    w1 = (w * sw).to(torch.float16).float()
    w2 = (w * sw - w1).to(torch.float16).float()
    f16x2 = (c64(x1, w1) + c64(x1, w2) + c64(x2, w1)) / (sx * sw)
    assert err(f16x2) < e32


@pytest.mark.parametrize("mode,expect", [("bf16x3", {"emo_conv_igemm_bf16x3": 28, "emo_conv_igemm_f32": 13}),
                                          ("f32", {"emo_conv_igemm_f32": 41}),
                                          # (every fp16-split launch is followed by its guarded bf16x3 recomputation launch)
                                          # (... and the fp16 split also takes the two 32-channel 3-D layers of the WarpGenerator)
                                          # (... and, since round 5, the decoder's four 1x1 layers -- 1536 -> 512 and the skips of its
                                          # up-blocks -- on the pointwise kernel, each followed by its guarded fp32 MFMA launch)
                                          # (... and the 3-channel warp head on the 32-row channel tile of the fp16 split)
                                          # (the image head, 128 -> 3 at the output resolution, is a stream in every mode:
                                          # emo_conv_head_f32, one call)
                                          (None, {"emo_conv_igemm_f16x2": 35, "emo_conv_igemm_bf16x3": 31, "emo_conv_igemm_f32": 6,
                                                  "emo_conv_igemm_f32_guarded": 4})])
def test_driver_pass_host_side_against_a_stub_library(monkeypatch, mode, expect):
    """the host side of the released R512 driver pass without a GPU (tools/host_overhead.py: every kernel entry point of the
    library returns at once): the launch plan sends the 28 3x3 / 3x3x3 layers the split kernel covers to it (30 in the fp16 split) (default mode: as the
    fp16 split, each launch followed by its guarded bf16x3 launch), the 1x1 / narrow 3-D / head convolutions to the fp32 MFMA
    kernel (default mode: the decoder's four 1x1 layers on the pointwise split kernel), and the whole pass is 95 C-ABI calls
    (+ 35 guards)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import host_overhead
    from emoportraits_amd import nets
    monkeypatch.delenv("EMO_CONV_PRECISION", raising=False)
    stub = host_overhead.install_stub(monkeypatch.setattr)
    cfg = config.hot_path_config(overrides={"image_size": 512})
    sd = random_init.trained_like_state_dict(cfg, seed=0, with_source=False)
    hp = nets.HotPath(sd, cfg, "cpu", with_source=False, precision=mode)
    assert hp.precision == (mode or nets.DEFAULT_PRECISION) and nets.DEFAULT_PRECISION == "f16x2"
    B = 16
    ccl = hp.prepare_canonical(torch.empty(1, 96, 16, 64, 64))
    stub.calls.clear()
    img = hp.driver_pass(ccl, torch.randn(1, 512, 4, 4), torch.randn(B, 128), torch.eye(4)[None].repeat(B, 1, 1).contiguous())
    assert tuple(img.shape) == (B, 3, 512, 512)
    convs = {k: v for k, v in stub.calls.items() if k.startswith("emo_conv_igemm")}
    assert convs == expect, convs
    # the image head as a stream; the WarpGenerator's four upsamplings leave the GroupNorm sums of their outputs behind
    assert stub.calls["emo_conv_head_f32"] == 1 and stub.calls["emo_upsample_trilinear_gn_sums_f32"] == 4
    assert stub.calls["emo_groupnorm_affine_from_sums_f32"] == 4 and "emo_upsample_trilinear_f32" not in stub.calls
    assert sum(stub.calls.values()) == 95 + (35 if mode is None else 0), dict(stub.calls)


def test_isa_audit_finds_a_scalar_operand_read_too_early_and_an_in_flight_destination():
    """tools/kernel_resources.py --audit on synthetic listings: (1) a vector-ALU reload of a spilled scalar right in front of an
    asm load that reads it (CDNA3: five wait states) is reported, the same with the wait states in between is not; (2) an
    instruction that touches the destination of a pinned load before a vmcnt covers it is reported"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as K
    bad = ["\tv_readlane_b32 s85, v253, 7", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]
    good = ["\tv_readlane_b32 s85, v253, 7", "\t;;#ASMSTART", "\ts_nop 4", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]
    other = ["\tv_readlane_b32 s80, v253, 7", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]
    salu = ["\ts_mov_b32 s85, s3", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]   # SALU writes are interlocked
    assert len(K.sgpr_hazards(bad)) == 1 and K.sgpr_hazards(bad)[0][2] == [85]
    assert K.sgpr_hazards(good) == [] and K.sgpr_hazards(other) == [] and K.sgpr_hazards(salu) == []
    rsrc = ["\tv_readfirstlane_b32 s73, v9", "\tv_add_u32_e32 v1, v2, v3", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]
    assert len(K.sgpr_hazards(rsrc)) == 1                                  # (a register of the resource tuple, one wait state before)
    # any vector-ALU instruction with a scalar destination is a writer: the carry of v_add_co (second operand), a compare
    carry = ["\tv_add_co_u32_e64 v4, s[84:85], v2, v3", "\t;;#ASMSTART", "\tglobal_load_lds_dwordx4 v9, s[84:85]", "\t;;#ASMEND"]
    cmp_ = ["\tv_cmp_lt_u32_e64 s[72:73], v2, v3", "\ts_nop 1", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]
    src_only = ["\tv_add_co_u32_e64 v4, s[10:11], s85, v3", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND"]
    assert len(K.sgpr_hazards(carry)) == 1 and K.sgpr_hazards(carry)[0][2] == [84, 85]
    assert len(K.sgpr_hazards(cmp_)) == 1 and K.sgpr_hazards(src_only) == []   # (a scalar SOURCE of a vector-ALU op is no write)
    loop = ["\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[46:49], v2, s[72:75], s85 offen", "\t;;#ASMEND",
            "\tv_mov_b32_e32 v100, v47", "\ts_waitcnt vmcnt(0)", "\tv_mov_b32_e32 v101, v47",
            "\tv_mfma_f32_32x32x16_f16 a[0:15], v[4:7], v[8:11], a[0:15]", "\ts_endpgm"]
    hits = K.inflight_reads(loop)
    assert len(hits) == 1 and "v100" in hits[0][1] and hits[0][2] == [47]


def test_smooth_pose_scan_is_the_reference_loop_bit_for_bit():
    """hostglue.ema_scan against a literal replay of notebooks/infer.py:571-581 in torch (`self.theta = pred[i] * m + self.theta *
    (1 - m)` per frame, state carried between calls); also: scanning a clip in two chunks with the carried state equals scanning
    it at once -- which is what lets animate_frames() run the scan per chunk before it shards the frames"""
    import numpy as np
    from emoportraits_amd import hostglue
    g = torch.Generator().manual_seed(12)
    pred = torch.randn(37, 4, 4, generator=g)
    for m in (0.5, 0.3, 0.9, 0.01):
        theta, want = None, []
        for lo, hi in ((0, 11), (11, 12), (12, 37)):                # three "calls" of forward(smooth_pose=True)
            if theta is None:
                theta = pred[lo].clone()
            for i in range(lo, hi):
                theta = pred[i] * m + theta * (1 - m)
                want.append(theta.clone())
        want = torch.stack(want)
        got, state = hostglue.ema_scan(pred.numpy(), None, m)
        assert got.dtype == np.float32 and np.array_equal(got, want.numpy()) and np.array_equal(state, want[-1].numpy())
        a, st = hostglue.ema_scan(pred[:20].numpy(), None, m)
        b, st = hostglue.ema_scan(pred[20:].numpy(), st, m)
        assert np.array_equal(np.concatenate([a, b]), got)


def test_planner_mirrors_the_c_launchers():
    """round-5 advisor findings: (1) the planner's fill targets come from the library's CU count (256 without a device), not from
    a constant; (2) a layer with <= 32 output channels keeps the fp16 split on maps its 32-row tile does not cover (a 64-row packing
    made on first use) instead of dropping to the fp32 MFMA kernel; (3) the pointwise launch form refuses what the C launcher
    refuses (more than 2^23 positions per sample) instead of planning a launch that fails; (4) EMO_F16X2_POINTWISE=0 also holds
    for an explicitly requested precision"""
    from emoportraits_amd import hip
    assert pack.cu_count() == hip.load().emo_device_cu_count() and pack.cu_count() % 8 == 0
    assert pack._fill_blocks() == 2 * pack.cu_count()
    w = torch.randn(32, 64, 3, 3, 3) / 40
    layer = pack.PackedConv("l32", w, None, "cpu", precision="f16x2")
    assert layer.plan_for(4096, 64, 64) [::2] == (pack.CFG_F, "f16x2")             # 4 x 64 tiles: the 32-row channel tile
    assert layer.plan_for(256, 32, 32)[::2] == (pack.CFG_D, "f16x2")               # 8 x 32 tiles: the half-empty 64-row tile
    assert layer.plan_for(4096, 64, 64, ups=True)[::2] == (pack.CFG_D, "f16x2")
    n32, n64 = layer.packed(pack.CFG_F, "f16x2").numel(), layer.packed(pack.CFG_D, "f16x2").numel()
    assert n64 == 2 * n32 == 2 * 64 * 64 * 27                                      # two fp16 planes, 64 rows (half of them zero)
    assert torch.equal(layer.packed(pack.CFG_D, "f16x2").view(4, 3, 3, 2, 3, 2, 64, 8)[..., :32, :].reshape(-1).float().abs().sum(),
                       layer.packed(pack.CFG_F, "f16x2").float().abs().sum())
    assert pack.f16x2_pointwise_launch_fits(512, 512, False, 1 << 16, 512, positions_per_sample=1 << 18)
    assert not pack.f16x2_pointwise_launch_fits(4096, 4096, False, 1 << 20, 512, positions_per_sample=(1 << 23) + 64)
    pw = pack.PackedConv("pw", torch.randn(256, 64, 1, 1) / 8, None, "cpu", precision="f16x2")
    assert pw.pointwise_split and pw.plan_for(1 << 14, 512, 512, in_elems_per_sample=64 * (1 << 18))[2] == "f16x2"
    assert pw.plan_for(1 << 20, 4096, 4096, in_elems_per_sample=64 * ((1 << 23) + 4096))[2] == "f32"
    old = pack.F16X2_POINTWISE
    try:
        pack.F16X2_POINTWISE = False
        off = pack.PackedConv("pw", torch.randn(256, 64, 1, 1) / 8, None, "cpu", precision="f16x2")
        assert not off.pointwise_split and off.precision == "f32"
    finally:
        pack.F16X2_POINTWISE = old
