"""GPU parity of the network executors (emoportraits_amd/nets.py) against
  (a) the committed golden outputs of the REAL reference (tests/golden/tiny_hotpath.pt, reduced width), and
  (b) the oracle (oracle/restate.py, pinned bit-exactly to the reference in the build container) at the released
      architecture, R256 and R512, with seeded random checkpoints in the reference key layout.

Tolerances (stated, all relative to max|reference tensor| unless "abs"):
  stage-wise (each stage fed the oracle's own input): conv/GN stacks 5e-5, image 5e-4 abs, samplers 5e-5 (explicit grids
      are bit-exact; the analytic head-pose grid differs from the CPU GEMM by <= 1 ulp in the coordinates);
  end-to-end (error of the predicted warp propagated through two trilinear samplers and ~40 convs): 1e-3, image 5e-3 abs.
fp32 kernels (exact-fp32 MFMA) with a different summation order than the CPU library; no reduced precision anywhere.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import restate as O  # noqa: E402

from emoportraits_amd import config, nets, random_init  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-4


def rel(got, ref):
    return (got.detach().cpu().double() - ref.double()).abs().max().item() / (ref.double().abs().max().item() + 1e-30)


def check(stage, got, ref, tol=RTOL):
    assert got.shape == ref.shape, (stage, got.shape, ref.shape)
    e = rel(got, ref)
    assert e <= tol, f"{stage}: rel err {e:.3e} > {tol:.1e}"
    return e


@pytest.fixture(scope="module")
def tiny(golden_dir):
    return torch.load(os.path.join(golden_dir, "tiny_hotpath.pt"), weights_only=False)


def test_tiny_hotpath_against_reference_golden(tiny):
    cfg = config.hot_path_config(overrides=tiny["cfg"])
    hp = nets.HotPath(tiny["state_dict"], cfg, DEV)
    d = lambda t: t.to(DEV)
    src = hp.source_pass(d(tiny["img"]), d(tiny["idt_embed"]), d(tiny["source_pose_embed"]), d(tiny["theta_src"]), keep=True)
    g = tiny["source"]
    check("source.warp_embed", src["warp_embed"], g["warp_embed"])
    check("source.latents", src["latents"], g["latents"])
    check("source.source_volume", src["source_volume"], g["source_volume"])
    e_pre = check("source.pre_canonical", src["pre_canonical"], g["pre_canonical"], 1e-3)
    e_can = check("source.canonical", src["canonical"], g["canonical"], 1e-3)
    print("PARITY tiny golden source: pre_canonical", f"{e_pre:.2e}", "canonical", f"{e_can:.2e}")
    # driver pass on the reference's canonical volume, both frames in one batch
    ccl = hp.prepare_canonical(d(g["canonical"]))
    drv = hp.driver_pass(ccl, d(tiny["idt_embed"]), d(tiny["target_pose_embed"]), d(tiny["theta_drv"]), keep=True)
    dd, ss = cfg["latent_volume_depth"], cfg["latent_volume_size"]
    ident = O.identity_grid_3d(dd, ss)[..., :3].view(1, dd, ss, ss, 3)
    for i in range(2):
        r = tiny["driver"][i]
        check(f"driver[{i}].warp_embed", drv["warp_embed"][i:i + 1], r["warp_embed"])
        delta_ref = (r["uv_warp"] - ident).permute(0, 4, 1, 2, 3)
        e_d = (drv["delta_uv"][i:i + 1].cpu() - delta_ref).abs().max().item()
        assert e_d <= 2e-4
        # end-to-end through the predicted warp: 1e-3 of max (conditioning of the sampler wrt 1e-5 warp differences)
        e = [check(f"driver[{i}].aligned", drv["aligned"][i:i + 1], r["aligned"], 1e-3),
             check(f"driver[{i}].deep_f", drv["deep_f"][i:i + 1], r["deep_f"], 1e-3),
             check(f"driver[{i}].img_f", drv["img_f"][i:i + 1], r["img_f"], 1e-3)]
        e_img = (drv["img"][i:i + 1].cpu() - r["img"]).abs().max().item()
        assert e_img <= 5e-3
        print(f"PARITY tiny golden driver[{i}]: delta_abs {e_d:.2e} aligned/deep_f/img_f", [f"{v:.2e}" for v in e], f"img_abs {e_img:.2e}")


def _smooth_volume(g):
    """feature-volume-like test data: band-limited noise (trilinear x4 upsampling of coarse noise).  White noise would
    make the sampler's output maximally sensitive to 1e-5-level differences in the predicted warp, which measures the
    conditioning of the path rather than the kernels."""
    coarse = torch.randn(1, 96, 4, 16, 16, generator=g)
    return torch.nn.functional.interpolate(coarse, scale_factor=4, mode="trilinear").contiguous()


def _full_size(S, B, seed):
    cfg = config.hot_path_config(overrides={"image_size": S})
    sd = random_init.random_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    rnd = lambda *s: torch.randn(*s, generator=g)
    inputs = dict(
        img=torch.rand(1, 3, S, S, generator=g), idt=rnd(1, 512, 4, 4), pose_s=rnd(1, 128), pose_t=rnd(B, 128),
        th_s=O.get_transform_matrix(1 + 0.05 * rnd(1, 3), 0.3 * rnd(1, 3), 0.05 * rnd(1, 3)),
        th_t=O.get_transform_matrix(1 + 0.05 * rnd(B, 3), 0.3 * rnd(B, 3), 0.05 * rnd(B, 3)),
        canonical=_smooth_volume(g))
    return cfg, sd, inputs


@pytest.mark.parametrize("S,B", [(256, 2), (512, 1)])
def test_driver_pass_released_architecture_vs_oracle(S, B):
    cfg, sd, x = _full_size(S, B, seed=S)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.driver_pass(sd, cfg, x["canonical"], x["idt"], x["pose_t"], x["th_t"])
    hp = nets.HotPath(sd, cfg, DEV, with_source=False)
    d = lambda t: t.to(DEV)
    ccl = hp.prepare_canonical(d(x["canonical"]))
    got = hp.driver_pass(ccl, d(x["idt"]), d(x["pose_t"]), d(x["th_t"]), keep=True)
    errs = {}
    errs["warp_embed"] = rel(got["warp_embed"], ref["warp_embed"])
    errs["delta_abs"] = (got["delta_uv"].cpu() - ref["delta_uv"]).abs().max().item()
    errs["aligned"] = rel(got["aligned"], ref["aligned"])
    errs["deep_f"] = rel(got["deep_f"], ref["deep_f"])
    errs["img_f"] = rel(got["img_f"], ref["img_f"])
    errs["img_abs"] = (got["img"].cpu() - ref["img"]).abs().max().item()
    # stage-wise: every stage fed with the ORACLE's input of that stage (isolates kernel error from conditioning)
    st = {}
    delta_ref = d(ref["delta_uv"])
    from emoportraits_amd import ops
    lay = "ndhwc"
    warped = ops.grid_sample3d(ccl, delta=delta_ref, in_layout=lay, out_layout=lay)
    aligned = ops.grid_sample3d(warped, theta=d(x["th_t"]), in_layout=lay, out_layout="ncdhw")
    st["samplers"] = rel(aligned, ref["aligned"])
    img, deep_f, img_f = hp.decoder(d(ref["aligned"]).view(B, -1, 64, 64))
    st["deep_f"] = rel(deep_f, ref["deep_f"])
    st["img_f"] = rel(img_f, ref["img_f"])
    st["img_abs"] = (img.cpu() - ref["img"]).abs().max().item()
    print(f"PARITY R{S} B={B} driver end-to-end:", {k: f"{v:.2e}" for k, v in errs.items()})
    print(f"PARITY R{S} B={B} driver stage-wise:", {k: f"{v:.2e}" for k, v in st.items()})
    assert errs["warp_embed"] <= 1e-5 and errs["delta_abs"] <= 1e-4, errs
    # measured on MI355X (archive/profiles/r1_parity.txt): samplers 8.5e-6, deep_f 2.9e-6, img_f 7.7e-6, img 1.0e-4 abs;
    # end-to-end: aligned 1.5e-4, deep_f 6.1e-5, img_f 1.2e-4, img 1.6e-3 abs (sigmoid of a WS-normalised 128-ch head:
    # the head's pre-activation gain turns 1e-4 relative feature error into 1e-3 of the [0,1] range)
    assert st["samplers"] <= 5e-5, st     # explicit grids are bit-exact; analytic theta differs by <= 1 ulp in coords
    assert st["deep_f"] <= 5e-5 and st["img_f"] <= 5e-5 and st["img_abs"] <= 5e-4, st
    assert errs["aligned"] <= 1e-3 and errs["deep_f"] <= 1e-3 and errs["img_f"] <= 1e-3 and errs["img_abs"] <= 5e-3, errs


def test_decoder_alone_is_tight():
    """same input tensor on both sides: isolates the conv/GN kernels from warp sensitivity"""
    cfg, sd, x = _full_size(256, 1, seed=7)
    feat = torch.randn(2, 1536, 64, 64, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref_img, ref_feat, ref_imgf = O.decoder(sd, "decoder_nw", feat, cfg)
    dec = nets.Decoder(sd, "decoder_nw", cfg, DEV)
    img, f2, imgf = dec(feat.to(DEV))
    e1 = check("deep_f", f2, ref_feat, 5e-5)
    e2 = check("img_f", imgf, ref_imgf, 5e-5)
    e3 = (img.cpu() - ref_img).abs().max().item()
    assert e3 <= 5e-5
    print("PARITY decoder alone:", e1, e2, e3)


def test_warp_generator_alone():
    cfg, sd, x = _full_size(256, 3, seed=9)
    emb = torch.randn(3, 512, 16, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        _, ref_delta = O.warp_generator(sd, "uv_generator_nw", emb, cfg)
    wg = nets.WarpGenerator(sd, "uv_generator_nw", cfg, DEV)
    got = wg(emb.to(DEV))
    assert got.shape == ref_delta.shape
    e = (got.cpu() - ref_delta).abs().max().item()
    assert e <= 1e-4, e
    print("PARITY warp generator abs:", e)


def test_source_pass_released_architecture_vs_oracle():
    cfg, sd, x = _full_size(256, 1, seed=11)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.source_pass(sd, cfg, x["img"], x["idt"], x["pose_s"], x["th_s"])
    hp = nets.HotPath(sd, cfg, DEV)
    d = lambda t: t.to(DEV)
    got = hp.source_pass(d(x["img"]), d(x["idt"]), d(x["pose_s"]), d(x["th_s"]), keep=True)
    e = [check("latents", got["latents"], ref["latents"]),
         check("source_volume", got["source_volume"], ref["source_volume"]),
         check("pre_canonical", got["pre_canonical"], ref["pre_canonical"], 1e-3),
         check("canonical", got["canonical"], ref["canonical"], 1e-3)]
    print("PARITY source pass R256:", e)


def test_batched_driver_equals_per_frame_calls():
    """size-independent property: the batch dimension is embarrassingly parallel.  Not bit-for-bit: the
    launch plan depends on the batch size (a batch-1 launch splits the K loop of the small layers over more blocks,
    pack.plan_launch), which re-associates the fp32 sums; that rounding noise reaches the image through the same gain as in the end-to-end
    parity test (weight-standardised 128-channel head: measured 4.9e-4 here, 1.6e-3 vs the oracle, bound 5e-3)."""
    cfg, sd, x = _full_size(256, 3, seed=13)
    hp = nets.HotPath(sd, cfg, DEV, with_source=False)
    d = lambda t: t.to(DEV)
    ccl = hp.prepare_canonical(d(x["canonical"]))
    full = hp.driver_pass(ccl, d(x["idt"]), d(x["pose_t"]), d(x["th_t"]))
    for i in range(3):
        one = hp.driver_pass(ccl, d(x["idt"]), d(x["pose_t"][i:i + 1]), d(x["th_t"][i:i + 1]))
        err = (one - full[i:i + 1]).abs().max().item()
        print("PARITY batched vs per-frame driver pass:", err)
        assert err <= 2e-3


def test_driver_pass_fp16_operand_mode_vs_oracle():
    """opt-in reduced precision of the stage-1 driver pass (HotPath(precision="f16")): fp16 MFMA operands, fp32
    accumulation.  Compared with the fp32 oracle at R256; the default fp32 path is tested above at 1e-3 / 5e-3."""
    cfg, sd, x = _full_size(256, 1, seed=17)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.driver_pass(sd, cfg, x["canonical"], x["idt"], x["pose_t"], x["th_t"])
    hp = nets.HotPath(sd, cfg, DEV, with_source=False, precision="f16")
    d = lambda t: t.to(DEV)
    got = hp.driver_pass(hp.prepare_canonical(d(x["canonical"])), d(x["idt"]), d(x["pose_t"]), d(x["th_t"]), keep=True)
    e_feat = (got["img_f"].cpu() - ref["img_f"]).abs().max().item() / ref["img_f"].abs().max().item()
    e_img = (got["img"].cpu() - ref["img"]).abs().max().item()
    e_mean = (got["img"].cpu() - ref["img"]).abs().mean().item()
    print("PARITY driver pass R256 fp16 operands:", f"features {e_feat:.2e} of max, image {e_img:.2e} abs max, {e_mean:.2e} abs mean")
    # seeded random weights with a saturated sigmoid head (the worst case, see the trained-like test below).  Round-2 kernel
    # (32x32x16 MFMA) with the WarpGenerators kept in fp32: measured features 1.8e-3 of max, image 7.6e-4 mean / 1.8e-2 worst
    # pixel (round 1, everything in fp16: 3.1e-2 / 1.3e-2 / 0.31)
    assert e_feat <= 1e-2 and e_mean <= 5e-3 and e_img <= 1e-1 and torch.isfinite(got["img"]).all()


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2", "f16"])
def test_driver_pass_with_trained_like_image_statistics(precision):
    """The bounds of the tests above are characterised on a seeded random checkpoint whose sigmoid head is saturated (random
    weight-standardised head: pre-activation std ~8), which turns 1e-4 of feature error into 1e-3 of the [0,1] range and
    makes single pixels flip.  The released checkpoint is not obtainable here; this test gives the seeded checkpoint the
    image statistics of a trained decoder instead (random_init.random_state_dict(image_head_gain=0.2): logits of a few
    units, images in mid-range) and states the tolerance on THAT: fp32 path 5e-4 worst pixel (measured 1.6e-4, mean 1.1e-5);
    fp16-operand mode (BASELINE configs[4], opt-in; decoder in fp16 operands, WarpGenerator fp32) 2e-3 mean / 2e-2 worst pixel
    (measured 3.9e-4 / 4.6e-3; with the WarpGenerator in fp16 as well: 3.2e-3 / 5.3e-2)."""
    S, B = 256, 2
    cfg = config.hot_path_config(overrides={"image_size": S})
    sd = random_init.random_state_dict(cfg, seed=31, image_head_gain=0.2)
    _, _, x = _full_size(S, B, seed=31)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.driver_pass(sd, cfg, x["canonical"], x["idt"], x["pose_t"], x["th_t"])
    frac_mid = ((ref["img"] > 0.02) & (ref["img"] < 0.98)).float().mean().item()
    hp = nets.HotPath(sd, cfg, DEV, with_source=False, precision=precision)
    d = lambda t: t.to(DEV)
    got = hp.driver_pass(hp.prepare_canonical(d(x["canonical"])), d(x["idt"]), d(x["pose_t"]), d(x["th_t"]), keep=True)
    diff = (got["img"].cpu() - ref["img"]).abs()
    e_max, e_mean = diff.max().item(), diff.mean().item()
    e_feat = rel(got["img_f"], ref["img_f"])
    print(f"PARITY driver pass R{S} trained-like image statistics ({precision} operands): image max {e_max:.2e} mean {e_mean:.2e}, "
          f"features {e_feat:.2e} of max; {frac_mid:.2f} of the reference pixels are unsaturated")
    assert frac_mid > 0.5, "the checkpoint is supposed to produce unsaturated images"
    if precision in ("f32", "bf16x3", "f16x2"):       # all three are fp32 paths: the same bound
        assert e_max <= 5e-4 and e_feat <= 1e-3
    else:
        assert e_mean <= 2e-3 and e_max <= 2e-2
