"""The hot path end to end WITHOUT a GPU: the package's own host code (emoportraits_amd.nets / ops / pack: launch planning, weight
packing, buffers, the overflow-word guard) drives the product's own kernel sources, compiled for the host (tests/emul/emulibs.py:
every convolution mode, the sampler, GroupNorm, resampling, the small operators), on CPU tensors -- and the result is compared
with the outputs of the REAL reference (tests/golden/tiny_hotpath.pt: the released architecture at reduced width, the fixture
__graft_entry__.smoke() checks on the GPU) and with the oracle restatement.  94 C-ABI calls per driver pass, 149 per source pass.

This is test infrastructure, not a CPU path of the product: the package is pointed at the host-compiled libraries by monkeypatching
emoportraits_amd.hip inside this test only; emoportraits_amd itself refuses to run without its GPU library.

    default            one driver frame (about 50 s on 8 cores)
    EMO_EMUL_FULL=1    + the source pass, both golden driver frames and the stage-2 refinement (about 5 minutes more)
"""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "emul"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import emulibs  # noqa: E402

pytestmark = pytest.mark.skipif(not emulibs.available(), reason="needs ROCm clang++ and the built product library (weight packing asks it for tile sizes)")
FULL = os.environ.get("EMO_EMUL_FULL") == "1"


@pytest.fixture()
def hot_path(monkeypatch):
    lib = emulibs.install(monkeypatch.setattr)
    from emoportraits_amd import config, nets
    monkeypatch.delenv("EMO_CONV_PRECISION", raising=False)
    tiny = torch.load(os.path.join(HERE, "golden", "tiny_hotpath.pt"), weights_only=False)
    cfg = config.hot_path_config(overrides=tiny["cfg"])
    hp = nets.HotPath(tiny["state_dict"], cfg, "cpu")
    assert hp.precision == "f16x2"                                     # the default mode: fp16 split + guarded recomputation
    return hp, tiny, cfg, lib


def test_driver_frame_through_the_emulated_kernels_matches_the_reference(hot_path):
    """pose theta and embedding -> WarpGenerator (upsampling with the next norm's sums, adaptive GroupNorm, 3-D convolutions) ->
    the two 3-D grid_sample calls -> Decoder (fp16-split 3x3 layers with their guarded launches, pointwise and fp32 MFMA layers,
    tile statistics) -> the image head: against the reference's own output for the frame (the GPU's smoke() bound) and the oracle"""
    import restate as O
    from emoportraits_amd import ops
    hp, tiny, cfg, lib = hot_path
    ref_c = tiny["source"]["canonical"]
    ccl = hp.prepare_canonical(ref_c)
    pose, theta = tiny["target_pose_embed"][:1], tiny["theta_drv"][:1]
    img = hp.driver_pass(ccl, tiny["idt_embed"], pose, theta)
    assert tuple(img.shape) == (1, 3, 64, 64) and torch.isfinite(img).all()
    e_gold = (img - tiny["driver"][0]["img"]).abs().max().item()
    with torch.no_grad():
        oracle = O.driver_pass(tiny["state_dict"], cfg, ref_c, tiny["idt_embed"], pose, theta)
    e_or = (img - oracle["img"]).abs().max().item()
    print(f"PARITY emulated driver frame: abs err vs reference golden {e_gold:.2e}, vs oracle {e_or:.2e}; {sum(lib.calls.values())} C-ABI calls")
    assert e_gold < 5e-4 and e_or < 5e-4                               # (the GPU reads 1.9e-4 / 1.0e-4 on two frames)
    assert lib.calls["emo_conv_igemm_f16x2"] == lib.calls["emo_conv_igemm_bf16x3"] >= 5      # every split launch followed by its guard
    assert ops.overflow_events("cpu") == {}                            # no layer recomputed: every guarded launch left at once
    u8 = ops.pack_rgb8(img)
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (1, 64, 64, 3)


@pytest.mark.skipif(not FULL, reason="EMO_EMUL_FULL=1: the source pass and both golden frames (minutes)")
def test_source_pass_and_both_frames_through_the_emulated_kernels(hot_path):
    hp, tiny, cfg, lib = hot_path
    canonical = hp.source_pass(tiny["img"], tiny["idt_embed"], tiny["source_pose_embed"], tiny["theta_src"])
    ref_c = tiny["source"]["canonical"]
    e_src = ((canonical - ref_c).abs().max() / ref_c.abs().max()).item()
    ccl = hp.prepare_canonical(ref_c)
    img = hp.driver_pass(ccl, tiny["idt_embed"], tiny["target_pose_embed"], tiny["theta_drv"])
    e_gold = max((img[i:i + 1] - tiny["driver"][i]["img"]).abs().max().item() for i in range(2))
    print(f"PARITY emulated source pass: canonical rel err {e_src:.2e}; two driver frames vs reference golden {e_gold:.2e}")
    assert e_src < 1e-3 and e_gold < 5e-3                              # __graft_entry__.smoke()'s bounds


@pytest.mark.skipif(not FULL, reason="EMO_EMUL_FULL=1: the stage-2 refinement (about a minute)")
def test_stage2_refinement_through_the_emulated_kernels(monkeypatch):
    """Stage2.refine (LocalEncoderOld + Decoder_stage2, BatchNorm default flags) on the golden outputs of
    the real reference (tests/golden/tiny_stage2.pt) -- the bounds of tests/test_stage2_gpu.py"""
    lib = emulibs.install(monkeypatch.setattr)
    from emoportraits_amd import stage2
    monkeypatch.delenv("EMO_CONV_PRECISION", raising=False)
    tiny = torch.load(os.path.join(HERE, "golden", "tiny_stage2.pt"), weights_only=False)
    s2 = stage2.Stage2(tiny["state_dict"], stage2.stage2_config(tiny["cfg"]), "cpu")
    got = s2.refine(tiny["img"], tiny["mask"], tiny["face_mask"], keep=True)
    rel = lambda a, b: (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)
    e = {k: rel(got[k], tiny[k]) for k in ("latents", "add")}
    e["out_abs"] = (got["out"] - tiny["out"]).abs().max().item()
    print("PARITY emulated stage 2 (tiny golden, bn):", {k: f"{v:.2e}" for k, v in e.items()}, f"{sum(lib.calls.values())} C-ABI calls")
    assert e["latents"] <= 5e-5 and e["add"] <= 2e-4 and e["out_abs"] <= 2e-4, e
