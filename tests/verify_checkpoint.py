"""Parity table of THIS build on a REAL checkpoint, in one command -- test infrastructure (it imports the CPU oracle as the
checker, which only files under tests/ may do; the HIP side is the product path, unchanged):

    python tests/verify_checkpoint.py <args.txt> <model.pth> [--batch 16] [--frames 0,7,15] [--modes default,f32,f16x2]
                                      [--no-source] [--seed 0] [--json table.json]

The released EMOPortraits weights (`logs.zip`, reference README.md:125-139) cannot be fetched into the build container, so
every tolerance this repository states rests on seeded checkpoints (tests/test_bench_config_parity_gpu.py).  A user who has
the release runs this script on `logs/<experiment>/args.txt` + `logs/<experiment>/checkpoints/<file>.pth`:

  * the checkpoint is loaded STRICTLY through the product's loader (emoportraits_amd.schema.check_state_dict -- the
    reference's own `load_state_dict(strict=False)`, notebooks/infer.py:124-131, hides missing / mis-shaped keys);
  * seeded inputs of the architecture's shapes (source image, identity / pose embeddings, head-pose matrices on the SURVEY.md
    section 8(d) distribution) go through the oracle (oracle/restate.py: the reference's modules restated over the raw
    state_dict, pinned bit-exactly to the reference) and through the HIP path at the BENCH launch plan (driver batch of
    `--batch` frames; frames `--frames` are compared, the oracle runs them one by one);
  * for every conv mode (`default` = what InferenceWrapper runs, `f32` = exact-fp32 MFMA everywhere, `f16x2` = the guarded
    fp16 split) it prints: source pass stage errors, driver pass stage errors, image max-abs error, share of identical uint8
    bytes, batch-1 vs batch-B agreement, and which fp16-split layers (if any) tripped their range check and were recomputed.

Exit code 0 when every row is inside the bounds of tests/test_bench_config_parity_gpu.py (end to end: stages 1e-3 of max,
image 5e-3 abs; warp delta 1e-4 abs), 1 otherwise.  Needs an MI355X; the oracle legs run on the host cores (about 2 s per
512^2 driver frame and 15 s for the source pass on 64 threads).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import restate as O  # noqa: E402  (checker)

from emoportraits_amd import config as cfg_mod  # noqa: E402
from emoportraits_amd import nets, ops, schema  # noqa: E402


def rel(got, ref):
    return (got.detach().cpu().double() - ref.double()).abs().max().item() / (ref.double().abs().max().item() + 1e-30)


def seeded_inputs(cfg, B, seed):
    S = cfg["image_size"]
    g = torch.Generator().manual_seed(seed + 1)
    rnd = lambda *s: torch.randn(*s, generator=g)
    E = cfg["lpe_output_channels_expression"]
    C = cfg["gen_max_channels"]
    es = cfg["gen_embed_size"]
    return dict(img=torch.rand(1, 3, S, S, generator=g), idt=rnd(1, C, es, es), pose_s=rnd(1, E), pose_t=rnd(B, E),
                th_s=O.get_transform_matrix(1 + 0.05 * rnd(1, 3), 0.3 * rnd(1, 3), 0.05 * rnd(1, 3)),
                th_t=O.get_transform_matrix(1 + 0.05 * rnd(B, 3), 0.3 * rnd(B, 3), 0.05 * rnd(B, 3)))


def verify(args_path, ckpt, batch=16, frames=(0, 7, 15), modes=("default", "f32", "f16x2"), with_source=True, seed=0,
           device="cuda:0", log=print):
    found = cfg_mod.parse_args_txt(args_path)
    cfg = cfg_mod.hot_path_config(found, released=False) if "norm_layer_type" in found else cfg_mod.hot_path_config(found)
    sd = torch.load(ckpt, map_location="cpu") if not isinstance(ckpt, dict) else ckpt
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd and not any(k.startswith("decoder_nw") for k in sd) else sd
    schema.check_state_dict(sd, cfg)                               # strict: raises KeyError naming what is missing / mis-shaped
    frames = tuple(f for f in frames if f < batch)
    x = seeded_inputs(cfg, batch, seed)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    d = lambda t: t.to(device)
    table = dict(args=str(args_path), image_size=cfg["image_size"], batch=batch, frames=list(frames), rows=[])
    with torch.no_grad():
        if with_source:
            ref_src = O.source_pass(sd, cfg, x["img"], x["idt"], x["pose_s"], x["th_s"])
            canonical = ref_src["canonical"]
        else:
            c, dd, s = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
            canonical = torch.randn(1, c, dd, s, s, generator=torch.Generator().manual_seed(seed + 2)) * 0.5
        ref_drv = {i: O.driver_pass(sd, cfg, canonical, x["idt"], x["pose_t"][i:i + 1], x["th_t"][i:i + 1]) for i in frames}
    ok = True
    for mode in modes:
        hp = nets.HotPath(sd, cfg, device, with_source=with_source, precision=None if mode == "default" else mode)
        row = dict(mode=mode, conv_precision=hp.precision)
        if with_source:
            got = hp.source_pass(d(x["img"]), d(x["idt"]), d(x["pose_s"]), d(x["th_s"]), keep=True)
            row["source"] = {k: rel(got[k], ref_src[k]) for k in ("latents", "source_volume", "pre_canonical", "canonical")}
            row["source_overflow_layers"] = sorted(v for v in hp.overflow_events().values() if v) if hp.precision == "f16x2" else []
            ok &= all(v <= 1e-3 for v in row["source"].values())
        ccl = hp.prepare_canonical(d(canonical))                    # the driver pass continues from the ORACLE's volume
        got = hp.driver_pass(ccl, d(x["idt"]), d(x["pose_t"]), d(x["th_t"]), keep=True)
        row["driver_overflow_layers"] = sorted(v for v in hp.overflow_events().values() if v) if hp.precision == "f16x2" else []
        u8 = ops.pack_rgb8(got["img"]).cpu()
        worst = dict(warp_embed=0.0, delta_abs=0.0, aligned=0.0, deep_f=0.0, img_f=0.0, img_abs=0.0, batch1_abs=0.0, u8_same=1.0)
        for i in frames:
            ref = ref_drv[i]
            one = hp.driver_pass(ccl, d(x["idt"]), d(x["pose_t"][i:i + 1]), d(x["th_t"][i:i + 1]))
            ref_u8 = (ref["img"].clamp(0, 1) * 255.0).to(torch.uint8).permute(0, 2, 3, 1)     # ToPILImage: mul(255).byte()
            e = dict(warp_embed=rel(got["warp_embed"][i:i + 1], ref["warp_embed"]),
                     delta_abs=(got["delta_uv"][i:i + 1].cpu() - ref["delta_uv"]).abs().max().item(),
                     aligned=rel(got["aligned"][i:i + 1], ref["aligned"]), deep_f=rel(got["deep_f"][i:i + 1], ref["deep_f"]),
                     img_f=rel(got["img_f"][i:i + 1], ref["img_f"]),
                     img_abs=(got["img"][i:i + 1].cpu() - ref["img"]).abs().max().item(),
                     batch1_abs=(one - got["img"][i:i + 1]).abs().max().item())
            for k, v in e.items():
                worst[k] = max(worst[k], v)
            worst["u8_same"] = min(worst["u8_same"], (u8[i:i + 1] == ref_u8).float().mean().item())
        row["driver"] = worst
        ok &= worst["warp_embed"] <= 1e-5 and worst["delta_abs"] <= 1e-4
        ok &= worst["aligned"] <= 1e-3 and worst["deep_f"] <= 1e-3 and worst["img_f"] <= 1e-3 and worst["img_abs"] <= 5e-3
        table["rows"].append(row)
        log(f"[{mode}] conv arithmetic {hp.precision}")
        if with_source:
            log("    source pass  " + "  ".join(f"{k} {v:.2e}" for k, v in row["source"].items()))
        log("    driver pass  " + "  ".join(f"{k} {v:.3e}" for k, v in worst.items()))
        if hp.precision == "f16x2":
            log(f"    fp16-split layers recomputed after their range check: source {row.get('source_overflow_layers', [])} "
                f"driver {row['driver_overflow_layers']}")
        del hp
        torch.cuda.empty_cache()
    table["ok"] = bool(ok)
    log("RESULT: " + ("inside the stated bounds" if ok else "OUTSIDE the stated bounds"))
    return table


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("args_txt")
    ap.add_argument("checkpoint")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", default="0,7,15")
    ap.add_argument("--modes", default="default,f32,f16x2")
    ap.add_argument("--no-source", action="store_true", help="skip the source pass (seeded canonical volume instead)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    table = verify(a.args_txt, a.checkpoint, a.batch, tuple(int(f) for f in a.frames.split(",")), tuple(a.modes.split(",")),
                   not a.no_source, a.seed)
    if a.json:
        with open(a.json, "wt") as f:
            json.dump(table, f, indent=1)
    sys.exit(0 if table["ok"] else 1)


if __name__ == "__main__":
    main()
