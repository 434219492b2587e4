"""Configuration contract of the hot path (SURVEY.md section 5 "Config / flags", F4).

The reference builds its networks from an argparse Namespace: trainer flags are dumped as `key: value` lines to
logs/<exp>/args.txt (train.py:80-83) and re-parsed at inference by utils/args.py:34-65 with type sniffing
('True'/'False' -> bool, digits -> int, float-parsable -> float), then `args_overwrite` is applied
(notebooks/infer.py:74-81).  experiments/args.txt in the repo is the *launch command* of the released model.

This module accepts both forms and keeps only the keys that shape the hot path.  Defaults are the argparse
defaults of models/stage_1/volumetric_avatar/va_arguments.py (line numbers in DEFAULTS).
"""
import shlex
from argparse import Namespace

# key: (va_arguments.py default, line)
DEFAULTS = {
    "image_size": (256, None),                      # train.py flag; released: 512
    "latent_volume_channels": (64, 249), "latent_volume_depth": (16, 251), "latent_volume_size": (64, 250),
    "gen_latent_texture_channels": (64, 247), "gen_latent_texture_depth": (16, 246), "gen_latent_texture_size": (64, 245),
    "gen_num_channels": (32, 204), "gen_max_channels": (512, 205), "gen_max_channels_unet3d": (512, 211),
    "enc_channel_mult": (2.0, None), "gen_embed_size": (4, 267), "gen_dummy_input_size": (4, None),
    "warp_output_size": (64, 286), "warp_channel_mult": (1.0, 284),
    "source_volume_num_blocks": (0, None),
    "dec_num_blocks": (8, None), "dec_channel_mult": (2.0, None), "dec_max_channels": (512, None),
    "im_dec_num_lrs_per_resolution": (1, None), "im_dec_ch_div_factor": (2.0, None),
    "lpe_output_channels_expression": (512, None), "local_encoder_input_size": (3, 166),
    "grid_sample_padding_mode": ("zeros", 185),
    "norm_layer_type": ("bn", None), "use_sn": (True, None), "use_ws": (False, None),
    "dec_use_adanorm": (False, None), "gen_use_adanorm": (False, None), "gen_use_adaconv": (False, None),
    "dec_use_adaconv": (False, None), "use_back": (True, None), "volume_rendering": (False, None),
    "warp_norm_grad": (False, None), "unet_first": (False, None), "cat_em": (False, 220),
    "no_channel_increase_3d_source": (True, 167), "tex_use_skip_resblock": (True, 294),
    "gen_activation_type": ("relu", 214), "gen_upsampling_type": ("trilinear", 216),
    "gen_downsampling_type": ("avgpool", 215), "warp_block_type": ("res", 283), "dec_up_block_type": ("res", 310),
    "enc_block_type": ("res", 291), "dec_bigger": (False, None), "use_tensor": (False, None),
    "pred_volume_num_blocks": (0, None), "dec_pred_seg": (False, None),
}

# the flags of the released model (experiments/args.txt) that differ from the defaults above
RELEASED = dict(
    image_size=512, norm_layer_type="gn", use_ws=True, use_sn=True, enc_channel_mult=4.0, gen_dummy_input_size=8,
    latent_volume_channels=96, gen_latent_texture_channels=96, source_volume_num_blocks=3,
    dec_num_blocks=6, dec_channel_mult=2.0, dec_max_channels=512, im_dec_num_lrs_per_resolution=2,
    im_dec_ch_div_factor=1.5, dec_use_adanorm=False, lpe_output_channels_expression=128, use_back=False,
    dec_pred_seg=False, use_tensor=False,
)


def _sniff(v):
    """utils/args.py:34-51 parse_args_line value typing"""
    if isinstance(v, str):
        if v.isdigit():
            return int(v)
        try:
            return float(v)
        except ValueError:
            pass
        if v == "True":
            return True
        if v == "False":
            return False
    return v


def parse_args_txt(path):
    """logs/<exp>/args.txt (`key: value` per line, utils/args.py:54-65) or a launch command line
    (experiments/args.txt) -> dict of ALL keys found."""
    text = open(path, "rt").read()
    out = {}
    if "--" in text and "\n" not in text.strip():
        toks = shlex.split(text)
        i = 0
        while i < len(toks):
            if toks[i].startswith("--") and i + 1 < len(toks) and not toks[i + 1].startswith("--"):
                out[toks[i][2:]] = _sniff(toks[i + 1])
                i += 2
            else:
                i += 1
        return out
    for line in text.splitlines():
        if not line.strip():
            continue
        parts = line.split(": ")
        if len(parts) < 2:
            continue
        out[parts[0]] = _sniff(": ".join(parts[1:]))
    return out


def hot_path_config(found=None, overrides=None, released=True):
    """dict with every DEFAULTS key: defaults <- (released flags) <- file contents <- overrides"""
    cfg = {k: v[0] for k, v in DEFAULTS.items()}
    if released:
        cfg.update(RELEASED)
    for src in (found or {}, overrides or {}):
        for k, v in src.items():
            if k in cfg:
                cfg[k] = type(cfg[k])(v) if not isinstance(cfg[k], bool) else (v is True or v == "True")
    validate(cfg)
    return cfg


def validate(cfg):
    """The HIP path implements the released architecture family; anything else fails loudly (no silent fallback)."""
    problems = []
    if cfg["norm_layer_type"] != "gn":
        problems.append("norm_layer_type must be 'gn'")
    for k in ("dec_use_adanorm", "gen_use_adanorm", "gen_use_adaconv", "dec_use_adaconv", "use_back",
              "volume_rendering", "warp_norm_grad", "unet_first", "cat_em", "dec_bigger", "use_tensor", "dec_pred_seg"):
        if cfg[k]:
            problems.append(f"{k}=True is not on the released hot path")
    if cfg["pred_volume_num_blocks"] != 0:
        problems.append("pred_volume_num_blocks must be 0")
    if not cfg["no_channel_increase_3d_source"]:
        problems.append("no_channel_increase_3d_source must be True")
    for k in ("gen_activation_type", "gen_upsampling_type", "gen_downsampling_type", "warp_block_type",
              "dec_up_block_type", "enc_block_type"):
        if cfg[k] != DEFAULTS[k][0]:
            problems.append(f"{k}={cfg[k]!r} unsupported")
    if cfg["latent_volume_channels"] != cfg["gen_latent_texture_channels"]:
        problems.append("latent_volume_channels != gen_latent_texture_channels")
    if cfg["warp_output_size"] != cfg["gen_latent_texture_size"]:
        problems.append("warp_output_size != gen_latent_texture_size (resize_warp) unsupported")
    # every feature map on the path has width image_size / 2^k down to the latent size; the conv kernels tile output widths
    # that are multiples of 128, or 64 / 32 / 16 / 8 (csrc/conv_api.hip shape_of_width) -- reject e.g. 384 or 768 here, not at
    # the first convolution
    S, L = cfg["image_size"], cfg["gen_latent_texture_size"]
    if S < L or S % L or (S // L) & (S // L - 1):
        problems.append(f"image_size={S} must be gen_latent_texture_size={L} times a power of two")
    elif any(w % 128 and w not in (64, 32, 16, 8) for w in [S >> k for k in range((S // L).bit_length())]):
        problems.append(f"image_size={S}: a feature-map width on the path is not tiled by the conv kernels "
                        f"(multiples of 128, or 64 / 32 / 16 / 8)")
    if not (cfg["use_sn"] and cfg["use_ws"]):
        problems.append("use_sn and use_ws must both be True (released key layout)")
    if problems:
        raise ValueError("unsupported configuration for the MI355X hot path: " + "; ".join(problems))


def as_namespace(cfg):
    return Namespace(**cfg)
