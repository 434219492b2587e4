"""Key layout of the hot-path part of a reference checkpoint (SURVEY.md section 5 "Checkpoint / resume").

`hot_path_schema(cfg)` enumerates every tensor `torch.save(model.state_dict())` of the reference holds for the
networks on the hot path, with its shape, by mirroring how the reference constructs its modules:
  ResBlock                 networks/volumetric_avatar/utils.py:661-788
  spectral-norm keys       utils/spectral_norm.py:193-218   (weight_orig / weight_u / weight_v)
  WS replacement rule      networks/volumetric_avatar/utils.py:1061-1096 (Conv2d after GroupNorm, Conv3d after
                           AdaptiveGroupNorm -> Conv*_ws with bias=True; loses its spectral norm)
It is used (a) to load checkpoints STRICTLY -- the reference's load_state_dict(strict=False) (notebooks/infer.py:131)
silently leaves random weights behind on a key mismatch -- and (b) to build seeded random checkpoints of the
released architecture for the benchmark (the released weights are not in the repo: README.md:125-139).
"""
import math


def _conv_keys(out, prefix, kind, cout, cin, k, dims, bias):
    kshape = (k,) * dims
    if kind == "sn":
        out[prefix + ".weight_orig"] = (cout, cin) + kshape
        out[prefix + ".weight_u"] = (cout,)
        out[prefix + ".weight_v"] = (cin * k ** dims,)
        if bias:
            out[prefix + ".bias"] = (cout,)
    elif kind == "ws":
        out[prefix + ".weight"] = (cout, cin) + kshape
        out[prefix + ".bias"] = (cout,)
    else:
        raise ValueError(kind)


def _res_block(out, prefix, cin, cout, dims, first_kind):
    out[prefix + ".block_feats.0.weight"] = (cin,)
    out[prefix + ".block_feats.0.bias"] = (cin,)
    _conv_keys(out, prefix + ".block_feats.2", first_kind, cout, cin, 3, dims, False)
    out[prefix + ".block_feats.3.weight"] = (cout,)
    out[prefix + ".block_feats.3.bias"] = (cout,)
    _conv_keys(out, prefix + ".block.0", "sn", cout, cout, 3, dims, False)
    if cin != cout:
        _conv_keys(out, prefix + ".skip.0", "sn", cout, cin, 1, dims, False)


def warp_generator_channels(cfg):
    nb = int(math.log(cfg["warp_output_size"] // cfg["gen_embed_size"], 2))
    f = lambda i: min(int(cfg["gen_num_channels"] * cfg["warp_channel_mult"] * 2 ** i), cfg["gen_max_channels"]) // 32 * 32
    return [f(nb)] + [f(i) for i in range(nb - 1, -1, -1)]


def decoder_channels(cfg):
    nup = int(math.log(cfg["image_size"] // cfg["gen_latent_texture_size"], 2))
    trunk = min(int(cfg["gen_num_channels"] * cfg["dec_channel_mult"] * 2 ** nup), cfg["dec_max_channels"])
    stages, c = [], trunk
    for _ in range(nup):
        c = max(int(c / cfg["im_dec_ch_div_factor"] / 32) * 32, cfg["gen_num_channels"])
        stages.append(c)
    return trunk, stages


def encoder_channels(cfg):
    nblk = int(math.log(cfg["image_size"] // cfg["latent_volume_size"], 2))
    c = int(cfg["gen_num_channels"] * cfg["enc_channel_mult"])
    chans = [c]
    for _ in range(nblk):
        c = min(c * 2, cfg["gen_max_channels"])
        chans.append(c)
    return chans


def unet3d_channels(cfg):
    nb = int(math.log(cfg["gen_latent_texture_size"] // cfg["gen_dummy_input_size"], 2))
    c = cfg["gen_latent_texture_channels"]
    mx = cfg["gen_max_channels_unet3d"]
    down = [c]
    for _ in range(nb):
        down.append(min(down[-1] * 2, mx))
    top = min(int(c * 2 ** nb), mx)
    up = [top] + [min(int(c * 2 ** i), mx) for i in range(nb - 1, -1, -1)]
    return nb, down, up


def _warp_generator(out, prefix, cfg):
    chans = warp_generator_channels(cfg)
    inp, gmax, es = cfg["gen_embed_size"], cfg["gen_max_channels"], cfg["gen_embed_size"]
    d, s = cfg["gen_latent_texture_depth"], cfg["warp_output_size"]
    out[prefix + ".identity_grid"] = (1, 3, d, s, s)
    _conv_keys(out, prefix + ".first_conv", "sn", chans[0] * inp, gmax, 1, 2, False)
    for i in range(len(chans) - 1):
        _res_block(out, f"{prefix}.blocks_3d.{i}", chans[i], chans[i + 1], 3, "ws")
        out[f"{prefix}.projector.u.{2 * i}"] = (chans[i], gmax)
        out[f"{prefix}.projector.v.{2 * i}"] = (es * es, 2)
        out[f"{prefix}.projector.u.{2 * i + 1}"] = (chans[i + 1], gmax)
        out[f"{prefix}.projector.v.{2 * i + 1}"] = (es * es, 2)
    out[prefix + ".pre_head.0.weight"] = (chans[-1],)
    out[prefix + ".pre_head.0.bias"] = (chans[-1],)
    _conv_keys(out, prefix + ".head.0.0", "sn", 3, chans[-1], 3, 3, True)


def driver_schema(cfg):
    """tensors the per-frame driver pass needs"""
    out = {}
    gmax, es = cfg["gen_max_channels"], cfg["gen_embed_size"]
    out["pose_unsqueeze_nw.weight"] = (gmax * es * es, cfg["lpe_output_channels_expression"])
    _conv_keys(out, "warp_embed_head_orig_nw", "sn", gmax, gmax, 1, 2, False)
    _warp_generator(out, "uv_generator_nw", cfg)
    trunk, stages = decoder_channels(cfg)
    cd = cfg["gen_latent_texture_channels"] * cfg["gen_latent_texture_depth"]
    p = "decoder_nw"
    _conv_keys(out, p + ".res_decoder.0", "sn", trunk, cd, 1, 2, False)
    for i in range(cfg["dec_num_blocks"]):
        _res_block(out, f"{p}.res_decoder.{i + 1}", trunk, trunk, 2, "ws")
    k, c = 0, trunk
    for st in stages:
        for j in range(cfg["im_dec_num_lrs_per_resolution"]):
            _res_block(out, f"{p}.img_decoder.dec_img_blocks.{k}", c, st, 2, "ws")
            c = st
            k += 1
    out[p + ".img_decoder.dec_img_head.0.weight"] = (c,)
    out[p + ".img_decoder.dec_img_head.0.bias"] = (c,)
    _conv_keys(out, p + ".img_decoder.dec_img_head.2", "ws", 3, c, 1, 2, True)
    return out


def source_schema(cfg):
    """tensors only the once-per-identity source pass needs"""
    out = {}
    S = cfg["image_size"]
    chans = encoder_channels(cfg)
    p = "local_encoder_nw"
    _conv_keys(out, f"{p}.from_rgb_{S}px", "sn", chans[0], cfg["local_encoder_input_size"], 7, 2, True)
    s = S
    for i in range(len(chans) - 1):
        _res_block(out, f"{p}.enc_{i}_block={s}px", chans[i], chans[i + 1], 2, "ws")
        s //= 2
    out[p + ".finale_layers.0.weight"] = (chans[-1],)
    out[p + ".finale_layers.0.bias"] = (chans[-1],)
    cd = cfg["latent_volume_channels"] * cfg["latent_volume_depth"]
    _conv_keys(out, p + ".finale_layers.2", "ws", cd, chans[-1], 1, 2, True)
    c = cfg["latent_volume_channels"]
    for i in range(cfg["source_volume_num_blocks"]):
        _res_block(out, f"volume_source_nw.net.net.{i}", c, c, 3, "sn")
    _warp_generator(out, "xy_generator_nw", cfg)
    nb, down, up = unet3d_channels(cfg)
    p = "volume_process_nw"
    for i in range(nb):
        _res_block(out, f"{p}.blocks_3d_down.{i}", down[i], down[i + 1], 3, "sn")
    dz = cfg["gen_dummy_input_size"]
    out[p + ".input_tensor"] = (1, up[0], dz, dz, dz)
    for i in range(nb):
        _res_block(out, f"{p}.blocks_3d_up.{i}", up[i], up[i + 1], 3, "sn")
        _res_block(out, f"{p}.skip_blocks_3d_up.{i}", up[i], up[i], 3, "sn")
    out[p + ".head.0.weight"] = (up[-1],)
    out[p + ".head.0.bias"] = (up[-1],)
    _conv_keys(out, p + ".head.2", "sn", up[-1], up[-1], 1, 3, True)
    return out


def hot_path_schema(cfg, with_source=True):
    out = driver_schema(cfg)
    if with_source:
        out.update(source_schema(cfg))
    return out


HOT_PATH_PREFIXES = ("pose_unsqueeze_nw.", "warp_embed_head_orig_nw.", "uv_generator_nw.", "xy_generator_nw.",
                     "decoder_nw.", "local_encoder_nw.", "volume_source_nw.", "volume_process_nw.")


def check_state_dict(sd, cfg, with_source=True):
    """Strict check of the hot-path keys of a checkpoint.  Raises KeyError listing every missing / unexpected /
    mis-shaped tensor (the reference would silently keep random weights instead)."""
    want = hot_path_schema(cfg, with_source)
    prefixes = HOT_PATH_PREFIXES if with_source else HOT_PATH_PREFIXES[:3] + ("decoder_nw.",)
    missing = [k for k in want if k not in sd]
    wrong = [f"{k}: checkpoint {tuple(sd[k].shape)} != expected {want[k]}" for k in want
             if k in sd and tuple(sd[k].shape) != tuple(want[k])]
    unexpected = [k for k in sd if k.startswith(prefixes) and k not in want]
    if missing or wrong or unexpected:
        msg = []
        if missing:
            msg.append(f"missing ({len(missing)}): " + ", ".join(missing[:12]) + (" ..." if len(missing) > 12 else ""))
        if wrong:
            msg.append(f"shape mismatch ({len(wrong)}): " + "; ".join(wrong[:8]) + (" ..." if len(wrong) > 8 else ""))
        if unexpected:
            msg.append(f"unexpected ({len(unexpected)}): " + ", ".join(unexpected[:12]) + (" ..." if len(unexpected) > 12 else ""))
        raise KeyError("checkpoint does not match the configured hot-path architecture -- " + " | ".join(msg))
    return True
