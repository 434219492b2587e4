"""Stage-2 refinement on the HIP kernels -- SURVEY.md section 8f-2 (BASELINE.json configs[4]).

Reference: notebooks/infer_s2.py:351-376 on models/stage_2/base/volumetric_avatar_two.py:338-445 --
`LocalEncoderOld` (networks/volumetric_avatar/local_encoder_old.py:25-117: 7x7 conv, stride-2 ResBlocks, norm+ReLU+1x1)
-> `Decoder_stage2` / `ImageDecoder_stage2` (decoder_s2_old.py:18-218, :346-472: 1x1, ResBlocks at the latent size,
nearest-x2 ResBlocks, 128/64/32-channel ResBlocks at full size, norm+ReLU+1x1+tanh) -> residual image added under
the (matte x face) mask and clamped.  Same ResBlock / conv / norm kernels as stage 1; BatchNorm (the stage-2 default,
volumetric_avatar_two.py:33) is an eval-mode static affine folded into the conv staging at load time.

The released stage-2 args.txt / checkpoint live in logs_s2.zip (README.md:125-139), not in the reference tree: the config
defaults below are the argparse defaults of volumetric_avatar_two.py:26-270; both norm variants ('bn', and 'gn' with
weight standardisation as stage 1 uses) are supported and pinned against the reference's modules
(oracle/validate_restatement.py, tests/golden/tiny_stage2.pt).
"""
import math
import pathlib
from argparse import Namespace

import torch

from . import config as cfg_mod
from . import ops, parallel
from .encoder import LocalEncoder
from .nets import Norm, ResBlock
from .pack import PackedConv

STAGE2_DEFAULTS = dict(   # volumetric_avatar_two.py line of each flag
    output_size_s2=512,                 # :189
    gen_latent_texture_size2=64,        # :192
    gen_latent_texture_channels2=64,    # :195
    gen_latent_texture_depth=16,        # :193
    gen_num_channels=32,                # :160
    gen_max_channels=512,               # :161
    enc_channel_mult_stage2=4.0,        # :180
    dec_channel_mult_stage2=4.0,        # :179
    dec_num_blocks_stage2=8,            # :178
    dec_max_channels2=512,              # :59
    norm_layer_type="bn",               # :33
    use_ws=False,                       # :130
    use_sn=True,                        # :129
)


def stage2_config(found=None, overrides=None):
    cfg = dict(STAGE2_DEFAULTS)
    for src in (found or {}, overrides or {}):
        for k, v in src.items():
            if k in cfg:
                cfg[k] = (v is True or v == "True") if isinstance(cfg[k], bool) else type(cfg[k])(v)
    if cfg["norm_layer_type"] not in ("bn", "gn"):
        raise ValueError("stage 2 on the MI355X path supports norm_layer_type 'bn' or 'gn'")
    if not cfg["use_sn"]:
        raise ValueError("use_sn must be True (released key layout)")
    return cfg


def _uses_ws(cfg):
    return bool(cfg["use_ws"]) and cfg["norm_layer_type"] == "gn"


def decoder_channels(cfg):
    nup = int(math.log(cfg["output_size_s2"] // cfg["gen_latent_texture_size2"], 2))
    trunk = min(int(cfg["gen_num_channels"] * cfg["dec_channel_mult_stage2"] * 2 ** nup), cfg["dec_max_channels2"])
    ups, c = [], trunk
    for _ in range(nup - 1):
        c = max(c // 2, cfg["gen_num_channels"])
        ups.append(c)
    return trunk, ups


# ---- checkpoint schema (strict loading) ---------------------------------------------------------------------------
def _norm_keys(out, prefix, c, bn):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    if bn:
        out[prefix + ".running_mean"] = (c,)
        out[prefix + ".running_var"] = (c,)
        out[prefix + ".num_batches_tracked"] = ()


def _conv_keys(out, prefix, kind, cout, cin, k, bias):
    if kind == "sn":
        out[prefix + ".weight_orig"] = (cout, cin, k, k)
        out[prefix + ".weight_u"] = (cout,)
        out[prefix + ".weight_v"] = (cin * k * k,)
        if bias:
            out[prefix + ".bias"] = (cout,)
    else:
        out[prefix + ".weight"] = (cout, cin, k, k)
        out[prefix + ".bias"] = (cout,)


def _res_block_keys(out, prefix, cin, cout, kind, bn):
    _norm_keys(out, prefix + ".block_feats.0", cin, bn)
    _conv_keys(out, prefix + ".block_feats.2", kind, cout, cin, 3, False)
    _norm_keys(out, prefix + ".block_feats.3", cout, bn)
    _conv_keys(out, prefix + ".block.0", "sn", cout, cout, 3, False)
    if cin != cout:
        _conv_keys(out, prefix + ".skip.0", "sn", cout, cin, 1, False)


def stage2_schema(cfg):
    out = {}
    bn = cfg["norm_layer_type"] == "bn"
    kind = "ws" if _uses_ws(cfg) else "sn"
    S = cfg["output_size_s2"]
    c = int(cfg["gen_num_channels"] * cfg["enc_channel_mult_stage2"])
    p = "local_encoder"
    _conv_keys(out, f"{p}.from_rgb_{S}px", "sn", c, 3, 7, True)
    s = S
    for i in range(int(math.log(S // cfg["gen_latent_texture_size2"], 2))):
        c2 = min(c * 2, cfg["gen_max_channels"])
        _res_block_keys(out, f"{p}.enc_{i}_block={s}px", c, c2, kind, bn)
        c, s = c2, s // 2
    _norm_keys(out, p + ".finale_layers.0", c, bn)
    cd = cfg["gen_latent_texture_channels2"] * cfg["gen_latent_texture_depth"]
    _conv_keys(out, p + ".finale_layers.2", kind, cd, c, 1, True)
    trunk, ups = decoder_channels(cfg)
    p = "decoder"
    _conv_keys(out, p + ".res_decoder.0", "sn", trunk, cd, 1, False)
    for i in range(cfg["dec_num_blocks_stage2"]):
        _res_block_keys(out, f"{p}.res_decoder.{i + 1}", trunk, trunk, kind, bn)
    c = trunk
    for i, u in enumerate(ups):
        _res_block_keys(out, f"{p}.img_decoder.dec_img_blocks.{i}", c, u, kind, bn)
        c = u
    for i, u in enumerate((128, 128, 64, 32)):                           # decoder_s2_old.py:391-417
        _res_block_keys(out, f"{p}.img_decoder.dec_img_feat_blocks.{i}", c, u, kind, bn)
        c = u
    _norm_keys(out, p + ".img_decoder.dec_img_head.0", c, bn)
    _conv_keys(out, p + ".img_decoder.dec_img_head.2", kind, 3, c, 1, True)
    return out


def check_state_dict(sd, cfg):
    want = stage2_schema(cfg)
    missing = [k for k in want if k not in sd]
    wrong = [f"{k}: {tuple(sd[k].shape)} != {want[k]}" for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])]
    unexpected = [k for k in sd if k.startswith(("local_encoder.", "decoder.")) and k not in want]
    if missing or wrong or unexpected:
        raise KeyError("stage-2 checkpoint does not match the configured architecture -- "
                       f"missing {missing[:8]} | shape mismatch {wrong[:6]} | unexpected {unexpected[:8]}")
    return True


def random_state_dict(cfg, seed=0):
    """seeded trained-like checkpoint in the stage-2 key layout (see emoportraits_amd/random_init.py)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in stage2_schema(cfg).items():
        if k.endswith(".num_batches_tracked"):
            sd[k] = torch.tensor(1000)
        elif k.endswith(".running_mean"):
            sd[k] = 0.2 * torch.randn(shape, generator=g)
        elif k.endswith(".running_var"):
            sd[k] = 0.5 + torch.rand(shape, generator=g)
        elif k.endswith(".weight_orig") or (k.endswith(".weight") and len(shape) > 1):
            fan_in = shape[1] * shape[2] * shape[3]
            sd[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif k.endswith(".weight_u") or k.endswith(".weight_v"):
            v = torch.randn(shape, generator=g)
            sd[k] = v / v.norm()
        elif k.endswith(".weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(shape, generator=g)
    for k in list(sd):
        if k.endswith(".weight_orig"):
            p = k[: -len(".weight_orig")]
            w = sd[k].reshape(sd[k].shape[0], -1)
            u, v = sd[p + ".weight_u"], sd[p + ".weight_v"]
            for _ in range(8):
                v = torch.nn.functional.normalize(torch.mv(w.t(), u), dim=0)
                u = torch.nn.functional.normalize(torch.mv(w, v), dim=0)
            sd[p + ".weight_u"], sd[p + ".weight_v"] = u, v
    return sd


# ---- executors ----------------------------------------------------------------------------------------------------
class DecoderStage2:
    """decoder_s2_old.py: Decoder_stage2 + ImageDecoder_stage2 -> residual image in [-1, 1] (tanh)"""

    def __init__(self, sd, prefix, cfg, device):
        kind = "ws" if _uses_ws(cfg) else "sn"
        _, ups = decoder_channels(cfg)
        self.first = PackedConv.from_state_dict(sd, prefix + ".res_decoder.0", "sn", device)
        self.trunk = [ResBlock(sd, f"{prefix}.res_decoder.{i + 1}", kind, device) for i in range(cfg["dec_num_blocks_stage2"])]
        self.up = [ResBlock(sd, f"{prefix}.img_decoder.dec_img_blocks.{i}", kind, device) for i in range(len(ups))]
        self.feat = [ResBlock(sd, f"{prefix}.img_decoder.dec_img_feat_blocks.{i}", kind, device) for i in range(4)]
        self.nh = Norm(sd, prefix + ".img_decoder.dec_img_head.0", device)
        self.head = PackedConv.from_state_dict(sd, prefix + ".img_decoder.dec_img_head.2", kind, device)

    def __call__(self, feat_2d):
        # GroupNorm variant: the tile statistics of every conv output travel with the tensor (nets.ResBlock); the BatchNorm
        # default has static affines and asks for none
        x, st = (ops.conv_igemm(feat_2d, self.first), None) if self.nh.bn else ops.conv_igemm(feat_2d, self.first, want_stats=True)
        for b in self.trunk:
            x, st = b(x, x_stats=st, want_stats=True)
        for b in self.up:
            x, st = b(x, ups=True, x_stats=st, want_stats=True)
        x, st = self.feat[0](x, ups=True, x_stats=st, want_stats=True)
        for b in self.feat[1:]:
            x, st = b(x, x_stats=st, want_stats=True)
        s, h = self.nh.affine(x, stats=st)
        return ops.conv_igemm(x, self.head, s, h, relu_in=True, act="tanh")


class Stage2:
    def __init__(self, state_dict, cfg, device="cuda:0", precision=None):
        """precision 'f16': the "fp16 MFMA convs" mode of BASELINE.json configs[4] (fp16 operands, fp32 accumulation,
        fp32 tensors); 'f32': exact-fp32 MFMA everywhere; 'bf16x3': fp32 on the bf16 matrix pipes in the 3x3 layers
        (csrc/conv_igemm_bf16x3.h); 'f16x2': the same layers as the device-checked two-term fp16 split with guarded bf16x3
        recomputation (nets.DEFAULT_PRECISION).  None: EMO_CONV_PRECISION, else nets.DEFAULT_PRECISION (as nets.HotPath)."""
        import os
        from . import nets
        from .pack import conv_precision
        self.cfg = cfg
        self.device = torch.device(device)
        if precision is None:
            precision = os.environ.get("EMO_CONV_PRECISION", nets.DEFAULT_PRECISION)
        self.precision = precision
        with conv_precision(precision):
            self.encoder = LocalEncoder(state_dict, "local_encoder", cfg, self.device, image_size=cfg["output_size_s2"],
                                        latent_size=cfg["gen_latent_texture_size2"], ws=_uses_ws(cfg))
            self.decoder = DecoderStage2(state_dict, "decoder", cfg, self.device)

    def refine(self, img, mask, face_mask, keep=False):
        """img [B,3,S2,S2] in [0,1] (stage-1 output at output_size_s2), mask = matte [B,1,S2,S2], face_mask [B,1,S2,S2]
        -> clamp(img + decoder(encoder(img*mask)) * (mask*face_mask), 0, 1)     (infer_s2.py:365-375)"""
        if self.precision == "f16x2":
            ops.clear_overflow_flags(self.device)       # (range-check words of the fp16-split layers: nets.HotPath._clear_flags)
        lat = self.encoder(ops.mul_mask(img, mask))
        add = self.decoder(lat)
        out = ops.stage2_compose(img, add, mask, face_mask)
        if keep:
            return dict(latents=lat, add=add, out=out)
        return out


class InferenceWrapper:
    """Mirror of notebooks/infer_s2.py:InferenceWrapper (:52-54 constructor, :351-376 forward) for the stage-2 model.
    The MODNet matte and the BiSeNet face mask are third-party nets outside the hot path: pass them as `embedders`
    {'matting': img -> [B,1,H,W], 'face_parsing': img -> [B,1,H,W]} or give the masks to forward()."""

    def __init__(self, experiment_name, which_epoch='latest', model_file_name='', use_gpu=True, num_gpus=1,
                 fixed_bounding_box=False, project_dir='./', torch_home='', debug=False, print_model=False,
                 args_overwrite={}, pose_momentum=0.5, experiment_name_s1=None, model_file_name_s1=None, cloth=False,
                 state_dict=None, args_path=None, embedders=None, precision=None):
        if not use_gpu:
            raise RuntimeError("emoportraits_amd runs on MI355X only: use_gpu=False is not supported (no CPU path)")
        self.cloth = cloth
        args_path = pathlib.Path(project_dir) / 'logs_s2' / experiment_name / 'args.txt' if args_path is None else args_path
        found = cfg_mod.parse_args_txt(args_path)                                  # infer_s2.py:64-66
        found['project_dir'] = project_dir
        for k, v in (args_overwrite or {}).items():
            found[k] = v
        self.args = Namespace(**found)
        self.cfg = stage2_config(found)
        if num_gpus > 1:
            self.rank, self.world = parallel.init_distributed()
        else:
            self.rank, self.world = 0, 1
        self.device = torch.device("cuda", parallel.local_device_index())
        torch.cuda.set_device(self.device)
        self.model_checkpoint_s2 = pathlib.Path(project_dir) / 'logs_s2' / experiment_name / 'checkpoints' / model_file_name
        self.model_dict_s2 = torch.load(self.model_checkpoint_s2, map_location='cpu') if state_dict is None else state_dict
        check_state_dict(self.model_dict_s2, self.cfg)                              # the reference loads strict=False (:116)
        self.model_two = Stage2(self.model_dict_s2, self.cfg, self.device, precision=precision)   # 'f16': configs[4] mode
        self.embedders = dict(embedders or {})

    def forward(self, img, cloth=False, mask=None, face_mask=None):
        """-> (pred_target_img, pred_target_img_resized, pred_target_img_ffhq, mask) like infer_s2.py:377-387, with
        images as uint8 [B,H,W,3] device tensors instead of PIL lists (pack on the device, one D2H when needed)."""
        S2 = self.cfg["output_size_s2"]
        img = img.to(self.device).float().contiguous()
        img0 = img                                                                 # returned un-resized (infer_s2.py:377-378)
        if img.shape[-1] != S2 or img.shape[-2] != S2:
            img = ops.resize2d(img, (S2, S2), "bilinear")                          # infer_s2.py:360-362
        if mask is None:
            if 'matting' not in self.embedders:
                raise RuntimeError("stage 2 needs the MODNet matte: pass mask= or embedders={'matting': fn}")
            mask = self.embedders['matting'](img)
        if face_mask is None:
            if cloth or self.cloth:
                face_mask = torch.ones_like(mask)                                    # infer_s2.py:366-368
            elif 'face_parsing' in self.embedders:
                face_mask = self.embedders['face_parsing'](img)
            else:
                raise RuntimeError("stage 2 needs the face-parsing mask: pass face_mask= or embedders={'face_parsing': fn}")
        mask = mask.to(self.device).float().contiguous()
        face_mask = face_mask.to(self.device).float().contiguous()
        out = self.model_two.refine(img, mask, face_mask)
        resized_u8 = ops.pack_rgb8(img)
        first_u8 = resized_u8 if img0 is img else ops.pack_rgb8(img0)
        return first_u8, resized_u8, ops.pack_rgb8(out), mask

    __call__ = forward
