"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference (read-only mount at /root/reference).

This module exists only in the build container: /root/reference does not exist on the GPU box,
so nothing under tests/ (-m gpu), bench.py or __graft_entry__.smoke() may import it.  It is used
by oracle/make_golden.py (fixture generation) and oracle/validate_restatement.py (pins
oracle/restate.py against the reference's own nn.Modules).

What it does:
  * installs a sys.meta_path finder that returns MagicMock packages for third-party modules
    the reference imports but that are absent here (torchvision, cv2, apex, mediapipe, ...);
  * puts /root/reference on sys.path and aliases the `EmoPortraits` package name
    (models/stage_1/volumetric_avatar/va_arguments.py:5 imports `EmoPortraits.networks`);
  * builds the argparse namespace = va_arguments defaults + the released launch command
    (experiments/args.txt), as notebooks/infer.py:74-81 would get it from logs/<exp>/args.txt;
  * builds a holder nn.Module with exactly the hot-path sub-networks of
    models/stage_1/volumetric_avatar/va.py:126-279 (same attribute names, same init order:
    weight_init -> spectral norm -> weight standardisation, va.py:86,113-118) and binds
    Model.predict_embed (va.py:813-885).
"""
import argparse
import importlib.abc
import importlib.machinery
import os
import shlex
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = os.environ.get("EMO_REFERENCE_ROOT", "/root/reference")

_STUBBED = {
    "repos", "ibug", "torchvision", "cv2", "skimage", "apex", "mediapipe", "albumentations", "lmdb",
    "wandb", "face_alignment", "facenet_pytorch", "sklearn", "tensorboardX", "pytorch_msssim", "lpips",
    "kornia", "imageio", "dlib", "face_parsing", "face_detection", "matplotlib", "tensorboard",
    "seaborn", "insightface", "onnxruntime", "mmcv", "decord", "av", "ffmpeg", "librosa", "IPython",
}


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        m.__file__ = None
        return m

    def exec_module(self, module):
        return None


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _STUBBED:
            return importlib.machinery.ModuleSpec(name, _StubLoader(), is_package=True)
        return None


_installed = False


def install():
    """Idempotently install the import hook and the reference on sys.path."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; this harness only runs in the build container")
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF_ROOT)
    pkg = types.ModuleType("EmoPortraits")
    pkg.__path__ = [REF_ROOT]
    sys.modules["EmoPortraits"] = pkg
    # torchvision is absent: the embedders' `torchvision.models.resnet*` constructors resolve to the restated
    # architecture in oracle/tv_resnet.py (see its header for what that does and does not pin)
    import torchvision as _tv_stub
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tv_resnet
    _tv_stub.models = tv_resnet
    sys.modules["torchvision.models"] = tv_resnet
    _installed = True


def released_args(image_size=512, overrides=None):
    """Namespace = va_arguments.py:11-357 defaults + tokens of experiments/args.txt (F4 of SURVEY.md)."""
    install()
    from models.stage_1.volumetric_avatar.va_arguments import VolumetricAvatarConfig

    p = argparse.ArgumentParser(conflict_handler="resolve")
    p.add_argument("--num_gpus", default=1, type=int)
    p.add_argument("--image_size", default=256, type=int)
    p.add_argument("--aug_warp_size", default=256, type=int)
    p.add_argument("--num_source_frames", default=1, type=int)
    p.add_argument("--num_target_frames", default=1, type=int)
    p.add_argument("--project_dir", default=REF_ROOT, type=str)
    p = VolumetricAvatarConfig.add_argparse_args(p)
    toks = shlex.split(open(os.path.join(REF_ROOT, "experiments", "args.txt")).read())
    toks = toks[toks.index("../train.py") + 1:]
    args, _unknown = p.parse_known_args(toks)
    args.image_size = image_size
    args.aug_warp_size = image_size
    args.num_gpus = 1
    args.num_source_frames = 1
    args.num_target_frames = 1
    args.print_norms = False
    for k, v in (overrides or {}).items():
        setattr(args, k, v)
    return args


def build_holder(args, nets=("local_encoder_nw", "pose_unsqueeze_nw", "warp_embed_head_orig_nw", "xy_generator_nw",
                             "uv_generator_nw", "volume_source_nw", "volume_process_nw", "decoder_nw"), seed=0):
    """The hot-path subset of va.Model (va.py:41-124,126-279) on a plain holder module."""
    install()
    import torch
    from torch import nn
    import torch.nn.functional as F
    from networks import volumetric_avatar
    from models.stage_1.volumetric_avatar import va as va_mod
    from models.stage_1.volumetric_avatar.va_arguments import VolumetricAvatarConfig
    from utils import weight_init, spectral_norm

    torch.manual_seed(seed)

    class Holder(nn.Module):
        predict_embed = va_mod.Model.predict_embed  # va.py:813-885, unbound

        def __init__(self):
            super().__init__()
            self.args = args
            self.rank = 0
            self.num_source_frames = 1
            self.num_target_frames = 1
            self.embed_size = args.gen_embed_size
            self.pred_mixing = args.gen_pred_mixing
            cfg = VolumetricAvatarConfig(args)
            self.va_config = cfg
            if "local_encoder_nw" in nets:
                self.local_encoder_nw = volumetric_avatar.LocalEncoder(cfg.local_encoder_cfg)            # va.py:133
            if "pose_unsqueeze_nw" in nets:
                self.pose_unsqueeze_nw = nn.Linear(args.lpe_output_channels_expression,
                                                   args.gen_max_channels * self.embed_size ** 2, bias=False)  # va.py:172
            if "warp_embed_head_orig_nw" in nets:
                self.warp_embed_head_orig_nw = nn.Conv2d(args.gen_max_channels * (2 if args.cat_em else 1),
                                                         args.gen_max_channels, (1, 1), bias=False)      # va.py:177
            if "xy_generator_nw" in nets:
                self.xy_generator_nw = volumetric_avatar.WarpGenerator(cfg.warp_generator_cfg)          # va.py:184
            if "uv_generator_nw" in nets:
                self.uv_generator_nw = volumetric_avatar.WarpGenerator(cfg.warp_generator_cfg)          # va.py:185
            if "volume_source_nw" in nets:
                self.volume_source_nw = volumetric_avatar.VPN_ResBlocks(cfg.VPN_resblocks_source_cfg)   # va.py:200
            if "volume_process_nw" in nets:
                self.volume_process_nw = volumetric_avatar.Unet3D(cfg.unet3d_cfg)                       # va.py:212
            if "decoder_nw" in nets:
                self.decoder_nw = volumetric_avatar.Decoder(cfg.decoder_cfg)                            # va.py:226
            self.grid_sample = lambda inputs, grid: F.grid_sample(                                      # va.py:264
                inputs.float(), grid.float(), padding_mode=args.grid_sample_padding_mode)
            grid_s = torch.linspace(-1, 1, args.latent_volume_size)                                      # va.py:101-105
            grid_z = torch.linspace(-1, 1, args.latent_volume_depth)
            w, v, u = torch.meshgrid(grid_z, grid_s, grid_s, indexing="ij")
            e = torch.ones_like(u)
            self.register_buffer("identity_grid_3d", torch.stack([u, v, w, e], dim=3).view(1, -1, 4),
                                 persistent=False)
            self.resize_warp = args.warp_output_size != args.gen_latent_texture_size

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        h = Holder()
        h.apply(weight_init.weight_init(args.init_type, args.init_gain))   # va.py:86
        if args.use_sn:
            spectral_norm.apply_sp_to_nets(h)                               # va.py:113-114
        if args.use_ws:
            volumetric_avatar.utils.apply_ws_to_nets(h)                     # va.py:117-118
    h.eval()
    return h


def randomize_affines(holder, seed=123, scale=0.2):
    """Random-init leaves every norm affine at (1,0), every bias at 0 and the WS-replaced convs at
    torch's default init; perturb them so that parity tests exercise every parameter."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in holder.named_parameters():
            if p.dim() == 1:
                p.add_(scale * torch.randn(p.shape, generator=g))
    return holder


def reference_source_pass(h, source_latents_in, idt_embed, source_pose_embed, theta_src):
    """Replays notebooks/infer.py:433-507 on synthetic inputs (no crop/mask/embedders).
    source_latents_in = masked source image [1,3,S,S]."""
    import torch
    a = h.args
    c, d, s = a.latent_volume_channels, a.latent_volume_depth, a.latent_volume_size
    with torch.no_grad():
        latents = h.local_encoder_nw(source_latents_in)                                   # infer.py:433
        grid = h.identity_grid_3d.repeat_interleave(1, dim=0)                             # infer.py:441
        inv = theta_src.float().inverse().type(theta_src.type())                         # infer.py:443
        source_rotation_warp = grid.bmm(inv[:, :3].transpose(1, 2)).view(-1, d, s, s, 3)  # infer.py:444
        dd = {"source_img": source_latents_in, "target_img": source_latents_in, "idt_embed": idt_embed,
              "source_pose_embed": source_pose_embed, "target_pose_embed": source_pose_embed}
        src_embed, _, _, embed_dict = h.predict_embed(dd)                                 # infer.py:459
        xy_warp = h.xy_generator_nw(src_embed)[0]                                         # infer.py:462
        vol = latents.view(1, c, d, s, s)                                                 # infer.py:485
        vol = h.volume_source_nw(vol)                                                     # infer.py:491
        tv = h.grid_sample(h.grid_sample(vol, source_rotation_warp), xy_warp)             # infer.py:499-500
        canonical = h.volume_process_nw(tv, embed_dict)                                   # infer.py:507
    return dict(latents=latents, source_rotation_warp=source_rotation_warp, xy_warp=xy_warp,
                source_volume=vol, pre_canonical=tv, canonical=canonical, warp_embed=src_embed["orig"])


def reference_driver_pass(h, canonical, idt_embed, target_pose_embed, theta_drv, image_like):
    """Replays notebooks/infer.py:583-637 for ONE driver frame (batch 1, as the reference does, F5)."""
    import torch
    a = h.args
    c, d, s = a.latent_volume_channels, a.latent_volume_depth, a.latent_volume_size
    with torch.no_grad():
        grid = h.identity_grid_3d.repeat_interleave(1, dim=0)                             # infer.py:583
        target_rotation_warp = grid.bmm(theta_drv[:, :3].transpose(1, 2)).view(-1, d, s, s, 3)  # infer.py:586
        dd = {"source_img": image_like, "target_img": image_like, "idt_embed": idt_embed,
              "source_pose_embed": target_pose_embed, "target_pose_embed": target_pose_embed}
        _, tgt_embed, _, embed_dict = h.predict_embed(dd)                                 # infer.py:609
        uv_warp, delta_uv = h.uv_generator_nw(tgt_embed)                                  # infer.py:612
        aligned = h.grid_sample(h.grid_sample(canonical, uv_warp), target_rotation_warp)  # infer.py:618-619
        feat = aligned.view(1, c * d, s, s)                                               # infer.py:627
        img, _, deep_f, img_f = h.decoder_nw(dd, embed_dict, feat, False, stage_two=True)  # infer.py:637
    return dict(target_rotation_warp=target_rotation_warp, uv_warp=uv_warp, delta_uv=delta_uv,
                aligned=aligned, img=img, deep_f=deep_f, img_f=img_f, warp_embed=tgt_embed["orig"])


# ----------------------------------------------------------------------------------------------------------------
# stage 2: notebooks/infer_s2.py + models/stage_2/base/volumetric_avatar_two.py
# ----------------------------------------------------------------------------------------------------------------
def stage2_args(overrides=None):
    """argparse defaults of models/stage_2/base/volumetric_avatar_two.py:26-270 (the released stage-2 args.txt lives in
    logs_s2.zip, which is not in the reference tree) + overrides"""
    install()
    # volumetric_avatar_two.py:20 imports datasets.Retinaface, a file that is absent from the reference tree
    import datasets as _ref_datasets  # the reference's own datasets/ package (REF_ROOT is first on sys.path)
    stub = MagicMock(name="datasets.Retinaface")
    sys.modules.setdefault("datasets.Retinaface", stub)
    if not hasattr(_ref_datasets, "Retinaface"):
        _ref_datasets.Retinaface = stub
    from models.stage_2.base import volumetric_avatar_two as m2
    p = argparse.ArgumentParser(conflict_handler="resolve")
    p.add_argument("--num_gpus", default=1, type=int)
    p.add_argument("--project_dir", default=REF_ROOT, type=str)
    p = m2.Model.add_argparse_args(p)
    args, _ = p.parse_known_args([])
    args.num_gpus = 1
    for k, v in (overrides or {}).items():
        setattr(args, k, v)
    return args


def build_stage2_holder(args, seed=0):
    """local_encoder + decoder of the stage-2 Model (volumetric_avatar_two.py:338-445), initialised and wrapped with
    spectral norm / weight standardisation exactly as Model.__init__ does (:461, :548-575)."""
    install()
    import contextlib
    import io
    import torch
    from torch import nn
    from networks import volumetric_avatar
    from utils import weight_init, spectral_norm, args as args_utils

    torch.manual_seed(seed)

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.args = args
            self.local_encoder = volumetric_avatar.LocalEncoderOld(
                use_amp_autocast=False, gen_upsampling_type=args.gen_upsampling_type,
                gen_downsampling_type=args.gen_downsampling_type, gen_input_image_size=args.output_size_s2,
                gen_latent_texture_size=args.gen_latent_texture_size2,
                gen_latent_texture_depth=args.gen_latent_texture_depth, warp_norm_grad=args.warp_norm_grad,
                gen_num_channels=args.gen_num_channels, enc_channel_mult=args.enc_channel_mult_stage2,
                norm_layer_type=args.norm_layer_type, num_gpus=args.num_gpus, gen_max_channels=args.gen_max_channels,
                enc_block_type=args.enc_block_type, gen_activation_type=args.gen_activation_type,
                gen_latent_texture_channels=args.gen_latent_texture_channels2, in_channels=3)
            self.decoder = volumetric_avatar.Decoder_stage2Old(
                eps=args.eps, image_size=args.output_size_s2, use_amp_autocast=False,
                gen_embed_size=args.gen_embed_size, gen_adaptive_kernel=args.gen_adaptive_kernel,
                gen_adaptive_conv_type=args.gen_adaptive_conv_type,
                gen_latent_texture_size=args.gen_latent_texture_size2,
                in_channels=args.gen_latent_texture_channels2 * args.gen_latent_texture_depth,
                gen_num_channels=args.gen_num_channels, dec_max_channels=args.dec_max_channels2, gen_use_adanorm=False,
                gen_activation_type=args.gen_activation_type, gen_use_adaconv=args.gen_use_adaconv,
                dec_channel_mult=args.dec_channel_mult_stage2, dec_num_blocks=args.dec_num_blocks_stage2,
                dec_up_block_type=args.dec_up_block_type, dec_pred_seg=args.dec_pred_seg,
                dec_seg_channel_mult=args.dec_seg_channel_mult, dec_pred_conf=args.dec_pred_conf,
                dec_conf_ms_names=args.dec_conf_ms_names, dec_conf_names=args.dec_conf_names,
                dec_conf_ms_scales=args.dec_conf_ms_scales, dec_conf_channel_mult=args.dec_conf_channel_mult,
                gen_downsampling_type=args.gen_downsampling_type, num_gpus=args.num_gpus,
                norm_layer_type=args.norm_layer_type)

    with contextlib.redirect_stdout(io.StringIO()):
        h = Holder()
        h.apply(weight_init.weight_init(args.init_type, args.init_gain))                       # :461
        if args.use_sn:                                                                          # :548-561
            spn_layers = args_utils.parse_str_to_list(args.spn_layers, sep=",")
            for name in args_utils.parse_str_to_list(args.spn_networks, sep=","):
                if hasattr(h, name):
                    getattr(h, name).apply(lambda mod: spectral_norm.apply_spectral_norm(mod, apply_to=spn_layers))
        if args.use_ws:                                                                          # :564-575
            for name in args_utils.parse_str_to_list(args.ws_networks, sep=","):
                if hasattr(h, name):
                    setattr(h, name, volumetric_avatar.utils.replace_conv_to_ws_conv(getattr(h, name), conv2d=True, conv3d=True))
    h.eval()
    return h


def randomize_bn_stats(holder, seed=7):
    """eval-mode BatchNorm of a fresh module is the identity (mean 0, var 1): give the running statistics values"""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, b in holder.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.3 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    return holder


def reference_stage2(h, img, mask, face_mask):
    """notebooks/infer_s2.py:351-376 with the third-party masks given (img already at output_size_s2)"""
    import torch
    with torch.no_grad():
        vol = h.local_encoder(img * mask)                                                        # infer_s2.py:370
        add, _, _, _ = h.decoder(None, None, vol, False, pred_feat=None)                         # :371
        add_m = add * (mask * face_mask)                                                         # :365,373
        out = (img + add_m).clamp(max=1, min=0)                                                  # :374-375
    return dict(latents=vol, add=add, out=out)


def make_trained_like(holder, iters=8):
    """bring every spectral-norm (u, v) pair to the dominant singular pair of its weight (the invariant training
    maintains, utils/spectral_norm.py:56-58): with the random unit vectors SpectralNorm.apply leaves behind, W/sigma
    has an arbitrary gain and a BatchNorm network overflows to inf/NaN"""
    import torch
    sd = holder.state_dict()
    with torch.no_grad():
        for k in list(sd):
            if k.endswith(".weight_orig"):
                p = k[: -len(".weight_orig")]
                w = sd[k].reshape(sd[k].shape[0], -1)
                u, v = sd[p + ".weight_u"], sd[p + ".weight_v"]
                for _ in range(iters):
                    v.copy_(torch.nn.functional.normalize(torch.mv(w.t(), u), dim=0))
                    u.copy_(torch.nn.functional.normalize(torch.mv(w, v), dim=0))
    return holder


# ----------------------------------------------------------------------------------------------------------------
# embedders: identity_embedder.py, expression_embedder.py, head_pose_regressor.py
# ----------------------------------------------------------------------------------------------------------------
def build_embedder_holder(args, seed=0):
    """idt_embedder_nw / expression_embedder_nw of va.Model (va.py:161,165) with weight_init -> spectral norm -> weight
    standardisation applied as Model.__init__ does (va.py:86,113-118), plus the HeadPoseRegressor object (va.py:258;
    its external head_pose_regressor.pth is not in the tree, so the resnet18(num_classes=9) keeps a seeded random init)."""
    install()
    import contextlib
    import io
    import torch
    from torch import nn
    from networks import volumetric_avatar
    from models.stage_1.volumetric_avatar.va_arguments import VolumetricAvatarConfig
    from utils import weight_init, spectral_norm
    import tv_resnet

    torch.manual_seed(seed)

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.args = args
            self.rank = 0
            cfg = VolumetricAvatarConfig(args)
            self.idt_embedder_nw = volumetric_avatar.IdtEmbed(cfg.idt_embedder_cfg)                  # va.py:161
            self.expression_embedder_nw = volumetric_avatar.ExpressionEmbed(cfg.exp_embedder_cfg)    # va.py:165

    with contextlib.redirect_stdout(io.StringIO()):
        h = Holder()
        h.apply(weight_init.weight_init(args.init_type, args.init_gain))
        if args.use_sn:
            spectral_norm.apply_sp_to_nets(h)
        if args.use_ws:
            volumetric_avatar.utils.apply_ws_to_nets(h)
        hpr = volumetric_avatar.HeadPoseRegressor.__new__(volumetric_avatar.HeadPoseRegressor)   # head_pose_regressor.py:12-19
        hpr.net = tv_resnet.resnet18(num_classes=9)
        hpr.net.eval()
    h.eval()
    h.head_pose_regressor = hpr      # a plain object, as in the reference (not an nn.Module: no SN / WS / GN swap)
    return h


def reference_idt_embed(h, masked_source):
    """notebooks/infer.py:432"""
    import torch
    with torch.no_grad():
        return h.idt_embedder_nw.forward_image(masked_source)


def reference_head_pose(h, crop):
    """notebooks/infer.py:437 / :562"""
    theta, scale, rotation, translation = h.head_pose_regressor.forward(crop, True)
    return dict(theta=theta, scale=scale, rotation=rotation, translation=translation)


def reference_expression(h, crop, theta):
    """notebooks/infer.py:596-606 (driver side; the source side, :445-455, is the same call on the source crop).  The mask
    entries are unused with use_seg=False."""
    import torch
    dd = {"source_img": crop, "source_mask": None, "source_theta": theta,
          "target_img": crop, "target_mask": None, "target_theta": theta}
    with torch.no_grad():
        dd = h.expression_embedder_nw(dd, True, False)
    return dict(pose_embed=dd["target_pose_embed"], img_align=dd["target_img_align"], align_warp=dd["align_warp"])


# ----------------------------------------------------------------------------------------------------------------
# wrapper host glue: notebooks/infer.py crop_image / get_mixing_theta
# ----------------------------------------------------------------------------------------------------------------
def import_reference_infer():
    """the reference's notebooks/infer.py as a module.  It imports `datasets.voxceleb2hq_pairs`, a file that is absent from
    the reference tree: stubbed (only the dataset class name is taken from it)."""
    install()
    import importlib
    import datasets as _ref_datasets  # noqa: F401  (the reference's own package, REF_ROOT is first on sys.path)
    sys.modules.setdefault("datasets.voxceleb2hq_pairs", MagicMock(name="datasets.voxceleb2hq_pairs"))
    nb = os.path.join(REF_ROOT, "notebooks")
    if nb not in sys.path:
        sys.path.insert(1, nb)
    return importlib.import_module("infer")


def reference_wrapper_stub(image_size, momentum=0.01, fixed_bounding_box=False, mix_old=True):
    """a bare object carrying the attributes InferenceWrapper.crop_image / get_mixing_theta read from `self`
    (notebooks/infer.py:142-150): the methods are called unbound on it, no model is built"""
    m = import_reference_infer()
    stub = types.SimpleNamespace(args=types.SimpleNamespace(image_size=image_size), center=None, size=None,
                                 momentum=momentum, fixed_bounding_box=fixed_bounding_box, mix_old=mix_old,
                                 remove_overflow=m.InferenceWrapper.remove_overflow)
    return m, stub
