"""Driver / source embedders on the HIP kernels -- SURVEY.md section 8f-1.

    reference module (file:line)                                                      here
    IdtEmbed.forward_image      networks/volumetric_avatar/identity_embedder.py:59-87  IdtEmbed
    HeadPoseRegressor.forward   networks/volumetric_avatar/head_pose_regressor.py:21-32 HeadPoseRegressor
    ExpressionEmbed.forward     networks/volumetric_avatar/expression_embedder.py:132-253,
      + ResNetWrapper.forward   :424-459 (inference form of notebooks/infer.py:452,601)  ExpressionEmbed

The backbones are `torchvision.models.resnet18/34/50` (third-party, pinned torchvision==0.9.1, environment.yml:432; not
in the reference tree): `ResNetTrunk` below replays the published architecture -- 7x7/2 stem, 3x3/2 max-pool, four stages
of BasicBlock / Bottleneck (stride on the 3x3, "v1.5"), post-activation residuals -- with torchvision's state_dict key
names, so released checkpoints load unchanged.  What the reference does to those backbones is reproduced at load time:
  * BatchNorm -> GroupNorm(32) when norm_layer_type == 'gn' (utils.py:1020-1038);
  * spectral norm on every Conv2d / Linear of idt_embedder_nw / expression_embedder_nw (utils/spectral_norm.py:12-41),
    folded as sigma = u . W v (eval mode, no power iteration);
  * weight standardisation where `replace_conv_to_ws_conv` hits (utils.py:1061-1096: a Conv2d whose previous or second
    previous sibling is a GroupNorm -> BasicBlock.conv2, Bottleneck.conv2/conv3; those lose SN and gain a bias).
HeadPoseRegressor is a plain torchvision resnet18(num_classes=9) with eval BatchNorm, loaded from its own file.

Kernel schedule per residual block (post-activation, so the norm of a conv output is applied where it is consumed):
    conv (emo_conv2d_generic_f32) -> GroupNorm statistics (emo_groupnorm_affine_f32) -> folded into the next conv's
    staging (+ReLU), or into emo_affine_add_relu_f32 at the block tail / emo_maxpool2d_f32 after the stem.
No torch compute beyond indexing / concatenating 4x4 pose matrices; every op raises if the HIP library is missing.
"""
import math

import torch

from . import ops
from .nets import Norm, _dev
from .pack import fold_sn, fold_ws, pack_generic

RESNET_LAYERS = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet34": ("basic", (3, 4, 6, 3)),
                 "resnet50": ("bottleneck", (3, 4, 6, 3))}
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # identity_embedder.py:56-57, expression_embedder.py:421-422
IMAGENET_STD = (0.229, 0.224, 0.225)

# key: va_arguments.py default (line)
EMBEDDER_DEFAULTS = dict(
    idt_backbone="resnet50",                 # :330
    idt_output_channels=512,                 # :332
    idt_output_size=4,                       # :333
    idt_image_size=256,                      # :241
    lpe_face_backbone="resnet18",            # :337
    lpe_final_pooling_type="avg",            # :339
    lpe_output_channels_expression=512,      # :341
    lpe_output_size=4,                       # :344
    exp_image_size=256,                      # :242
    use_smart_scale=False,                   # :261
    expr_custom_w=False,                     # :146
    norm_layer_type="bn", use_sn=True, use_ws=False,
)
EMBEDDER_RELEASED = dict(norm_layer_type="gn", use_ws=True, lpe_output_channels_expression=128)   # experiments/args.txt


def embedder_config(found=None, overrides=None, released=True):
    cfg = dict(EMBEDDER_DEFAULTS)
    if released:
        cfg.update(EMBEDDER_RELEASED)
    for src in (found or {}, overrides or {}):
        for k, v in src.items():
            if k in cfg:
                cfg[k] = (v is True or v == "True") if isinstance(cfg[k], bool) else type(cfg[k])(v)
    problems = []
    if cfg["norm_layer_type"] not in ("gn", "bn"):
        problems.append("norm_layer_type must be 'gn' or 'bn'")
    if cfg["lpe_final_pooling_type"] != "avg":
        problems.append("lpe_final_pooling_type='transformer' unsupported")
    if cfg["use_smart_scale"] or cfg["expr_custom_w"]:
        problems.append("use_smart_scale / expr_custom_w unsupported")
    for k in ("idt_backbone", "lpe_face_backbone"):
        if cfg[k] not in RESNET_LAYERS:
            problems.append(f"{k}={cfg[k]!r} unsupported")
    if problems:
        raise ValueError("unsupported embedder configuration for the MI355X path: " + "; ".join(problems))
    return cfg


# ---- checkpoint schema ---------------------------------------------------------------------------------------------
def _conv_kind(cfg, block_kind, name):
    """which wrapper the reference leaves on a backbone conv (see module docstring)"""
    ws_hit = {"basic": ("conv2",), "bottleneck": ("conv2", "conv3")}[block_kind]
    if cfg["use_ws"] and cfg["norm_layer_type"] == "gn" and name in ws_hit:
        return "ws"
    return "sn" if cfg["use_sn"] else "plain"


def _conv_keys(out, prefix, kind, cout, cin, k):
    if kind == "sn":
        out[prefix + ".weight_orig"] = (cout, cin, k, k)
        out[prefix + ".weight_u"] = (cout,)
        out[prefix + ".weight_v"] = (cin * k * k,)
    else:
        out[prefix + ".weight"] = (cout, cin, k, k)
        if kind == "ws":
            out[prefix + ".bias"] = (cout,)


def _norm_keys(out, prefix, c, bn):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    if bn:
        out[prefix + ".running_mean"] = (c,)
        out[prefix + ".running_var"] = (c,)
        out[prefix + ".num_batches_tracked"] = ()


def trunk_plan(arch):
    """[(block prefix suffix, cin, planes, stride, has_downsample)] of torchvision's _make_layer"""
    kind, counts = RESNET_LAYERS[arch]
    exp = 1 if kind == "basic" else 4
    plan, cin = [], 64
    for li, nb in enumerate(counts):
        planes = 64 * 2 ** li
        for bi in range(nb):
            stride = 2 if (li > 0 and bi == 0) else 1
            plan.append((f"layer{li + 1}.{bi}", cin, planes, stride, stride != 1 or cin != planes * exp))
            cin = planes * exp
    return kind, exp, plan


def trunk_schema(out, prefix, arch, cfg, wrapped=True):
    """conv1 .. layer4.  wrapped=False: a bare torchvision net (HeadPoseRegressor): plain convs, BatchNorm"""
    bn = (not wrapped) or cfg["norm_layer_type"] == "bn"
    kind, exp, plan = trunk_plan(arch)
    ck = (lambda name: _conv_kind(cfg, kind, name)) if wrapped else (lambda name: "plain")
    stem_kind = ("sn" if cfg["use_sn"] else "plain") if wrapped else "plain"
    _conv_keys(out, prefix + ".conv1", stem_kind, 64, 3, 7)
    _norm_keys(out, prefix + ".bn1", 64, bn)
    for name, cin, planes, stride, down in plan:
        p = f"{prefix}.{name}"
        if kind == "basic":
            _conv_keys(out, p + ".conv1", ck("conv1"), planes, cin, 3)
            _norm_keys(out, p + ".bn1", planes, bn)
            _conv_keys(out, p + ".conv2", ck("conv2"), planes, planes, 3)
            _norm_keys(out, p + ".bn2", planes, bn)
        else:
            _conv_keys(out, p + ".conv1", ck("conv1"), planes, cin, 1)
            _norm_keys(out, p + ".bn1", planes, bn)
            _conv_keys(out, p + ".conv2", ck("conv2"), planes, planes, 3)
            _norm_keys(out, p + ".bn2", planes, bn)
            _conv_keys(out, p + ".conv3", ck("conv3"), planes * 4, planes, 1)
            _norm_keys(out, p + ".bn3", planes * 4, bn)
        if down:
            _conv_keys(out, p + ".downsample.0", stem_kind, planes * exp, cin, 1)
            _norm_keys(out, p + ".downsample.1", planes * exp, bn)
    return 512 * exp


def idt_schema(cfg, prefix="idt_embedder_nw"):
    out = {}
    c = trunk_schema(out, prefix + ".net", cfg["idt_backbone"], cfg)
    _conv_keys(out, prefix + ".net.fc", "sn" if cfg["use_sn"] else "plain", cfg["idt_output_channels"], c, 1)
    return out


def expression_schema(cfg, prefix="expression_embedder_nw"):
    out = {}
    E = cfg["lpe_output_channels_expression"]
    c = trunk_schema(out, prefix + ".net_face.net", cfg["lpe_face_backbone"], cfg)
    _conv_keys(out, prefix + ".net_face.net.fc", "sn" if cfg["use_sn"] else "plain", E, c, 1)
    n_in = E * cfg["lpe_output_size"] ** 2
    if cfg["use_sn"]:
        out[prefix + ".net_face.pose_head.weight_orig"] = (E, n_in)
        out[prefix + ".net_face.pose_head.weight_u"] = (E,)
        out[prefix + ".net_face.pose_head.weight_v"] = (n_in,)
    else:
        out[prefix + ".net_face.pose_head.weight"] = (E, n_in)
    return out


def head_pose_schema():
    out = {}
    trunk_schema(out, "net", "resnet18", None, wrapped=False)
    out = {k[len("net."):]: v for k, v in out.items()}
    out["fc.weight"] = (9, 512)
    out["fc.bias"] = (9,)
    return out


# registered buffers of the reference modules that carry no learned state (constants rebuilt here)
_BUFFERS = (".mean", ".std", ".identity_grid", ".identity_grid_512", ".aligned_keypoints")


def check_state_dict(sd, want, prefix):
    """strict load: every expected tensor present with its shape, nothing unexpected under `prefix`"""
    missing = [k for k in want if k not in sd]
    wrong = [f"{k}: {tuple(sd[k].shape)} != {want[k]}" for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])]
    unexpected = [k for k in sd if k.startswith(prefix) and k not in want and not k.endswith(_BUFFERS)]
    if missing or wrong or unexpected:
        raise KeyError(f"checkpoint does not match the configured '{prefix}' architecture -- missing {missing[:8]} | "
                       f"shape mismatch {wrong[:6]} | unexpected {unexpected[:8]}")
    return True


def random_state_dict(want, seed=0):
    """seeded trained-like tensors for a schema: kaiming weights, unit-ish norm affines, spectral-norm vectors from
    power iteration (random u, v make sigma tiny and the activations explode -- in the reference as well)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in want.items():
        if k.endswith(".num_batches_tracked"):
            sd[k] = torch.tensor(1000)
        elif k.endswith(".running_mean"):
            sd[k] = 0.2 * torch.randn(shape, generator=g)
        elif k.endswith(".running_var"):
            sd[k] = 0.5 + torch.rand(shape, generator=g)
        elif len(shape) > 1:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[k] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif k.endswith((".weight_u", ".weight_v")):
            sd[k] = torch.zeros(shape)
        elif k.endswith(".weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(shape, generator=g)
    for k in list(sd):
        if k.endswith(".weight_orig"):
            p = k[: -len(".weight_orig")]
            w = sd[k].reshape(sd[k].shape[0], -1).double()
            v = torch.randn(w.shape[1], generator=g).double()
            for _ in range(30):
                u = torch.nn.functional.normalize(w @ v, dim=0)
                v = torch.nn.functional.normalize(w.t() @ u, dim=0)
            sd[p + ".weight_u"], sd[p + ".weight_v"] = u.float(), v.float()
    return sd


# ---- executors -----------------------------------------------------------------------------------------------------
class GenericConv:
    """one backbone conv: wrapper folded at load, weight transposed to [K][CoutP] for emo_conv2d_generic_f32"""

    def __init__(self, sd, prefix, stride, pad, device):
        if (prefix + ".weight_orig") in sd:
            w, b = fold_sn(sd[prefix + ".weight_orig"].float(), sd[prefix + ".weight_u"].float(),
                           sd[prefix + ".weight_v"].float()), None
        elif (prefix + ".bias") in sd:
            w, b = fold_ws(sd[prefix + ".weight"].float()), sd[prefix + ".bias"].float()
        else:
            w, b = sd[prefix + ".weight"].float(), None
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.stride, self.pad = stride, pad
        self.wt = pack_generic(w).to(device)
        self.bias = None if b is None else _dev(b, device)

    def __call__(self, x, scale=None, shift=None, relu_in=False):
        return ops.conv2d_generic(x, self.wt, self.cout, self.kh, self.kw, self.stride, self.pad, self.bias, scale, shift,
                                  relu_in)


class _Block:
    def __init__(self, sd, p, kind, stride, down, device):
        self.kind = kind
        if kind == "basic":
            self.convs = [GenericConv(sd, p + ".conv1", stride, 1, device), GenericConv(sd, p + ".conv2", 1, 1, device)]
        else:
            self.convs = [GenericConv(sd, p + ".conv1", 1, 0, device), GenericConv(sd, p + ".conv2", stride, 1, device),
                          GenericConv(sd, p + ".conv3", 1, 0, device)]
        self.norms = [Norm(sd, f"{p}.bn{i + 1}", device) for i in range(len(self.convs))]
        self.down = (GenericConv(sd, p + ".downsample.0", stride, 0, device), Norm(sd, p + ".downsample.1", device)) \
            if down else None

    def __call__(self, x):
        h = self.convs[0](x)
        s, t = self.norms[0].affine(h)
        for conv, norm in zip(self.convs[1:], self.norms[1:]):
            h = conv(h, s, t, relu_in=True)
            s, t = norm.affine(h)
        if self.down is not None:
            d = self.down[0](x)
            sd_, td_ = self.down[1].affine(d)
            return ops.affine_add_relu(h, s, t, d, sd_, td_)
        return ops.affine_add_relu(h, s, t, x)


class ResNetTrunk:
    """conv1 .. layer4 of torchvision's ResNet.forward"""

    def __init__(self, sd, prefix, arch, device):
        kind, exp, plan = trunk_plan(arch)
        self.stem = GenericConv(sd, prefix + ".conv1", 2, 3, device)
        self.stem_norm = Norm(sd, prefix + ".bn1", device)
        self.blocks = [_Block(sd, f"{prefix}.{name}", kind, stride, down, device) for name, _, _, stride, down in plan]
        self.out_channels = 512 * exp

    def __call__(self, x, in_scale=None, in_shift=None):
        c = self.stem(x, in_scale, in_shift, relu_in=False)
        s, t = self.stem_norm.affine(c)
        x = ops.maxpool2d(c, 3, 2, 1, s, t, relu=True)
        for b in self.blocks:
            x = b(x)
        return x


class _ImagenetAffine:
    """(x - mean) / std as the per-(n,c) input affine of the stem conv (applied before its zero padding, like the
    reference's explicit normalisation); constants resident on the device so that calls are graph-capturable"""

    def __init__(self, sd, prefix, device):
        mean = sd.get(prefix + ".mean", torch.tensor(IMAGENET_MEAN)).reshape(-1).double()
        std = sd.get(prefix + ".std", torch.tensor(IMAGENET_STD)).reshape(-1).double()
        self.sc, self.sh = (1.0 / std).float()[None].to(device), (-mean / std).float()[None].to(device)

    def __call__(self, n):
        return self.sc.expand(n, -1).contiguous(), self.sh.expand(n, -1).contiguous()


def _adaptive_avgpool(x, size):
    h, w = x.shape[-2:]
    if (h, w) == (size, size):
        return x
    if h % size or w % size:
        raise RuntimeError(f"adaptive average pool {h}x{w} -> {size}x{size} is not an integer-window pool")
    return ops.avgpool(x, (h // size, w // size))


class IdtEmbed:
    """identity_embedder.py:59-87: bilinear resize to idt_image_size, ImageNet normalisation, trunk, 1x1 `fc` conv,
    adaptive average pool (in that order), mean over the source frames (one frame here)"""

    def __init__(self, sd, cfg, device, prefix="idt_embedder_nw"):
        check_state_dict(sd, idt_schema(cfg, prefix), prefix + ".")
        self.cfg, self.device = cfg, device
        self.in_affine = _ImagenetAffine(sd, prefix, device)
        self.trunk = ResNetTrunk(sd, prefix + ".net", cfg["idt_backbone"], device)
        self.fc = GenericConv(sd, prefix + ".net.fc", 1, 0, device)

    def __call__(self, masked_source):
        S = self.cfg["idt_image_size"]
        x = masked_source.to(self.device).float().contiguous()
        if x.shape[-2:] != (S, S):
            x = ops.resize2d(x, (S, S), "bilinear")
        x = self.fc(self.trunk(x, *self.in_affine(x.shape[0])))
        return _adaptive_avgpool(x, self.cfg["idt_output_size"])

    forward_image = __call__


class HeadPoseRegressor:
    """head_pose_regressor.py:11-32; `state_dict` is the content of args.head_pose_regressor_path"""

    def __init__(self, state_dict, device):
        check_state_dict(state_dict, head_pose_schema(), "")
        sd = {"net." + k: v for k, v in state_dict.items()}
        self.device = device
        self.trunk = ResNetTrunk(sd, "net", "resnet18", device)
        self.w, self.b = _dev(state_dict["fc.weight"], device), _dev(state_dict["fc.bias"], device)

    def forward(self, x, return_srt=False):
        x = x.to(self.device).float().contiguous()
        if x.shape[2] != 128 or x.shape[3] != 128:
            x = ops.resize2d(x, (128, 128), "bilinear")
        f = self.trunk(x)
        B = f.shape[0]
        f = _adaptive_avgpool(f, 1).reshape(B, -1, 1)
        p = ops.add(ops.small_gemm(self.w, f, 1).reshape(B, 9), self.b)                     # fc with bias
        scale, rotation, translation = (p[:, i:i + 3].contiguous() for i in (0, 3, 6))
        theta = ops.pose_theta(scale, rotation, translation)
        return (theta, scale, rotation, translation) if return_srt else theta

    __call__ = forward


class ExpressionEmbed:
    """expression_embedder.py:132-253 as notebooks/infer.py:452,601 calls it (estimate_kp_by_net=True, use_seg=False, eval)
    + ResNetWrapper.forward :441-459.  The reference feeds cat(source, target) = the same crop twice and keeps one half;
    every op is per sample, so one copy is computed."""

    def __init__(self, sd, cfg, device, prefix="expression_embedder_nw"):
        check_state_dict(sd, expression_schema(cfg, prefix), prefix + ".")
        self.cfg, self.device, self.prefix = cfg, device, prefix + ".net_face"
        self.in_affine = _ImagenetAffine(sd, self.prefix, device)
        self.trunk = ResNetTrunk(sd, self.prefix + ".net", cfg["lpe_face_backbone"], device)
        self.fc = GenericConv(sd, self.prefix + ".net.fc", 1, 0, device)
        p = self.prefix + ".pose_head"
        w = fold_sn(sd[p + ".weight_orig"].float(), sd[p + ".weight_u"].float(), sd[p + ".weight_v"].float()) \
            if (p + ".weight_orig") in sd else sd[p + ".weight"].float()
        self.w_head = _dev(w, device)
        self.grid_size = cfg["exp_image_size"] // 2                                        # expression_embedder.py:87
        self.zoom = torch.diag(torch.tensor([0.5, 0.5, 1.0])).to(device)                   # :196-198
        self.keep = torch.tensor([0, 1, 3], device=device)                                 # rows / cols of the 2-D transform
        self.last_row = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]], device=device)

    def align_theta(self, theta):
        """:178-200: 4x4 inverse of the head pose, rows/cols (0,1,3) -> 2-D affine, 2x zoom-in, first two rows"""
        t4 = theta.to(self.device).float()
        t4 = torch.cat([t4[:, :3], self.last_row.expand(t4.shape[0], -1, -1)], dim=1)      # :183-187 (theta_[:, :3] + e4)
        inv2d = ops.mat4_inverse(t4.contiguous()).index_select(2, self.keep).index_select(1, self.keep)
        return torch.matmul(inv2d, self.zoom)[:, :2].contiguous()

    def forward(self, crop, theta, want_aligned=False):
        crop = crop.to(self.device).float().contiguous()
        a = self.align_theta(theta)
        aligned, warp = ops.grid_sample2d(crop, theta=a, size=self.grid_size, want_grid=True)   # :221-231
        B = aligned.shape[0]
        x = self.fc(self.trunk(aligned, *self.in_affine(B)))
        x = _adaptive_avgpool(x, self.cfg["lpe_output_size"]).reshape(B, -1, 1)
        pose = ops.small_gemm(self.w_head, x, 1).reshape(B, -1)
        return (pose, aligned, warp) if want_aligned else pose

    __call__ = forward
