"""Seeded random checkpoints with the reference's exact key layout (emoportraits_amd/schema.py), for the benchmark,
smoke test and full-size parity tests: the released weights are not in the repo (README.md:125-139, Google Drive).

`trained_like=True` (default) makes the tensors behave like a trained checkpoint rather than like the reference's
raw initialisation: spectral-norm vectors u, v are brought to the dominant singular pair with a few power
iterations (the invariant training maintains, utils/spectral_norm.py:56-58), so W/sigma has spectral norm ~1
instead of the arbitrary gain a random (u, v) pair gives; norm affines are near (1, 0).  With
`trained_like=False`, u and v are random unit vectors exactly as SpectralNorm.apply leaves them
(utils/spectral_norm.py:203-205) -- activations then reach 1e8 and rounding differences are chaotically
amplified, which is what BASELINE config 1 ("random-init weights") exercises.
"""
import math

import torch

from .schema import hot_path_schema


def _kaiming(shape, g):
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * math.sqrt(2.0 / max(fan_in, 1))


def random_state_dict(cfg, seed=0, trained_like=True, with_source=True, image_head_gain=None, warp_head_gain=None):
    """image_head_gain: scales the affine of the last GroupNorm of the image decoder (dec_img_head.0).  The head is a
    weight-standardised 1x1 conv over 128 ReLU channels followed by a sigmoid: with unit norm weights its pre-activation has
    a standard deviation of ~8, i.e. a random network paints saturated 0/1 images and every rounding difference that moves a
    pre-activation across zero flips a pixel.  A trained decoder produces natural images: logits within a few units.  A gain
    of ~0.2 gives the seeded checkpoint that statistic (pre-activation std ~1.5) without touching anything else.

    warp_head_gain: scales the affine of the GroupNorm in front of the tanh head of both WarpGenerators (pre_head.0) and the
    head's bias.  The head
    is a spectrally normalised 3x3x3 conv over 32 ReLU channels: with unit norm weights its pre-activation has a standard
    deviation of ~0.6, i.e. a random generator emits deltas of +-1 in normalised coordinates -- 32 voxels -- and four fifths
    of the sample points leave the volume.  A trained generator emits the small expression / canonicalisation offsets the
    volumes were trained with: a gain of ~0.02 keeps |delta| below one voxel (0.03 in x, y)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    schema = hot_path_schema(cfg, with_source)
    for k, shape in schema.items():
        if k.endswith(".identity_grid"):
            d, s = shape[2], shape[3]
            gs, gz = torch.linspace(-1, 1, s), torch.linspace(-1, 1, d)
            w, v, u = torch.meshgrid(gz, gs, gs, indexing="ij")
            sd[k] = torch.stack([u, v, w], 0)[None].contiguous()          # warp_generator_resnet.py:121-124
        elif k.endswith(".weight_orig") or (k.endswith(".weight") and len(shape) > 1):
            sd[k] = _kaiming(shape, g)
        elif k.endswith(".input_tensor"):
            sd[k] = torch.randn(shape, generator=g)                       # unet_3d.py:99
        elif ".projector.u." in k:
            a = math.sqrt(3.0 / shape[1])
            sd[k] = (torch.rand(shape, generator=g) * 2 - 1) * a          # utils.py:1130-1133
        elif ".projector.v." in k:
            a = math.sqrt(3.0 / shape[0])
            sd[k] = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif k.endswith(".weight_u") or k.endswith(".weight_v"):
            v = torch.randn(shape, generator=g)
            sd[k] = v / v.norm().clamp_min(1e-12)
        elif k.endswith(".weight"):                                       # norm gamma
            sd[k] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.1 * torch.randn(shape, generator=g)
        else:
            raise AssertionError("unclassified key " + k)
    if trained_like:
        for k in list(sd):
            if k.endswith(".weight_orig"):
                p = k[: -len(".weight_orig")]
                w = sd[k].reshape(sd[k].shape[0], -1)
                u, v = sd[p + ".weight_u"], sd[p + ".weight_v"]
                for _ in range(8):
                    v = torch.mv(w.t(), u)
                    v = v / v.norm().clamp_min(1e-12)
                    u = torch.mv(w, v)
                    u = u / u.norm().clamp_min(1e-12)
                sd[p + ".weight_u"], sd[p + ".weight_v"] = u, v
    if warp_head_gain is not None:
        for net in ("uv_generator_nw", "xy_generator_nw"):
            for k in (net + ".pre_head.0.weight", net + ".pre_head.0.bias", net + ".head.0.0.bias"):
                if k in sd:
                    sd[k] = sd[k] * float(warp_head_gain)
    if image_head_gain is not None:
        for k in ("decoder_nw.img_decoder.dec_img_head.0.weight", "decoder_nw.img_decoder.dec_img_head.0.bias"):
            sd[k] = sd[k] * float(image_head_gain)
    return sd


def trained_like_state_dict(cfg, seed=0, with_source=True):
    """The seeded checkpoint every full-size tolerance and the benchmark are characterised on (the released weights are not
    obtainable here, README.md:125-139): spectral norms ~1, norm affines near (1, 0) with a 10 % spread, activations O(1)..O(50)
    at every stage, predicted warps within one voxel of the identity, unsaturated image (logit std ~1.5)."""
    return random_state_dict(cfg, seed=seed, trained_like=True, with_source=with_source, image_head_gain=0.2, warp_head_gain=0.01)
