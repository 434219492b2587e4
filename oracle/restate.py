"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of the EMOPortraits volumetric-avatar
inference hot path.  Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module, and only as the checker / the reported baseline.

It re-states, as plain functional torch-CPU fp32 code over a *raw reference state_dict*
(the exact key layout `torch.save(model.state_dict())` of the reference produces), what the
reference's nn.Modules compute.  Every function cites the reference file:line it follows
(paths relative to the reference root).  The arithmetic of conv / group_norm / interpolate /
avg_pool / grid_sample lives in PyTorch ATen (third-party to the reference, pinned
pytorch=1.13.1 in environment.yml:173; 2.10 here -- semantics unchanged for these ops), and is
called here through torch's CPU kernels; the 3-D sampler additionally has a from-first-principles
restatement in `grid_sample3d_restated` (numpy) and oracle/grid_sample3d.c (plain C), both
pinned bit-exactly against torch's CPU `F.grid_sample`.

Pinning: the reference ships no tests / golden vectors for this path, so this
oracle is pinned against outputs of the reference's own modules run in the build container
(oracle/validate_restatement.py; fixtures in tests/golden/ made by oracle/make_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# configuration (subset of models/stage_1/volumetric_avatar/va_arguments.py:11-357 that shapes
# the hot path; values = released launch command experiments/args.txt)
# ----------------------------------------------------------------------------------------------
RELEASED_CFG = dict(
    image_size=512,
    latent_volume_channels=96, latent_volume_depth=16, latent_volume_size=64,
    gen_latent_texture_channels=96, gen_latent_texture_depth=16, gen_latent_texture_size=64,
    gen_num_channels=32, gen_max_channels=512, gen_max_channels_unet3d=512, enc_channel_mult=4.0,
    gen_embed_size=4, gen_dummy_input_size=8, warp_output_size=64, warp_channel_mult=1.0,
    source_volume_num_blocks=3,
    dec_num_blocks=6, dec_channel_mult=2.0, dec_max_channels=512,
    im_dec_num_lrs_per_resolution=2, im_dec_ch_div_factor=1.5,
    lpe_output_channels_expression=128, local_encoder_input_size=3,
    grid_sample_padding_mode="zeros",
)


def cfg_from_args(args):
    """Pick the hot-path keys out of an argparse Namespace (reference or product side)."""
    return {k: getattr(args, k) for k in RELEASED_CFG}


# ----------------------------------------------------------------------------------------------
# weight derivation: spectral norm and weight standardisation are recomputed per forward by the
# reference in eval mode; both are pure functions of the stored tensors.
# ----------------------------------------------------------------------------------------------
def sn_weight(sd, prefix):
    """utils/spectral_norm.py:96-168 with do_power_iteration=False (eval): W / (u . (W_mat v))."""
    w = sd[prefix + ".weight_orig"]
    u = sd[prefix + ".weight_u"]
    v = sd[prefix + ".weight_v"]
    w_mat = w.reshape(w.shape[0], -1)                       # spectral_norm.py:84-94, dim=0
    sigma = torch.dot(u, torch.mv(w_mat, v))                # spectral_norm.py:163
    return w / sigma                                        # spectral_norm.py:165


def ws_weight(w):
    """networks/volumetric_avatar/utils.py:893-900 (Conv2d_ws) / :908-914 (Conv3d_ws)."""
    m = w
    for dim in range(1, w.dim()):
        m = m.mean(dim=dim, keepdim=True)
    w = w - m
    std = w.view(w.size(0), -1).std(dim=1).view(-1, *([1] * (w.dim() - 1))) + 1e-5
    return w / std.expand_as(w)


def conv_params(sd, prefix, kind):
    """kind: 'sn' (weight_orig/u/v [+bias]), 'ws' (weight+bias, standardised), 'plain'."""
    if kind == "sn":
        w = sn_weight(sd, prefix)
    elif kind == "ws":
        w = ws_weight(sd[prefix + ".weight"])
    elif kind == "plain":
        w = sd[prefix + ".weight"]
    else:
        raise ValueError(kind)
    return w, sd.get(prefix + ".bias", None)


def conv(x, sd, prefix, kind, padding=0):
    w, b = conv_params(sd, prefix, kind)
    if w.dim() == 4:
        return F.conv2d(x, w, b, padding=padding)
    return F.conv3d(x, w, b, padding=padding)


# ----------------------------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------------------------
def group_norm(x, sd, prefix, groups=32, eps=1e-5):
    """norm_layers['gn'|'gn_3d'] = nn.GroupNorm(32, C) (networks/volumetric_avatar/utils.py:953-957);
    norm_layers['bn'] = nn.BatchNorm2d (utils.py:948) in eval mode when the checkpoint carries running statistics
    (stage 2 with its default flags, models/stage_2/base/volumetric_avatar_two.py:33)."""
    if (prefix + ".running_mean") in sd:
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                            sd[prefix + ".bias"], False, 0.0, eps)
    return F.group_norm(x, groups, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def ada_group_norm(x, sd, prefix, ada, groups=32, eps=1e-5):
    """AdaptiveGroupNorm.forward (utils.py:302-325) with the parameters assigned by
    assign_adaptive_norm_params (utils.py:983-995): gamma = weight[None] + d_gamma, beta likewise."""
    d_gamma, d_beta = ada
    # Reference quirk, reproduced on purpose: AdaptiveGroupNorm.__init__ (utils.py:303-308) builds the
    # nn.GroupNorm base with affine=False but then assigns self.weight / self.bias, so
    # nn.GroupNorm.forward (called at utils.py:312) DOES apply them: the static affine is applied
    # twice -- once inside the base group_norm and once more through ada_weight = weight + d_gamma.
    y = F.group_norm(x, groups, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)
    gamma = sd[prefix + ".weight"][None] + d_gamma          # [B,C]
    beta = sd[prefix + ".bias"][None] + d_beta
    shape = [x.shape[0], x.shape[1]] + [1] * (x.dim() - 2)
    return y * gamma.view(shape) + beta.view(shape)


def projector_norm(sd, prefix, n_norms, embed_orig):
    """ProjectorNorm.forward (utils.py:1137-1151): u[C,512] @ orig[B,512,16] @ v[16,2] -> (gamma,beta)."""
    out = []
    for i in range(n_norms):
        u = sd[f"{prefix}.u.{i}"]
        v = sd[f"{prefix}.v.{i}"]
        p = u[None].matmul(embed_orig).matmul(v[None])      # [B,C,2]
        out.append((p[..., 0], p[..., 1]))
    return out


# ----------------------------------------------------------------------------------------------
# ResBlock (networks/volumetric_avatar/utils.py:661-788)
# ----------------------------------------------------------------------------------------------
def res_block(x, sd, prefix, first_kind, upsample=None, downsample=None, ada=None):
    """block_feats = [norm, relu, conv(k3,p1), norm, relu]; block = [conv(k3,p1) (, avgpool)];
    skip = [conv(k1) if Cin!=Cout (, avgpool)]  (utils.py:711-763).
    first_kind: 'ws' where the WS replacement rule hit block_feats.2 (utils.py:1061-1096: Conv2d after
    nn.GroupNorm, Conv3d after AdaptiveGroupNorm), else 'sn'.  The second conv and the skip conv are
    child #0 of their Sequential and therefore always keep spectral norm.
    upsample: None or a callable applied to BOTH main and skip inputs before anything else
    (utils.py:764-781, efficient_upsampling=False).  downsample: None or callable (avg-pool) applied
    after the main conv and after the skip conv (utils.py:744-760).
    ada: None, or ((dg1,db1),(dg2,db2)) for the two AdaptiveGroupNorms."""
    inputs = x
    out = upsample(x) if upsample is not None else x
    if ada is None:
        h = group_norm(out, sd, prefix + ".block_feats.0")
    else:
        h = ada_group_norm(out, sd, prefix + ".block_feats.0", ada[0])
    h = F.relu(h)
    h = conv(h, sd, prefix + ".block_feats.2", first_kind, padding=1)
    if ada is None:
        h = group_norm(h, sd, prefix + ".block_feats.3")
    else:
        h = ada_group_norm(h, sd, prefix + ".block_feats.3", ada[1])
    h = F.relu(h)
    main = conv(h, sd, prefix + ".block.0", "sn", padding=1)
    if downsample is not None:
        main = downsample(main)
    skip = upsample(inputs) if upsample is not None else inputs
    if (prefix + ".skip.0.weight_orig") in sd:
        skip = conv(skip, sd, prefix + ".skip.0", "sn")
    if downsample is not None:
        skip = downsample(skip)
    return main + skip


# ----------------------------------------------------------------------------------------------
# a3: head-pose affine (utils/point_transforms.py:188-242)
# ----------------------------------------------------------------------------------------------
def get_transform_matrix(scale, rotation, translation):
    b = scale.shape[0]
    eye = torch.eye(4, dtype=scale.dtype)[None].repeat_interleave(b, dim=0)
    S = eye.clone()
    if scale.shape[1] == 3:                                  # point_transforms.py:199-206
        S[:, 0, 0], S[:, 1, 1], S[:, 2, 2] = scale[:, 0], scale[:, 1], scale[:, 2]
    else:
        S[:, 0, 0] = S[:, 1, 1] = S[:, 2, 2] = scale[:, 0]
    R = eye.clone()
    rotation = rotation.clamp(-math.pi / 2, math.pi)         # point_transforms.py:211
    yaw, pitch, roll = rotation[:, 0], rotation[:, 1], rotation[:, 2]
    yc, ys, pc, ps, rc, rs = yaw.cos(), yaw.sin(), pitch.cos(), pitch.sin(), roll.cos(), roll.sin()
    R[:, 0, 0] = yc * pc                                     # point_transforms.py:221-231
    R[:, 0, 1] = yc * ps * rs - ys * rc
    R[:, 0, 2] = yc * ps * rc + ys * rs
    R[:, 1, 0] = ys * pc
    R[:, 1, 1] = ys * ps * rs + yc * rc
    R[:, 1, 2] = ys * ps * rc - yc * rs
    R[:, 2, 0] = -ps
    R[:, 2, 1] = pc * rs
    R[:, 2, 2] = pc * rc
    T = eye.clone()
    T[:, 0, 3], T[:, 1, 3], T[:, 2, 3] = translation[:, 0], translation[:, 1], translation[:, 2]
    return S @ R @ T                                         # point_transforms.py:240


# ----------------------------------------------------------------------------------------------
# a2: rotation warp = affine image of the identity lattice
# ----------------------------------------------------------------------------------------------
def identity_grid_3d(d, s):
    """models/stage_1/volumetric_avatar/va.py:101-105 -> [1, d*s*s, 4] rows (u, v, w, 1)."""
    grid_s = torch.linspace(-1, 1, s)
    grid_z = torch.linspace(-1, 1, d)
    w, v, u = torch.meshgrid(grid_z, grid_s, grid_s, indexing="ij")
    e = torch.ones_like(u)
    return torch.stack([u, v, w, e], dim=3).view(1, -1, 4)


def rotation_warp(theta, d, s, inverse=False):
    """driver: notebooks/infer.py:583-586 (grid.bmm(theta[:, :3]^T)); source: infer.py:441-444 uses
    theta.float().inverse().  theta [B,4,4] -> [B,d,s,s,3].  (The reference is batch-1: F5.)"""
    if inverse:
        theta = theta.float().inverse().type(theta.type())
    grid = identity_grid_3d(d, s).repeat_interleave(theta.shape[0], dim=0)
    return grid.bmm(theta[:, :3].transpose(1, 2)).view(-1, d, s, s, 3)


# ----------------------------------------------------------------------------------------------
# a1: 3-D trilinear grid_sample
# ----------------------------------------------------------------------------------------------
def grid_sample(vol, grid, padding_mode="zeros"):
    """Model.grid_sample (va.py:264-265): F.grid_sample(inputs.float(), grid.float(), padding_mode=...);
    mode 'bilinear' (= trilinear on 5-D), align_corners=False (implicit default)."""
    return F.grid_sample(vol.float(), grid.float(), padding_mode=padding_mode, align_corners=False)


def _reflect(x, size):
    """ATen/native/GridSampler.h:89-106 reflect_coordinates with twice_low=-1, twice_high=2*size-1
    (align_corners=False), followed by clip_coordinates (:58-60)."""
    f32 = np.float32
    twice_low, twice_high = -1, 2 * size - 1
    mn = f32(twice_low) / f32(2)
    span = f32(twice_high - twice_low) / f32(2)
    x = np.abs(x - mn).astype(f32)
    extra = np.fmod(x, span).astype(f32)
    flips = np.floor(x / span).astype(f32)
    even = (np.fmod(flips, f32(2)) == 0)
    x = np.where(even, extra + mn, span - extra + mn).astype(f32)
    return np.minimum(f32(size - 1), np.maximum(x, f32(0))).astype(f32)


def grid_sample3d_restated(vol, grid, padding_mode="zeros"):
    """From-first-principles numpy fp32 restatement of ATen's grid_sampler_3d forward (CPU path,
    trilinear, align_corners=False) -- SURVEY.md section 8 row a1.  vol [N,C,D,H,W] float32, grid [N,Do,Ho,Wo,3].
    Per output voxel: unnormalise ((g+1)*size-1)/2 (GridSampler.h:27-36), optional border/reflection,
    floor, 8 corners, weights (x1-ix)(y1-iy)(z1-iz)..., bounds-checked gather, accumulate in the order
    tnw,tne,tsw,tse,bnw,bne,bsw,bse."""
    f32 = np.float32
    vol = np.ascontiguousarray(vol, dtype=f32)
    grid = np.ascontiguousarray(grid, dtype=f32)
    N, C, D, H, W = vol.shape
    _, Do, Ho, Wo, _ = grid.shape
    out = np.zeros((N, C, Do, Ho, Wo), dtype=f32)

    def unnorm(g, size):
        return (((g + f32(1)) * f32(size)) - f32(1)) / f32(2)

    for n in range(N):
        gx, gy, gz = grid[n, ..., 0], grid[n, ..., 1], grid[n, ..., 2]
        ix, iy, iz = unnorm(gx, W), unnorm(gy, H), unnorm(gz, D)
        if padding_mode == "border":
            ix = np.minimum(f32(W - 1), np.maximum(ix, f32(0)))
            iy = np.minimum(f32(H - 1), np.maximum(iy, f32(0)))
            iz = np.minimum(f32(D - 1), np.maximum(iz, f32(0)))
        elif padding_mode == "reflection":
            ix, iy, iz = _reflect(ix, W), _reflect(iy, H), _reflect(iz, D)
        x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
        x1, y1, z1 = x0 + f32(1), y0 + f32(1), z0 + f32(1)
        wx0, wx1 = (x1 - ix).astype(f32), (ix - x0).astype(f32)
        wy0, wy1 = (y1 - iy).astype(f32), (iy - y0).astype(f32)
        wz0, wz1 = (z1 - iz).astype(f32), (iz - z0).astype(f32)
        acc = np.zeros((C, Do, Ho, Wo), dtype=f32)
        # order tnw, tne, tsw, tse, bnw, bne, bsw, bse (t/b = z0/z1, n/s = y0/y1, w/e = x0/x1)
        for (zz, wz) in ((z0, wz0), (z1, wz1)):
            for (yy, wy) in ((y0, wy0), (y1, wy1)):
                for (xx, wx) in ((x0, wx0), (x1, wx1)):
                    wgt = ((wx * wy).astype(f32) * wz).astype(f32)
                    # NaN/inf coordinates compare false and are treated as out of bounds
                    with np.errstate(invalid="ignore"):
                        inb = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H) & (zz >= 0) & (zz < D)
                    xi = np.where(inb, xx, 0).astype(np.int64)
                    yi = np.where(inb, yy, 0).astype(np.int64)
                    zi = np.where(inb, zz, 0).astype(np.int64)
                    v = vol[n][:, zi, yi, xi]                 # [C,Do,Ho,Wo]
                    contrib = (v * wgt[None]).astype(f32)
                    acc = np.where(inb[None], (acc + contrib).astype(f32), acc)
        out[n] = acc
    return out


# ----------------------------------------------------------------------------------------------
# a4: predict_embed (models/stage_1/volumetric_avatar/va.py:813-885), target branch only
# ----------------------------------------------------------------------------------------------
def predict_warp_embed(sd, pose_embed, idt_embed, cfg):
    """warp_embed = Conv1x1_SN((Linear(pose_embed).view(B,512,4,4) + idt_embed) * 0.5) -> [B,512,16]
    (va.py:820-823 pose_unsqueeze_nw; :857 warp_embed_head_orig_nw; :861 view)."""
    es = cfg["gen_embed_size"]
    b = pose_embed.shape[0]
    e = F.linear(pose_embed, sd["pose_unsqueeze_nw.weight"]).view(b, -1, es, es)
    x = (e + idt_embed.repeat_interleave(b // idt_embed.shape[0], dim=0)) * 0.5
    w = sn_weight(sd, "warp_embed_head_orig_nw")
    o = F.conv2d(x, w)
    return o.view(b, o.shape[1], es * es)


# ----------------------------------------------------------------------------------------------
# a5: WarpGenerator (networks/volumetric_avatar/warp_generator_resnet.py:38-181)
# ----------------------------------------------------------------------------------------------
def warp_generator_channels(cfg):
    nb = int(math.log(cfg["warp_output_size"] // cfg["gen_embed_size"], 2))
    f = lambda i: min(int(cfg["gen_num_channels"] * cfg["warp_channel_mult"] * 2 ** i), cfg["gen_max_channels"]) // 32 * 32
    return [f(nb)] + [f(i) for i in range(nb - 1, -1, -1)]   # warp_generator_resnet.py:58-76


def warp_generator(sd, prefix, warp_embed, cfg):
    """returns (warp [B,d,s,s,3], deltas [B,3,d,s,s]).  warp_generator_resnet.py:125-181."""
    chans = warp_generator_channels(cfg)
    nb = len(chans) - 1
    inp = cfg["gen_embed_size"]                               # gen_dummy_input_size of this cfg (va_arguments.py:556)
    out_depth = cfg["gen_latent_texture_depth"]
    n_depth_resize = int(math.log(cfg["gen_latent_texture_size"] // inp, 2))
    params = projector_norm(sd, prefix + ".projector", 2 * nb, warp_embed)
    b = warp_embed.shape[0]
    x = warp_embed.view(b, -1, inp, inp)
    x = F.conv2d(x, sn_weight(sd, prefix + ".first_conv")).view(b, -1, inp, inp, inp)   # :138-141
    size = [inp, inp, inp]
    for i in range(1, nb + 1):
        size[1] *= 2
        size[2] *= 2
        depth_new = min(out_depth * 2 ** (n_depth_resize - i), size[1]) if i < n_depth_resize else out_depth
        up = depth_new > size[0]
        down = depth_new < size[0]
        size[0] = depth_new
        if up:
            x = F.interpolate(x, scale_factor=2, mode="trilinear")                # :163-164
        else:
            x = F.interpolate(x, scale_factor=(1, 2, 2), mode="trilinear")        # :165-166
        x = res_block(x, sd, f"{prefix}.blocks_3d.{i - 1}", "ws", ada=(params[2 * (i - 1)], params[2 * (i - 1) + 1]))
        if down:
            x = F.avg_pool3d(x, kernel_size=(2, 1, 1), stride=(2, 1, 1))          # :118, :170-171
    x = F.relu(group_norm(x.float(), sd, prefix + ".pre_head.0"))                 # :173-174
    w, bias = conv_params(sd, prefix + ".head.0.0", "sn")
    deltas = torch.tanh(F.conv3d(x, w, bias, padding=1))                          # :99-107,176
    warp = (sd[prefix + ".identity_grid"] + deltas).permute(0, 2, 3, 4, 1)        # :178
    return warp, deltas


# ----------------------------------------------------------------------------------------------
# a9: Decoder + ImageDecoder (networks/volumetric_avatar/decoder.py:52-150,152-238,241-358,398-410)
# ----------------------------------------------------------------------------------------------
def decoder_channels(cfg):
    nup = int(math.log(cfg["image_size"] // cfg["gen_latent_texture_size"], 2))
    trunk = min(int(cfg["gen_num_channels"] * cfg["dec_channel_mult"] * 2 ** nup), cfg["dec_max_channels"])  # decoder.py:59
    stages = []
    c = trunk
    for _ in range(nup):
        c = max(int(c / cfg["im_dec_ch_div_factor"] / 32) * 32, cfg["gen_num_channels"])                   # decoder.py:277
        stages.append(c)
    return trunk, stages


def decoder(sd, prefix, feat_2d, cfg):
    """returns (img, feat_2d_after_res_decoder, img_feat) as Decoder.forward(..., stage_two=True)
    (decoder.py:152-238; the adaptive-norm projector list is empty for the released flags, F8)."""
    _, stages = decoder_channels(cfg)
    x = F.conv2d(feat_2d, sn_weight(sd, prefix + ".res_decoder.0"))               # decoder.py:76-82
    for i in range(cfg["dec_num_blocks"]):
        x = res_block(x, sd, f"{prefix}.res_decoder.{i + 1}", "ws")               # decoder.py:84-91
    feat = x
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")              # utils.py:684-685
    k = 0
    for _ in stages:
        for j in range(cfg["im_dec_num_lrs_per_resolution"]):
            x = res_block(x, sd, f"{prefix}.img_decoder.dec_img_blocks.{k}", "ws", upsample=up if j == 0 else None)
            k += 1
    img_feat = x
    h = F.relu(group_norm(x.float(), sd, prefix + ".img_decoder.dec_img_head.0"))
    w, bias = conv_params(sd, prefix + ".img_decoder.dec_img_head.2", "ws")
    img = torch.sigmoid(F.conv2d(h, w, bias))                                     # decoder.py:347-358
    return img, feat, img_feat


# ----------------------------------------------------------------------------------------------
# a6: LocalEncoder (networks/volumetric_avatar/local_encoder.py:48-125)
# ----------------------------------------------------------------------------------------------
def local_encoder(sd, prefix, img, cfg, image_size=None, latent_size=None, ws=True):
    """also networks/volumetric_avatar/local_encoder_old.py:25-117 (stage 2: identical structure, other sizes).
    ws: whether the WS replacement hit the convs that follow a GroupNorm (use_ws and norm 'gn')."""
    s = img.shape[2]
    image_size = cfg["image_size"] if image_size is None else image_size
    latent_size = cfg["latent_volume_size"] if latent_size is None else latent_size
    kind = "ws" if ws else "sn"
    nblk = int(math.log(image_size // latent_size, 2))
    w, b = conv_params(sd, f"{prefix}.from_rgb_{s}px", "sn")
    x = F.conv2d(img, w, b, padding=3)                                            # local_encoder.py:64-73
    pool = lambda t: F.avg_pool2d(t, 2)                                           # AvgPool2d(stride) utils.py:962
    for i in range(nblk):
        x = res_block(x, sd, f"{prefix}.enc_{i}_block={s}px", kind, downsample=pool)
        s //= 2
    x = F.relu(group_norm(x, sd, prefix + ".finale_layers.0"))
    w, b = conv_params(sd, prefix + ".finale_layers.2", kind)
    return F.conv2d(x, w, b)                                                      # local_encoder.py:97-111


# ----------------------------------------------------------------------------------------------
# a7: VPN_ResBlocks (vpn_resblocks.py:34-49 -> resblocks_3d.py:10-62); plain GroupNorm => no WS (SN both convs)
# ----------------------------------------------------------------------------------------------
def vpn_resblocks(sd, prefix, vol, cfg):
    for i in range(cfg["source_volume_num_blocks"]):
        vol = res_block(vol, sd, f"{prefix}.net.net.{i}", "sn")
    return vol


# ----------------------------------------------------------------------------------------------
# a8: Unet3D (networks/volumetric_avatar/unet_3d.py:44-194 init, :196-290 forward)
# ----------------------------------------------------------------------------------------------
def unet3d(sd, prefix, vol, cfg):
    nb = int(math.log(cfg["gen_latent_texture_size"] // cfg["gen_dummy_input_size"], 2))
    init_depth = out_depth = cfg["gen_latent_texture_depth"]
    s = vol.shape[-1]
    x = vol
    feats = []
    size = [init_depth, s, s]
    for i in range(nb):                                                           # unet_3d.py:206-233
        up = down = False
        if i < nb - 1:
            size[1] //= 2
            size[2] //= 2
            depth_new = min(size[0] * 2, size[1])
            up, down = depth_new > size[0], depth_new < size[0]
            size[0] = depth_new
            if up:
                x = F.interpolate(x, scale_factor=(2, 1, 1), mode="trilinear")
        x = res_block(x, sd, f"{prefix}.blocks_3d_down.{i}", "sn")
        feats.append(x)
        if i < nb - 1:
            x = F.avg_pool3d(x, 2, 2) if down else F.avg_pool3d(x, (1, 2, 2), (1, 2, 2))
    feats = feats[::-1]
    x = sd[prefix + ".input_tensor"].repeat_interleave(feats[0].shape[0], dim=0)  # unet_3d.py:254
    size = [x.shape[2], x.shape[3], x.shape[4]]
    for i, feat in enumerate(feats, 1):                                           # unet_3d.py:258-286
        size[1] *= 2
        size[2] *= 2
        depth_new = min(out_depth * 2 ** (nb - i), size[1])
        up, down = depth_new > size[0], depth_new < size[0]
        size[0] = depth_new
        if up:
            x = F.interpolate(x, scale_factor=2, mode="trilinear")
        else:
            x = F.interpolate(x, scale_factor=(1, 2, 2), mode="trilinear")
        skip = res_block(feat, sd, f"{prefix}.skip_blocks_3d_up.{i - 1}", "sn")
        x = res_block(x + skip, sd, f"{prefix}.blocks_3d_up.{i - 1}", "sn")
        if down:
            x = F.avg_pool3d(x, (2, 1, 1), (2, 1, 1))
    x = F.relu(group_norm(x, sd, prefix + ".head.0"))
    w, b = conv_params(sd, prefix + ".head.2", "sn")
    return F.conv3d(x, w, b)                                                      # unet_3d.py:146-153,288


# ----------------------------------------------------------------------------------------------
# whole passes, as notebooks/infer.py replays them
# ----------------------------------------------------------------------------------------------
def source_pass(sd, cfg, source_img_masked, idt_embed, source_pose_embed, theta_src):
    """notebooks/infer.py:433-507 with the third-party embedders replaced by their outputs."""
    c, d, s = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
    pm = cfg["grid_sample_padding_mode"]
    latents = local_encoder(sd, "local_encoder_nw", source_img_masked, cfg)       # infer.py:433
    rot = rotation_warp(theta_src, d, s, inverse=True)                            # infer.py:441-444
    emb = predict_warp_embed(sd, source_pose_embed, idt_embed, cfg)               # infer.py:459
    xy_warp, _ = warp_generator(sd, "xy_generator_nw", emb, cfg)                  # infer.py:462
    vol = vpn_resblocks(sd, "volume_source_nw", latents.view(1, c, d, s, s), cfg)  # infer.py:485-491
    pre = grid_sample(grid_sample(vol, rot, pm), xy_warp, pm)                     # infer.py:499-500
    canonical = unet3d(sd, "volume_process_nw", pre, cfg)                         # infer.py:507
    return dict(latents=latents, source_rotation_warp=rot, xy_warp=xy_warp, source_volume=vol,
                pre_canonical=pre, canonical=canonical, warp_embed=emb)


def driver_pass(sd, cfg, canonical, idt_embed, target_pose_embed, theta_drv):
    """notebooks/infer.py:583-637, batched over B driver frames (the reference loops batch-1 calls;
    every op below is per-sample independent, so batching does not change per-frame results)."""
    c, d, s = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
    pm = cfg["grid_sample_padding_mode"]
    b = target_pose_embed.shape[0]
    rot = rotation_warp(theta_drv, d, s)                                          # infer.py:583-586
    emb = predict_warp_embed(sd, target_pose_embed, idt_embed, cfg)               # infer.py:609
    uv_warp, delta = warp_generator(sd, "uv_generator_nw", emb, cfg)              # infer.py:612
    vol = canonical.expand(b, -1, -1, -1, -1) if canonical.shape[0] == 1 else canonical
    aligned = grid_sample(grid_sample(vol, uv_warp, pm), rot, pm)                 # infer.py:618-619
    feat = aligned.reshape(b, c * d, s, s)                                        # infer.py:627
    img, deep_f, img_f = decoder(sd, "decoder_nw", feat, cfg)                     # infer.py:637
    return dict(target_rotation_warp=rot, uv_warp=uv_warp, delta_uv=delta, aligned=aligned,
                img=img, deep_f=deep_f, img_f=img_f, warp_embed=emb)


# ----------------------------------------------------------------------------------------------
# stage 2: notebooks/infer_s2.py:351-376 on models/stage_2/base/volumetric_avatar_two.py:338-445
# ----------------------------------------------------------------------------------------------
STAGE2_DEFAULT_CFG = dict(   # argparse defaults of volumetric_avatar_two.py:26-270 (released stage-2 args are not in the repo)
    output_size_s2=512, gen_latent_texture_size2=64, gen_latent_texture_channels2=64, gen_latent_texture_depth=16,
    gen_num_channels=32, gen_max_channels=512, enc_channel_mult_stage2=4.0, dec_channel_mult_stage2=4.0,
    dec_num_blocks_stage2=8, dec_max_channels2=512, norm_layer_type="bn", use_ws=False,
)


def stage2_cfg_from_args(args):
    return {k: getattr(args, k) for k in STAGE2_DEFAULT_CFG}


def stage2_decoder_channels(cfg2):
    nup = int(math.log(cfg2["output_size_s2"] // cfg2["gen_latent_texture_size2"], 2))
    trunk = min(int(cfg2["gen_num_channels"] * cfg2["dec_channel_mult_stage2"] * 2 ** nup), cfg2["dec_max_channels2"])
    ups, c = [], trunk
    for _ in range(nup - 1):
        c = max(c // 2, cfg2["gen_num_channels"])                                  # decoder_s2_old.py:378
        ups.append(c)
    return trunk, ups


def decoder_stage2(sd, prefix, feat_2d, cfg2):
    """Decoder_stage2 + ImageDecoder_stage2 (networks/volumetric_avatar/decoder_s2_old.py:18-218, :346-472):
    1x1 -> dec_num_blocks ResBlocks -> (num_up-1) nearest-x2 ResBlocks -> x2 ResBlock to 128 + three ResBlocks
    128/64/32 -> norm + ReLU + 1x1 + tanh.  Returns the additive residual image."""
    kind = "ws" if (cfg2["use_ws"] and cfg2["norm_layer_type"] == "gn") else "sn"
    _, ups = stage2_decoder_channels(cfg2)
    x = F.conv2d(feat_2d, sn_weight(sd, prefix + ".res_decoder.0"))
    for i in range(cfg2["dec_num_blocks_stage2"]):
        x = res_block(x, sd, f"{prefix}.res_decoder.{i + 1}", kind)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    for i in range(len(ups)):
        x = res_block(x, sd, f"{prefix}.img_decoder.dec_img_blocks.{i}", kind, upsample=up)
    x = res_block(x, sd, f"{prefix}.img_decoder.dec_img_feat_blocks.0", kind, upsample=up)
    for i in (1, 2, 3):
        x = res_block(x, sd, f"{prefix}.img_decoder.dec_img_feat_blocks.{i}", kind)
    h = F.relu(group_norm(x.float(), sd, prefix + ".img_decoder.dec_img_head.0"))
    w, b = conv_params(sd, prefix + ".img_decoder.dec_img_head.2", kind)
    return torch.tanh(F.conv2d(h, w, b))


def stage2_forward(sd, cfg2, img, mask, face_mask):
    """infer_s2.py:351-376 with the third-party masks (MODNet matte `mask`, BiSeNet `face_mask`) given and `img`
    already at output_size_s2: add = decoder(encoder(img*mask)) * (mask*face_mask); out = clamp(img + add, 0, 1)"""
    ws = cfg2["use_ws"] and cfg2["norm_layer_type"] == "gn"
    lat = local_encoder(sd, "local_encoder", img * mask, cfg2, image_size=cfg2["output_size_s2"],
                        latent_size=cfg2["gen_latent_texture_size2"], ws=ws)
    add = decoder_stage2(sd, "decoder", lat, cfg2)
    out = (img + add * (mask * face_mask)).clamp(max=1, min=0)
    return dict(latents=lat, add=add, out=out)


# ----------------------------------------------------------------------------------------------
# f1: embedders.  ResNet body: torchvision 0.9.1 `models.resnet*` -- a third-party
# dependency absent here, restated from the published architecture (see oracle/tv_resnet.py for the parity status);
# everything around it follows the reference files cited per function.
# ----------------------------------------------------------------------------------------------
RESNET_LAYERS = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet34": ("basic", (3, 4, 6, 3)),
                 "resnet50": ("bottleneck", (3, 4, 6, 3))}
IMAGENET_MEAN = (0.485, 0.456, 0.406)     # identity_embedder.py:56-57, expression_embedder.py:421-422
IMAGENET_STD = (0.229, 0.224, 0.225)


def _resnet_conv(x, sd, prefix, stride, padding):
    """conv of a (possibly wrapped) torchvision ResNet: spectral norm leaves `weight_orig/_u/_v`
    (utils/spectral_norm.py:96-168), the WS replacement leaves `weight` + a new `bias` (utils.py:1080-1083), an
    untouched torchvision conv has `weight` only (head_pose_regressor.py:14)."""
    if (prefix + ".weight_orig") in sd:
        w, b = sn_weight(sd, prefix), None
    elif (prefix + ".bias") in sd:
        w, b = ws_weight(sd[prefix + ".weight"]), sd[prefix + ".bias"]
    else:
        w, b = sd[prefix + ".weight"], None
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def resnet_trunk(sd, prefix, x, arch):
    """conv1 .. layer4 of torchvision's ResNet.forward (what IdtEmbed._forward_impl identity_embedder.py:59-69 and
    ResNetWrapper._forward_impl expression_embedder.py:424-439 call); norms are GroupNorm(32) after replace_bn_to_gn
    (utils.py:1020-1038) or eval BatchNorm, told apart by the running statistics in the state_dict."""
    kind, counts = RESNET_LAYERS[arch]
    norm = lambda t, p: group_norm(t, sd, p)
    x = F.relu(norm(_resnet_conv(x, sd, prefix + ".conv1", 2, 3), prefix + ".bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, nblocks in enumerate(counts):
        for bi in range(nblocks):
            p = f"{prefix}.layer{li + 1}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = x
            if kind == "basic":
                out = F.relu(norm(_resnet_conv(x, sd, p + ".conv1", stride, 1), p + ".bn1"))
                out = norm(_resnet_conv(out, sd, p + ".conv2", 1, 1), p + ".bn2")
            else:
                out = F.relu(norm(_resnet_conv(x, sd, p + ".conv1", 1, 0), p + ".bn1"))
                out = F.relu(norm(_resnet_conv(out, sd, p + ".conv2", stride, 1), p + ".bn2"))
                out = norm(_resnet_conv(out, sd, p + ".conv3", 1, 0), p + ".bn3")
            if any(k.startswith(p + ".downsample.0.") for k in sd):
                identity = norm(_resnet_conv(x, sd, p + ".downsample.0", stride, 0), p + ".downsample.1")
            x = F.relu(out + identity)
    return x


def _imagenet_normalise(x):
    mean = torch.tensor(IMAGENET_MEAN)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD)[None, :, None, None]
    return (x - mean) / std


def idt_embed(sd, prefix, masked_source, arch="resnet50", idt_image_size=256, idt_output_size=4):
    """IdtEmbed.forward_image (identity_embedder.py:76-87): bilinear resize, ImageNet normalisation, trunk, the 1x1
    `fc` conv BEFORE the adaptive average pool (:66-67)."""
    x = F.interpolate(masked_source, size=(idt_image_size, idt_image_size), mode="bilinear")
    x = resnet_trunk(sd, prefix + ".net", _imagenet_normalise(x), arch)
    x = _resnet_conv(x, sd, prefix + ".net.fc", 1, 0)
    return F.adaptive_avg_pool2d(x, idt_output_size)


def head_pose(sd, crop):
    """HeadPoseRegressor.forward (head_pose_regressor.py:21-32): bilinear resize to 128, torchvision resnet18 with a 9-way
    fc, split into scale / rotation / translation, get_transform_matrix."""
    if crop.shape[2] != 128 or crop.shape[3] != 128:
        crop = F.interpolate(crop, size=(128, 128), mode="bilinear")
    x = resnet_trunk({"net." + k: v for k, v in sd.items()}, "net", crop, "resnet18")
    x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    out = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    scale, rotation, translation = out.split([3, 3, 3], dim=1)
    return dict(theta=get_transform_matrix(scale, rotation, translation), scale=scale, rotation=rotation,
                translation=translation)


def expression_align_theta(theta):
    """expression_embedder.py:178-200 (use_smart_scale=False): 4x4 inverse, keep rows/cols (0,1,3) -> 2-D affine, then the
    2x zoom-in `@ diag(0.5, 0.5, 1)`, first two rows."""
    eye = torch.zeros(theta.shape[0], 1, 4)
    eye[:, :, 3] = 1
    t4 = torch.cat([theta[:, :3, :], eye], dim=1)
    inv2d = t4.float().inverse()[:, :, [0, 1, 3]][:, [0, 1, 3]]
    scale = torch.zeros_like(inv2d)
    scale[:, [0, 1], [0, 1]] = 0.5
    scale[:, 2, 2] = 1
    return torch.bmm(inv2d, scale)[:, :2]


def expression_embed(sd, prefix, crop, theta, arch="resnet18", exp_image_size=256, lpe_output_size=4):
    """ExpressionEmbed.forward in the inference form notebooks/infer.py:452,601 calls it (estimate_kp_by_net=True,
    use_seg=False, eval): align the crop with the inverse head pose (expression_embedder.py:176-222), then
    ResNetWrapper.forward (:441-459) = ImageNet normalisation, trunk, 1x1 `fc` conv, (eval dropout), adaptive average
    pool, flatten, spectral-norm Linear.  The reference runs it on cat(source, target) = the same crop twice; GroupNorm is
    per sample, so one copy is computed."""
    gs = exp_image_size // 2
    lin = torch.linspace(-1, 1, gs)
    v, u = torch.meshgrid(lin, lin, indexing="ij")
    ident = torch.stack([u, v, torch.ones_like(u)], dim=2).view(1, -1, 3)
    a = expression_align_theta(theta)
    warp = ident.repeat_interleave(crop.shape[0], dim=0).bmm(a.transpose(1, 2)).view(crop.shape[0], gs, gs, 2)
    aligned = F.grid_sample(crop.float(), warp.float(), align_corners=False)
    p = prefix + ".net_face"
    x = resnet_trunk(sd, p + ".net", _imagenet_normalise(aligned), arch)
    x = _resnet_conv(x, sd, p + ".net.fc", 1, 0)
    x = torch.flatten(F.adaptive_avg_pool2d(x, lpe_output_size), 1)
    return dict(pose_embed=F.linear(x, sn_weight(sd, p + ".pose_head")), img_align=aligned, align_warp=warp)
