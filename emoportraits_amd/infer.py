"""Drop-in `InferenceWrapper` for the MI355X hot path -- SURVEY.md section 8(b), seams b1 (user API) and b2 (model attributes).

Mirrors notebooks/infer.py of the reference: same constructor and `forward` signature (infer.py:63-65, :355-357),
same files (`<project_dir>/<folder>/<experiment_name>/args.txt`, `.../checkpoints/<model_file_name>`), same cached
attributes after a source call (`idt_embed`, `source_latent_volume`, `target_latent_volume`, `pred_source_theta`, ...)
and the same return value `(List[PIL.Image], Tensor[B,3,S,S])` / `None`.

The embedders that feed the hot path -- IdtEmbed (ResNet-50), ExpressionEmbed and HeadPoseRegressor (ResNet-18s), SURVEY.md
section 8f-1 -- run on the HIP kernels too (emoportraits_amd/embedders.py): they are built from the `idt_embedder_nw.*` /
`expression_embedder_nw.*` keys of the checkpoint and from `head_pose_regressor_path` (args.txt or constructor argument,
va_arguments.py:26) whenever those are present.

What is NOT here, by scope: the third-party nets outside the checkpoint -- face detector / cropper (mediapipe), face
parsing (BiSeNet), matting (MODNet).  They plug in through `embedders=` (any callables; an entry there also overrides a
native embedder); without them `forward` needs `crop=False` and `source_mask=`.  Embeddings can also be supplied directly
through the reference's own hooks `custome_target_pose_embed` / `custome_target_theta_embed` (infer.py:565-566,603-604)
and their source-side counterparts added here (`custome_source_pose_embed`, `custome_source_theta_embed`,
`custome_idt_embed`).  Missing pieces raise -- nothing falls back silently.

Extension over the reference (which is batch-1, F5): driver inputs may carry a batch dimension, and
`animate()` streams N driver frames in device-sized batches, sharded across ranks (emoportraits_amd/parallel.py).
"""
import os
import pathlib
from argparse import Namespace

import torch

from . import config as cfg_mod
from . import embedders as emb_mod
from . import graphs, hostglue, nets, ops, parallel, schema


# smooth_pose in animate_frames: the crops of a rank's shard stay resident between the head-pose pass and the render pass up to
# this many bytes (3 MB per 512^2 frame; beyond it they are cropped a second time)
_SMOOTH_KEEP_BYTES = int(float(os.environ.get("EMO_SMOOTH_KEEP_GB", "16")) * (1 << 30))


class HipModel:
    """The `self.model` attribute seam (b2): reference attribute names and call signatures over the HIP executors."""

    def __init__(self, hot_path, args):
        self.hp = hot_path
        self.args = args
        c = hot_path.cfg
        d, s = c["latent_volume_depth"], c["latent_volume_size"]
        gs, gz = torch.linspace(-1, 1, s), torch.linspace(-1, 1, d)
        w, v, u = torch.meshgrid(gz, gs, gs, indexing="ij")
        # models/stage_1/volumetric_avatar/va.py:101-105
        self.identity_grid_3d = torch.stack([u, v, w, torch.ones_like(u)], dim=3).view(1, -1, 4).to(hot_path.device)
        self._ident3 = torch.stack([u, v, w], 0)[None].contiguous().to(hot_path.device)   # [1,3,d,s,s]
        self.embed_size = c["gen_embed_size"]
        self.resize_warp_func = lambda x: x        # warp_output_size == gen_latent_texture_size is enforced by config

    # va.py:264-265
    def grid_sample(self, inputs, grid):
        return ops.grid_sample3d(inputs.float().contiguous(), grid.float().contiguous(),
                                 padding_mode=self.hp.pad)

    def local_encoder_nw(self, img):
        return self.hp.local_encoder(img.float().contiguous())

    def volume_source_nw(self, vol):
        return self.hp.volume_source(vol.contiguous())

    def volume_process_nw(self, vol, embed_dict=None):
        return self.hp.volume_process(vol.contiguous())

    def _warp_from_delta(self, delta):
        # warp_generator_resnet.py:178: (identity_grid + deltas).permute(0, 2, 3, 4, 1) -- a permuted view, as there
        return ops.add(delta, self._ident3.reshape(-1)).permute(0, 2, 3, 4, 1)

    def xy_generator_nw(self, embed_dict):
        delta = self.hp.xy_generator(embed_dict["orig"].contiguous())
        return [self._warp_from_delta(delta), delta]

    def uv_generator_nw(self, embed_dict):
        delta = self.hp.uv_generator(embed_dict["orig"].contiguous())
        return [self._warp_from_delta(delta), delta]

    def decoder_nw(self, data_dict, embed_dict, feat_2d, input_flip_feat=False, stage_two=False, **_):
        img, feat, img_f = self.hp.decoder(feat_2d.contiguous())
        if stage_two:
            return img, None, feat, img_f
        return img, None, None, None

    def predict_embed(self, data_dict):
        """va.py:813-885 -> (source, target, mixing, embed_dict); the mixing branch is a training construct (None)."""
        idt = data_dict["idt_embed"]
        out = []
        for key in ("source_pose_embed", "target_pose_embed"):
            e = self.hp.embed(data_dict[key].float().contiguous(), idt.float().contiguous())
            out.append({"orig": e, "orig_d": e, "ada_v": data_dict[key]})
        return out[0], out[1], None, {}


class InferenceWrapper:
    def __init__(self, experiment_name, which_epoch='latest', model_file_name='', use_gpu=True, num_gpus=1,
                 fixed_bounding_box=False, project_dir='./', folder='mp_logs', model_='va',
                 torch_home='', debug=False, print_model=False, print_params=True, args_overwrite={}, state_dict=None,
                 pose_momentum=0.5, rank=0, args_path=None, embedders=None, head_pose_regressor_path=None,
                 use_graphs=True, precision=None):
        if not use_gpu:
            raise RuntimeError("emoportraits_amd runs on MI355X only: use_gpu=False is not supported (no CPU path)")
        if model_ != 'va':
            raise ValueError("only the stage-1 'va' model is on the MI355X hot path")
        self.use_gpu, self.debug, self.num_gpus = use_gpu, debug, num_gpus
        args_path = pathlib.Path(project_dir) / folder / experiment_name / 'args.txt' if args_path is None else args_path
        found = cfg_mod.parse_args_txt(args_path)                                  # infer.py:74-76
        found['project_dir'] = project_dir
        for k, v in (args_overwrite or {}).items():                                # infer.py:79-81
            found[k] = v
        self.args = Namespace(**found)
        self.cfg = cfg_mod.hot_path_config(found, released=False) if 'norm_layer_type' in found else \
            cfg_mod.hot_path_config(found)
        for k, v in self.cfg.items():
            setattr(self.args, k, v)
        if torch_home:
            os.environ['TORCH_HOME'] = torch_home
        # one process per GPU (infer.py:94-105 uses the same env:// rendezvous)
        if num_gpus > 8:
            raise RuntimeError("at most 8 GPUs per node")                          # infer.py:104-105 (bare raise there)
        if num_gpus > 1:
            self.rank, self.world = parallel.init_distributed()
        else:
            self.rank, self.world = 0, 1
        self.device = torch.device("cuda", parallel.local_device_index())
        torch.cuda.set_device(self.device)

        self.model_checkpoint = pathlib.Path(project_dir) / folder / experiment_name / 'checkpoints' / model_file_name
        self.model_dict = torch.load(self.model_checkpoint, map_location='cpu') if state_dict is None else state_dict
        schema.check_state_dict(self.model_dict, self.cfg)        # strict: the reference's strict=False hides mismatches
        self.hot_path = nets.HotPath(self.model_dict, self.cfg, self.device, precision=precision)   # 'f16': opt-in, see nets.HotPath
        self.model = HipModel(self.hot_path, self.args)
        if rank == 0 and print_params:
            n = sum(v.numel() for k, v in self.model_dict.items() if k.startswith(schema.HOT_PATH_PREFIXES))
            print(f'Number of hot-path parameters: {n}')
        native = self._native_embedders(found, head_pose_regressor_path)
        self.embedders = {**native, **dict(embedders or {})}
        # hipGraph replay of the per-frame sequences (emoportraits_amd/graphs.py); only this repo's own executors are
        # captured, user-supplied embedder callables always run eagerly
        # On by default: a drop-in user calls forward() one frame at a time, which is launch-bound without the replay
        # (bench.py extras: emotion_driver_forward_fps, latency_b1_ms).  The first call of an input signature runs eagerly,
        # the second captures, later ones replay (use_graphs='eager_first' semantics; use_graphs=False never captures).
        self.use_graphs = bool(use_graphs)
        self._graphed = {}
        if self.use_graphs:
            first = 0 if use_graphs == 'capture_first' else 1
            self._graphed['driver'] = graphs.Graphed(
                lambda pose, theta: self.hot_path.driver_pass(self._canonical_cl, self.idt_embed, pose, theta), eager_calls=first)
            hp_net, ex_net = native.get('head_pose_regressor'), native.get('expression_embedder')
            if hp_net is not None and self.embedders['head_pose_regressor'] is hp_net:
                self._graphed['head_pose_regressor'] = graphs.Graphed(lambda crop: hp_net.forward(crop, True), eager_calls=first)
            if ex_net is not None and self.embedders['expression_embedder'] is ex_net:
                self._graphed['expression_embedder'] = graphs.Graphed(lambda crop, theta: ex_net(crop, theta, True)[:2],
                                                                      eager_calls=first)

        self.fixed_bounding_box = fixed_bounding_box
        self.momentum = 0.01
        self.center = None
        self.size = None
        self.pose_momentum = pose_momentum
        self.theta = None
        self.norm_momentum = 0.1
        self.delta_yaw = None
        self.delta_pitch = None
        self.resize_warp = False
        self.use_seg = bool(found.get('use_seg', False))
        self.target_latent_volume = None
        self._canonical_cl = None
        self._crop_tracker = None

    # ------------------------------------------------------------------------------------------------------
    def _native_embedders(self, found, head_pose_regressor_path):
        """IdtEmbed / ExpressionEmbed from the checkpoint (va.py:161,165), HeadPoseRegressor from its own file (va.py:258)"""
        out = {}
        sd = self.model_dict
        has = lambda p: any(k.startswith(p) for k in sd)
        if has("idt_embedder_nw.") or has("expression_embedder_nw."):
            ecfg = emb_mod.embedder_config(found, released='norm_layer_type' not in found)
            self.embedder_cfg = ecfg
            if has("idt_embedder_nw."):
                out['idt_embedder'] = emb_mod.IdtEmbed(sd, ecfg, self.device)
            if has("expression_embedder_nw."):
                out['expression_embedder'] = emb_mod.ExpressionEmbed(sd, ecfg, self.device)
        path = head_pose_regressor_path or found.get('head_pose_regressor_path')
        if head_pose_regressor_path is not None or (path and os.path.isfile(str(path))):
            out['head_pose_regressor'] = emb_mod.HeadPoseRegressor(torch.load(path, map_location='cpu'), self.device)
        return out

    def _set_source_cache(self, canonical=None, idt_embed=None):
        """per-identity cache; with graphs the captured sequences hold the buffer addresses, so a new identity is copied
        INTO the existing buffers instead of rebinding them"""
        if idt_embed is not None:
            if self.use_graphs and getattr(self, 'idt_embed', None) is not None and self.idt_embed.shape == idt_embed.shape:
                self.idt_embed.copy_(idt_embed)
            else:
                self.idt_embed = idt_embed.clone() if self.use_graphs else idt_embed
        if canonical is not None:
            self.target_latent_volume = canonical
            cl = self.hot_path.prepare_canonical(canonical)
            if self.use_graphs and self._canonical_cl is not None and self._canonical_cl.shape == cl.shape:
                self._canonical_cl.copy_(cl)
            else:
                self._canonical_cl = cl

    def _head_pose(self, crop):
        g = self._graphed.get('head_pose_regressor')
        if g is not None:
            return g(crop)
        return self._need('head_pose_regressor', 'a driver call')(crop, True)

    def _expression(self, crop, theta, what):
        """-> (pose_embed, aligned crop or None).  The aligned 128^2 crop is what the reference exposes as
        `target_img_align` (expression_embedder.py:233, infer.py:608); a user-supplied callable may return either the
        embedding alone or a tuple (embedding, aligned, ...)."""
        g = self._graphed.get('expression_embedder')
        if g is not None:
            out = g(crop, theta.float().contiguous())
        else:
            fn = self._need('expression_embedder', what)
            out = fn(crop, theta, True) if isinstance(fn, emb_mod.ExpressionEmbed) else fn(crop, theta)
        if isinstance(out, (tuple, list)):
            return out[0], (out[1] if len(out) > 1 else None)
        return out, None

    def _drive(self, pose, theta):
        g = self._graphed.get('driver')
        if g is not None:
            return g(pose, theta)
        return self.hot_path.driver_pass(self._canonical_cl, self.idt_embed, pose, theta)

    def _need(self, name, what):
        fn = self.embedders.get(name)
        if fn is None:
            raise RuntimeError(
                f"{what} needs the '{name}' network: it was not passed via InferenceWrapper(embedders={{'{name}': callable}}) "
                f"and, for the embedders this package runs itself, its weights were not found (checkpoint keys / "
                f"head_pose_regressor_path); the custome_* arguments can supply its output instead")
        return fn

    def convert_to_tensor(self, image):
        """infer.py:211-223: PIL / ndarray / tensor -> float tensor [B,3,H,W] in [0,1]"""
        import numpy as np
        if isinstance(image, torch.Tensor):
            t = image.float()
            return t[None] if t.dim() == 3 else t
        if isinstance(image, (list, tuple)):
            return torch.cat([self.convert_to_tensor(i) for i in image])
        arr = np.asarray(image)
        t = torch.from_numpy(arr.copy())
        if t.dtype == torch.uint8:
            t = t.float() / 255.0
        t = t.float()
        if t.dim() == 2:
            t = t[..., None].expand(-1, -1, 3)
        return t.permute(2, 0, 1)[None]

    def _prepare_image(self, image):
        S = self.cfg["image_size"]
        t = self.convert_to_tensor(image)[:, :3].to(self.device).contiguous()
        if t.shape[-2:] != (S, S):
            t = ops.resize2d(t, (S, S), "bicubic")                                  # infer.py:399-401
        return t

    def crop_image(self, image, faces, use_smoothed_crop=False, scale=1):
        """notebooks/infer.py:301-352: square window around each face box (host arithmetic, emoportraits_amd/hostglue.py),
        read in place from the frame and resized to image_size with the bicubic kernel, clipped to [0,1].
        image: list of [3,H,W] tensors or a [B,3,H,W] tensor; faces: list of (x0, y0, x1, y1) or None.
        Returns (crops [B,3,S,S] on the device, face_check, face_scale_stats) like the reference."""
        import numpy as np
        S = self.cfg["image_size"]
        if use_smoothed_crop and self._crop_tracker is None:
            self._crop_tracker = hostglue.CropTracker(self.momentum, self.fixed_bounding_box)
        crops, face_check, face_scale_stats = [], np.ones(len(image), dtype=bool), []
        for b, face in enumerate(faces):
            frame = image[b]
            win = hostglue.crop_window(face, frame.shape[2], frame.shape[1],
                                       self._crop_tracker if use_smoothed_crop else None, scale)
            if win is None:
                face_check[b] = False
                crops.append(torch.zeros((1, 3, S, S), device=self.device))
                face_scale_stats.append(0)
                continue
            x_lo, y_lo, side, face_scale = win
            frame = frame[None, :3].to(self.device).float().contiguous()
            crops.append(ops.resize2d(frame, (S, S), "bicubic", window=(x_lo, y_lo, side, side), clamp01=True))
            face_scale_stats.append(face_scale)
        if self._crop_tracker is not None:
            self.center, self.size = self._crop_tracker.center, self._crop_tracker.size
        return torch.cat(crops), face_check, face_scale_stats

    def _detect_and_crop(self, images):
        """crop=True (notebooks/infer.py:376-393, :515-546): face detector (third party: mediapipe in the reference, here the
        'face_detector' callable: PIL image -> relative box (xmin, ymin, width, height) or None) + crop_image.  A 'cropper'
        callable, if given, replaces the whole step."""
        if 'cropper' in self.embedders:
            return self.embedders['cropper'](images).to(self.device)
        det = self._need('face_detector', 'crop=True')
        images = images if isinstance(images, (list, tuple)) else [images]
        faces, tensors = [], []
        for img in images:
            rel = det(img)
            t = self.convert_to_tensor(img)[0, :3]
            faces.append(None if rel is None else hostglue.detection_to_face(*rel, t.shape[2], t.shape[1]))
            tensors.append(t)
        crops, self.face_check, self.face_scale_stats = self.crop_image(tensors, faces)
        return crops

    def get_mixing_theta(self, source_theta, target_theta):
        """notebooks/infer.py:686-736 (host scipy polar decomposition there as well)"""
        mixed = hostglue.mixing_theta(source_theta.detach().cpu().numpy(), target_theta.detach().cpu().numpy(), self.mix_old)
        return torch.from_numpy(mixed).float().to(self.device)

    def _theta_from(self, embed):
        """(scale, rotation, translation) as the reference's custome_target_theta_embed (-> get_transform_matrix,
        infer.py:565-566), or an already formed [B,4,4] theta tensor (extension)"""
        if isinstance(embed, torch.Tensor):
            return embed.to(self.device).float().contiguous(), None
        srt = tuple(t.to(self.device).float().contiguous() for t in embed)
        return ops.pose_theta(*srt), srt

    def _smooth_thetas(self, thetas):
        """notebooks/infer.py:571-581 for a batch of thetas IN FRAME ORDER: the EMA runs once on the host over the 16 floats
        per frame (hostglue.ema_scan: bit-identical to the reference's per-frame loop of device ops), `self.theta` carries the
        state between calls as it does there.  One device -> host read of B x 16 floats instead of 3 B tiny launches + clones."""
        state = None if self.theta is None else self.theta.detach().cpu().numpy()
        sm, state = hostglue.ema_scan(thetas.detach().float().cpu().numpy(), state, self.pose_momentum)
        self.theta = torch.from_numpy(state).to(self.device)
        return torch.from_numpy(sm).to(self.device)

    def to_image(self, img_u8_hwc):
        from PIL import Image
        return Image.fromarray(img_u8_hwc)

    # ------------------------------------------------------------------------------------------------------
    def forward(self, source_image=None, driver_image=None, source_mask=None, source_mask_add=0, driver_mask=None,
                crop=True, reset_tracking=False, smooth_pose=False, hard_normalize=False, soft_normalize=False,
                delta_yaw=None, delta_pitch=None, cloth=False, thetas_pass='', theta_n=0, target_theta=True,
                mix=False, mix_old=True, c_source_latent_volume=None, c_target_latent_volume=None,
                custome_target_pose_embed=None, custome_target_theta_embed=None, no_grad_infer=True,
                modnet_mask=False, custome_source_pose_embed=None, custome_source_theta_embed=None,
                custome_idt_embed=None):
        self.no_grad_infer = no_grad_infer
        self.target_theta = target_theta
        with torch.no_grad():
            if reset_tracking:
                self.center = self.size = self.theta = self.delta_yaw = self.delta_pitch = None
                self._crop_tracker = None
            self.mix, self.mix_old = mix, mix_old
            if delta_yaw is not None:
                self.delta_yaw = delta_yaw
            if delta_pitch is not None:
                self.delta_pitch = delta_pitch
            c, d, s = self.cfg["latent_volume_channels"], self.cfg["latent_volume_depth"], self.cfg["latent_volume_size"]

            if source_image is not None:
                if crop:
                    source_img_crop = self._detect_and_crop(source_image)
                else:
                    source_img_crop = self._prepare_image(source_image)
                self.source_image = source_image
                self.source_image_crop = source_img_crop
                # infer.py:408-420: the face-parsing mask (> 0.6) ALWAYS multiplies the crop; source_mask only replaces
                # source_img_mask.  Deviation, only when no 'face_parsing' network is installed (BiSeNet is third party):
                # source_mask then stands in for the face mask as well.
                if 'face_parsing' in self.embedders:
                    face_mask_source = (self.embedders['face_parsing'](source_img_crop) > 0.6).float()  # infer.py:408-411
                elif source_mask is not None:
                    face_mask_source = source_mask.to(self.device).float()
                else:
                    raise RuntimeError("a source call needs source_mask= (or a 'face_parsing' embedder): the reference "
                                       "masks the source with BiSeNet face parsing (infer.py:410-417)")
                source_img_mask = source_mask.to(self.device).float() if source_mask is not None else face_mask_source
                if modnet_mask:
                    source_img_mask = self._need('matting', 'modnet_mask=True')(source_img_crop)
                if source_mask_add:
                    source_img_mask = source_img_mask.clamp(max=1, min=0)
                source_img_crop = (source_img_crop * face_mask_source).float()
                self.source_img_crop_m = source_img_crop
                self.source_img_mask = source_img_mask
                masked = (source_img_crop * source_img_mask).contiguous()
                if custome_idt_embed is not None:
                    self._set_source_cache(idt_embed=custome_idt_embed.to(self.device).float().contiguous())
                else:
                    self._set_source_cache(idt_embed=self._need('idt_embedder', 'a source call')(masked))  # infer.py:432
                if custome_source_theta_embed is not None:
                    pred_source_theta = self._theta_from(custome_source_theta_embed)[0]
                else:
                    pred_source_theta = self._need('head_pose_regressor', 'a source call')(source_img_crop)  # :437
                self.pred_source_theta = pred_source_theta
                if custome_source_pose_embed is not None:
                    source_pose_embed = custome_source_pose_embed.to(self.device).float().contiguous()
                else:
                    source_pose_embed, self.source_img_align = self._expression(source_img_crop, pred_source_theta,
                                                                                'a source call')
                self.pred_source_pose_embed = source_pose_embed
                self.source_img = source_img_crop

                hp = self.hot_path
                source_latents = hp.local_encoder(masked)                                              # infer.py:433
                emb = hp.embed(source_pose_embed, self.idt_embed)                                      # infer.py:459
                delta_xy = hp.xy_generator(emb)                                                        # infer.py:462
                vol = source_latents.view(1, c, d, s, s)
                if self.cfg["source_volume_num_blocks"] > 0:
                    vol = hp.volume_source(vol)                                                        # infer.py:490-491
                self.source_latent_volume = vol if c_source_latent_volume is None else \
                    c_source_latent_volume.to(self.device).float().contiguous()
                inv = ops.mat4_inverse(pred_source_theta.float().contiguous())                         # infer.py:443, on the device
                self._source_theta_inv = inv
                self.source_rotation_warp = ops.affine_grid3d(inv, (d, s, s))                          # infer.py:441-444
                self.source_xy_warp_resize = delta_xy
                rot = ops.grid_sample3d(ops.volume_to_channels_last(self.source_latent_volume), theta=inv, padding_mode=hp.pad,
                                        in_layout="ndhwc", out_layout="ndhwc")                         # infer.py:499-500
                tv = ops.grid_sample3d(rot, delta=delta_xy, padding_mode=hp.pad, in_layout="ndhwc", out_layout="ncdhw")
                self.target_latent_volume_1 = tv if c_target_latent_volume is None else \
                    c_target_latent_volume.to(self.device).float().contiguous()
                self._set_source_cache(canonical=hp.volume_process(self.target_latent_volume_1))      # infer.py:507

            if driver_image is None and custome_target_pose_embed is None:
                return None                                                                            # infer.py:644-646
            if self.target_latent_volume is None:
                raise RuntimeError("call forward with a source_image first (no cached canonical volume)")

            driver_img_crop = None
            if driver_image is not None:
                driver_img_crop = self._detect_and_crop(driver_image) if crop else self._prepare_image(driver_image)
            if custome_target_theta_embed is not None:                                                 # infer.py:565-566
                pred_target_theta, self.pred_target_srt = self._theta_from(custome_target_theta_embed)
            elif driver_img_crop is None:
                raise RuntimeError("forward(driver_image=None, custome_target_pose_embed=...) also needs "
                                   "custome_target_theta_embed=: without a driver frame there is nothing to regress the head "
                                   "pose from (the reference dereferences the missing crop at infer.py:562)")
            else:
                pred_target_theta, *srt = self._head_pose(driver_img_crop)                             # infer.py:562
                self.pred_target_srt = tuple(srt)
            if mix:                                                                                    # infer.py:568-569
                pred_target_theta = self.get_mixing_theta(self.pred_source_theta, pred_target_theta)
            if smooth_pose:                                                                            # infer.py:571-581
                pred_target_theta = self._smooth_thetas(pred_target_theta)
            self.pred_target_theta = pred_target_theta
            theta_used = pred_target_theta if target_theta else self.pred_source_theta
            # the reference runs the expression embedder on every driver frame (infer.py:596-601) and only then overrides
            # its output (:603-604); target_img_align (:608) comes from that run
            self.target_img_align = None
            if driver_img_crop is None and custome_target_pose_embed is None:
                raise RuntimeError("forward(driver_image=None, custome_target_theta_embed=...) also needs "
                                   "custome_target_pose_embed=: without a driver frame there is no expression to embed")
            if driver_img_crop is not None and (custome_target_pose_embed is None or 'expression_embedder' in self.embedders):
                target_pose_embed, self.target_img_align = self._expression(driver_img_crop, pred_target_theta,
                                                                            'a driver call')
            if custome_target_pose_embed is not None:                                                  # infer.py:603-604
                target_pose_embed = custome_target_pose_embed.to(self.device).float().contiguous()
            self.target_pose_embed = target_pose_embed
            B = target_pose_embed.shape[0]
            if theta_used.shape[0] != B:
                theta_used = theta_used.expand(B, -1, -1)
            img = self._drive(target_pose_embed, theta_used.float().contiguous())                      # infer.py:612-637
            u8 = ops.pack_rgb8(img).cpu().numpy()                                                      # infer.py:641-643
            return [self.to_image(u8[i]) for i in range(B)], img

    __call__ = forward

    # ------------------------------------------------------------------------------------------------------
    def animate(self, target_pose_embeds, target_srt, batch_size=16, as_uint8=True):
        """1 source -> N driver frames (the BASELINE metric).  Frames are sharded contiguously across ranks
        (SURVEY.md section 8e); each rank walks its shard in batches of `batch_size`.  Yields (first_frame_index, frames)
        with frames a uint8 [B,H,W,3] (or fp32 [B,3,H,W]) DEVICE tensor -- no host sync inside the loop."""
        if self._canonical_cl is None:
            raise RuntimeError("call forward with a source_image first")
        N = target_pose_embeds.shape[0]
        lo, hi = parallel.shard_range(N, self.rank, self.world)
        for b0 in range(lo, hi, batch_size):
            b1 = min(b0 + batch_size, hi)
            pose = target_pose_embeds[b0:b1].to(self.device).float().contiguous()
            srt = [t[b0:b1].to(self.device).float().contiguous() for t in target_srt]
            theta = ops.pose_theta(*srt)
            img = self._drive(pose, theta)
            yield b0, (ops.pack_rgb8(img) if as_uint8 else img)

    # ------------------------------------------------------------------------------------------------------
    def animate_frames(self, frames, batch_size=16, windows=None, ring=3, to_host=True, smooth_pose=False):
        """Video in -> video out, device resident (notebooks/infer.py:511-556, :562-601, :641-644 per
        frame there).  frames: uint8 [N,H,W,3] tensor (host, ideally pinned, or device) or an iterable of such chunks --
        decoded video frames, uploaded as BYTES.  Per batch, all on the device and without a host synchronisation:
            byte -> fp32 CHW (emo_unpack_rgb8) -> crop windows read in place + bicubic resize to image_size, the whole batch in
            one launch (emo_resize2d_windows_f32; `windows[i] = (x_lo, y_lo, side)` from the face detector +
            hostglue.crop_window, host arithmetic; None = whole frame) -> HeadPoseRegressor -> ExpressionEmbed -> hot path ->
            uint8 HWC.
        The driver-side matte (MODNet) of the reference is computed but unused with use_seg=False (infer.py:592-601): skipped.
        smooth_pose (infer.py:571-581) is a scan over the FRAME ORDER, so it runs before the frames are sharded (SURVEY.md
        section 8e): per chunk, every rank regresses the head pose of its own shard, the thetas (16 floats per frame) are
        gathered on every rank, the EMA runs once on the host over the whole chunk (hostglue.ema_scan, state carried from chunk
        to chunk in `self.theta` exactly as the reference carries it from call to call), and only then does each rank render
        its shard with its slice of the smoothed thetas -- 1 rank and N ranks produce the same frames.  That pass costs one
        host synchronisation per chunk; the crops of the shard stay resident between the two passes (3 MB per frame).
        to_host: results go D2H into a ring of `ring` pinned buffers on a copy stream; a batch is yielded once ITS copy
        event has completed, i.e. the host only ever waits for a batch that is `ring - 1` batches behind the GPU.
        Yields (first_frame_index, uint8 [b,S,S,3]) -- a view of a pinned ring slot, valid ONLY until the generator is resumed
        (the next batch's copy may be queued into the same slot right away: consume or copy it before calling next()) --
        or, with to_host=False, the device tensor.  Frames are sharded contiguously across ranks as in animate()."""
        if self._canonical_cl is None:
            raise RuntimeError("call forward with a source_image first")
        S = self.cfg["image_size"]
        chunks = [frames] if isinstance(frames, torch.Tensor) else frames
        copy_stream = torch.cuda.Stream(device=self.device) if to_host else None
        slots, pending = [], []          # pinned buffers; (first index, slot, n frames, event) in flight

        def drain(keep):
            while len(pending) > keep:
                b0, slot, nb, ev = pending.pop(0)
                ev.synchronize()
                yield b0, slots[slot][:nb]

        upload_stream = torch.cuda.Stream(device=self.device)

        def uploaded(chunk, spans):
            """(b0, b1, uint8 frames on the device) for every span, with the upload of span i + 1 enqueued on a copy stream BEFORE
            span i is handed out -- i.e. before its kernels are enqueued -- so that a host chunk's H2D copy (12.6 MB per 16 frames
            at 512^2: 0.25 ms) runs beside the previous batch's compute instead of in front of its own (on the compute stream the
            copy serialises with the kernels).  Device-resident chunks pass through."""
            ahead = None
            for span in list(spans) + [None]:
                nxt = None
                if span is not None:
                    b0, b1 = span
                    src = chunk[b0:b1]
                    if src.is_cuda:
                        nxt = (b0, b1, src.contiguous(), None)
                    else:
                        with torch.cuda.stream(upload_stream):
                            t = src.to(self.device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(upload_stream)
                        nxt = (b0, b1, t, ev)
                if ahead is not None:
                    p0, p1, t, ev = ahead
                    if ev is not None:
                        torch.cuda.current_stream().wait_event(ev)
                        t.record_stream(torch.cuda.current_stream())
                    yield p0, p1, t
                ahead = nxt

        def crops_of(u8, base, b0, b1):
            x = ops.unpack_rgb8(u8)
            if windows is not None:
                wins = [(w[0], w[1], w[2], w[2]) for w in windows[base + b0:base + b1]]
                return ops.resize2d_windows(x, (S, S), wins, "bicubic", clamp01=True)
            if x.shape[-2:] != (S, S):
                return ops.resize2d(x, (S, S), "bicubic")
            return x

        base, k = 0, 0
        for chunk in chunks:
            if chunk.dtype != torch.uint8 or chunk.dim() != 4 or chunk.shape[-1] != 3:
                raise ValueError("frames must be uint8 [N,H,W,3]")
            n = chunk.shape[0]
            lo, hi = parallel.shard_range(n, self.rank, self.world)
            spans = [(b0, min(b0 + batch_size, hi)) for b0 in range(lo, hi, batch_size)]
            smoothed, kept = None, {}
            if smooth_pose:
                keep_crops = (hi - lo) * 3 * S * S * 4 <= _SMOOTH_KEEP_BYTES
                local = []
                for b0, b1, u8 in uploaded(chunk, spans):
                    crops = crops_of(u8, base, b0, b1)
                    local.append(self._head_pose(crops)[0].clone())
                    if keep_crops:
                        kept[b0] = crops
                local = torch.cat(local) if local else torch.empty((0, 4, 4), device=self.device)
                every = parallel.gather_shards(local, n, self.rank, self.world)        # [n,4,4] on every rank, frame order
                smoothed = self._smooth_thetas(every)[lo:hi]
            # (every span whose crops stayed resident from the head-pose pass needs no second upload)
            todo = [sp for sp in spans if sp[0] not in kept]
            fresh = uploaded(chunk, todo)
            for b0, b1 in spans:
                crops = kept.pop(b0, None)
                if crops is None:
                    f0, f1, u8 = next(fresh)
                    assert (f0, f1) == (b0, b1)
                    crops = crops_of(u8, base, b0, b1)
                theta = smoothed[b0 - lo:b1 - lo] if smoothed is not None else self._head_pose(crops)[0]
                self.pred_target_theta = theta                                   # (as forward() leaves it: infer.py:584)
                pose, _ = self._expression(crops, theta, 'a driver call')
                out = ops.pack_rgb8(self._drive(pose, theta.float().contiguous()))
                if not to_host:
                    yield base + b0, out
                    continue
                if len(slots) < ring:
                    slots.append(torch.empty((batch_size, S, S, 3), dtype=torch.uint8, pin_memory=True))
                slot = k % ring
                k += 1
                copy_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(copy_stream):
                    slots[slot][:b1 - b0].copy_(out, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                out.record_stream(copy_stream)
                pending.append((base + b0, slot, b1 - b0, ev))
                yield from drain(ring - 1)
            base += n
        yield from drain(0)

    def share_source(self, src_rank=0):
        """RCCL broadcast of the per-identity cache computed on `src_rank`: canonical volume
        (25 MB) + idt_embed (32 KB) + source theta."""
        c, d, s = self.cfg["latent_volume_channels"], self.cfg["latent_volume_depth"], self.cfg["latent_volume_size"]
        # idt_embed is [1, idt_output_channels, idt_output_size, idt_output_size] of the checkpoint's embedder config: the
        # receivers learn its shape from the broadcast header; what the warp embedding needs is checked on the source rank
        es = self.cfg["gen_embed_size"]
        cache = parallel.broadcast_source_cache(
            dict(canonical=self.target_latent_volume, idt_embed=getattr(self, 'idt_embed', None),
                 theta_src=getattr(self, 'pred_source_theta', None)),
            shapes=dict(canonical=(1, c, d, s, s), theta_src=(1, 4, 4)), names=['canonical', 'idt_embed', 'theta_src'],
            src=src_rank, device=self.device, world=self.world, rank=self.rank)
        if cache["idt_embed"].numel() != self.cfg["gen_max_channels"] * es * es:
            raise RuntimeError(f"idt_embed {tuple(cache['idt_embed'].shape)} does not match the warp embedding "
                               f"({self.cfg['gen_max_channels']} channels x {es}x{es})")
        self.pred_source_theta = cache["theta_src"]
        self._set_source_cache(canonical=cache["canonical"], idt_embed=cache["idt_embed"])
