"""LocalEncoder of the source pass (networks/volumetric_avatar/local_encoder.py:26-125) on the HIP kernels."""
import math

from . import ops
from .nets import Norm, ResBlock
from .pack import PackedConv, folded_conv


class LocalEncoder:
    """from_rgb 7x7 (SN, bias) -> num_2d_blocks x ResBlock(stride-2 avg-pool) -> GN+ReLU+1x1 (WS) -> [1, c*d, s, s]"""

    def __init__(self, sd, prefix, cfg, device, image_size=None, latent_size=None, ws=True):
        """image_size / latent_size / ws default to the stage-1 config; stage 2 (local_encoder_old.py:25-117, same
        structure) passes its own.  ws: the WS replacement hit the convs that follow a GroupNorm."""
        S = cfg["image_size"] if image_size is None else image_size
        self.image_size = S
        kind = "ws" if ws else "sn"
        nblk = int(math.log(S // (cfg["latent_volume_size"] if latent_size is None else latent_size), 2))
        w, b = folded_conv(sd, f"{prefix}.from_rgb_{S}px", "sn")
        # the 7x7 2-D conv runs on the same implicit-GEMM kernel as a depth-7 conv over the image ROWS:
        # [N,3,H,W] is viewed as [N,3,D=H,1,W] and the weight [Co,3,7,7] as [Co,3,KD=7,1,7]
        self.from_rgb = PackedConv(f"{prefix}.from_rgb_{S}px", w.unsqueeze(3), b, device)
        self.blocks = []
        s = S
        for i in range(nblk):
            self.blocks.append(ResBlock(sd, f"{prefix}.enc_{i}_block={s}px", kind, device))
            s //= 2
        self.nh = Norm(sd, prefix + ".finale_layers.0", device)
        self.finale = PackedConv.from_state_dict(sd, prefix + ".finale_layers.2", kind, device)

    def __call__(self, img, want_stats=False):
        N, C, H, W = img.shape
        if H != self.image_size or W != self.image_size:
            raise ValueError(f"LocalEncoder was built for {self.image_size}px inputs (layer names embed the size)")
        x, st = ops.conv_igemm(img.view(N, C, H, 1, W), self.from_rgb, want_stats=True)
        x = x.view(N, -1, H, W)
        for blk in self.blocks:
            x = blk(x, down=(2, 2), x_stats=st)
            st = None                       # the pooled block output has no tile statistics
        s, h = self.nh.affine(x)
        return ops.conv_igemm(x, self.finale, s, h, relu_in=True, want_stats=want_stats)
