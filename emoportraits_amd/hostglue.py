"""Host-side scalar logic of the wrapper around the hot path: face-box -> square crop window with
its tracking state, and the source/driver pose mixing.  In the reference these are numpy / scipy code on the host as well
(notebooks/infer.py); here they are kept apart from the kernels so that they can be pinned on CPU against outputs of the
reference's own methods (tests/golden/hostglue.pt, oracle/make_golden.py).
"""
import numpy as np


def detection_to_face(rel_xmin, rel_ymin, rel_width, rel_height, img_w, img_h):
    """mediapipe relative bounding box -> the pixel box the reference crops around (notebooks/infer.py:385-391, :529-536):
    the top edge is pulled up (ymin * 0.9), the bottom edge pushed down (height * 1.2) and clamped to the image"""
    return np.array([img_w * rel_xmin,
                     img_h * rel_ymin * 0.9,
                     img_w * (rel_xmin + rel_width),
                     min(img_h * (rel_ymin + rel_height * 1.2), img_h - 1)])


def remove_overflow(center, size, w, h):
    """notebooks/infer.py:243-261: shrink a square window symmetrically until it fits the w x h image; returns the new
    (even) side length"""
    half = size / 2
    box = np.asarray([center[0] - half, center[1] - half, center[0] + half, center[1] + half], dtype=np.float64)
    over = max(0.0 if box[0] >= 0 else -box[0], 0.0 if box[1] >= 0 else -box[1],
               0.0 if box[2] <= w else box[2] - w, 0.0 if box[3] <= h else box[3] - h)
    box[:2] += over
    box[2:] -= over
    side = int((box[2] - box[0] + box[3] - box[1]) / 2)
    return side - side % 2


class CropTracker:
    """state of `use_smoothed_crop` (notebooks/infer.py:317-327): exponential moving average of the box centre and size
    over the frame sequence; `fixed_bounding_box` freezes the first one"""

    def __init__(self, momentum=0.01, fixed_bounding_box=False):
        self.momentum, self.fixed = momentum, fixed_bounding_box
        self.center = self.size = None

    def reset(self):
        self.center = self.size = None

    def update(self, center, size):
        if self.center is None:
            self.center, self.size = center, size
        elif not self.fixed:
            self.center = center * self.momentum + self.center * (1 - self.momentum)
            self.size = size * self.momentum + self.size * (1 - self.momentum)
        return self.center, self.size


def crop_window(face, img_w, img_h, tracker=None, scale=1):
    """notebooks/infer.py:301-352 (crop_image), the per-frame arithmetic: face box (x0, y0, x1, y1) -> (x_lo, y_lo, side,
    face_scale) of the square that is then resized to image_size.  Returns None for a missing face."""
    if face is None:
        return None
    center = np.asarray([(face[2] + face[0]) // 2, (face[3] + face[1]) // 2])
    size = (face[2] - face[0] + face[3] - face[1]) * scale
    if tracker is not None:
        center, size = tracker.update(center, size)
    center = center.round().astype(int)
    size = int(round(size))
    size -= size % 2
    side = remove_overflow(center, size, img_w, img_h)
    return int(center[0] - side // 2), int(center[1] - side // 2), side, side / size


def mixing_theta(source_theta, target_theta, mix_old=True):
    """notebooks/infer.py:686-736 (get_mixing_theta): keep the source's stretch (scale/shear from the polar decomposition of
    its linear part) and take rotation + translation from the driver.  numpy [B,>=3,4] x [B*T,>=3,4] -> float64 [B*T,3,4].
    As in the reference the driver poses are rolled by one along the batch axis of the sources (a no-op for one source)."""
    from scipy import linalg
    source_theta = np.asarray(source_theta, dtype=np.float64)[:, :3, :]
    target_theta = np.asarray(target_theta, dtype=np.float64)[:, :3, :]
    B = source_theta.shape[0]
    T = target_theta.shape[0] // B
    target_theta = np.roll(target_theta.reshape(B, T, 3, 4), 1, axis=0).reshape(B * T, 3, 4)

    def homogeneous(t):
        m = np.tile(np.eye(4), (t.shape[0], 1, 1))
        m[:, :3, :] = t
        return m

    src, tgt = homogeneous(source_theta), homogeneous(target_theta)
    translation = np.tile(np.eye(4), (B * T, 1, 1))
    translation[:, :3, 3] = tgt[:, :3, 3]
    src_lin, tgt_lin = src.copy(), tgt.copy()
    src_lin[:, :3, 3] = 0
    tgt_lin[:, :3, 3] = 0
    out = []
    for b in range(B):
        try:
            _, src_stretch = linalg.polar(src_lin[b])
        except Exception:                                   # decomposition failed: fall back to the driver pose (:718-719)
            out += [tgt[b * T + t] for t in range(T)]
            continue
        for t in range(T):
            i = b * T + t
            try:
                tgt_rot, tgt_stretch = linalg.polar(tgt_lin[i])
            except Exception:
                out.append(src_stretch)                                                     # :724-725
                continue
            if mix_old:
                out.append(translation[i] @ tgt_rot @ src_stretch)                          # :727-728
            else:
                out.append(src_stretch * tgt_stretch.mean() / src_stretch.mean() @ tgt_rot @ translation[i])   # :729-730
    return np.stack(out)[:, :3]


def ema_scan(values, state, momentum):
    """The `smooth_pose` recurrence of notebooks/infer.py:571-581 over a whole clip on the host:
        theta_i = pred[i] * momentum + theta_{i-1} * (1 - momentum),   theta_{-1} = state, or pred[0] on the first call
    values [n, ...] fp32 (n x 16 floats: cheap), state [...] or None -> (smoothed [n, ...], new state).  fp32 element for
    element in the reference's operation order (two rounded products, one rounded sum; `1 - momentum` formed in double and
    rounded to fp32 once, as torch does with a Python scalar), so the result is bit-identical to the reference's per-frame loop
    of device ops -- and, being a scan over the FRAME ORDER, it has to run before the frames are sharded across ranks
    (SURVEY.md section 8e)."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    m, om = np.float32(momentum), np.float32(1 - momentum)
    cur = v[0].copy() if state is None else np.asarray(state, dtype=np.float32).reshape(v.shape[1:])
    out = np.empty_like(v)
    for i in range(v.shape[0]):
        cur = v[i] * m + cur * om
        out[i] = cur
    return out, cur
