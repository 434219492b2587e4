"""Frames in -> frames out through the device-resident path (the reference loops
`InferenceWrapper.forward(driver_image=frame)` over decoded video frames, notebooks/infer.py:511-644).

    python tools/animate_video.py --project <project_dir> --experiment <exp> --checkpoint <file> \
        --source source.png --frames <dir of PNG/JPG frames | frames.npy (uint8 [N,H,W,3])> --out <dir> [--batch 16]
        [--windows windows.json]   # optional per-frame crop windows [[x_lo, y_lo, side], ...] from a face detector

Frame I/O is host work (PIL / numpy): decoded frames are handed to InferenceWrapper.animate_frames as uint8 chunks in pinned
memory; crop, bicubic resize, both embedders, the hot path and the uint8 packing run on the GPU without a host sync, and
finished batches come back through the pinned D2H ring while the next ones are being computed.  Face detection, parsing
and matting are third-party networks without sources in the reference tree: the source image must come with its mask
(--source-mask, default all ones) and crop windows, if any, are precomputed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_frames(path, chunk):
    """yields pinned uint8 [n,H,W,3] chunks"""
    if path.endswith(".npy"):
        arr = np.load(path, mmap_mode="r")
        for a in range(0, arr.shape[0], chunk):
            yield torch.from_numpy(np.ascontiguousarray(arr[a:a + chunk])).pin_memory()
        return
    from PIL import Image
    names = sorted(f for f in os.listdir(path) if f.lower().endswith((".png", ".jpg", ".jpeg")))
    for a in range(0, len(names), chunk):
        imgs = [np.asarray(Image.open(os.path.join(path, f)).convert("RGB")) for f in names[a:a + chunk]]
        yield torch.from_numpy(np.stack(imgs)).pin_memory()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--project", required=True)
    ap.add_argument("--experiment", required=True)
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--folder", default="logs")
    ap.add_argument("--head-pose-regressor", default=None)
    ap.add_argument("--source", required=True)
    ap.add_argument("--source-mask", default=None)
    ap.add_argument("--frames", required=True)
    ap.add_argument("--windows", default=None)
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--graphs", action="store_true")
    a = ap.parse_args()
    from PIL import Image
    from notebooks.infer import InferenceWrapper
    w = InferenceWrapper(experiment_name=a.experiment, model_file_name=a.checkpoint, project_dir=a.project, folder=a.folder,
                         head_pose_regressor_path=a.head_pose_regressor, use_graphs=a.graphs)
    S = w.cfg["image_size"]
    src = Image.open(a.source).convert("RGB")
    mask = torch.ones(1, 1, S, S) if a.source_mask is None else \
        torch.from_numpy(np.asarray(Image.open(a.source_mask).convert("L").resize((S, S)), dtype=np.float32) / 255.0)[None, None]
    w.forward(source_image=src, crop=False, source_mask=mask)
    windows = json.load(open(a.windows)) if a.windows else None
    os.makedirs(a.out, exist_ok=True)
    t0, n = time.perf_counter(), 0
    for first, u8 in w.animate_frames(load_frames(a.frames, 8 * a.batch), batch_size=a.batch, windows=windows):
        arr = u8.numpy()
        for j in range(arr.shape[0]):
            Image.fromarray(arr[j]).save(os.path.join(a.out, f"{first + j:06d}.png"))
        n += arr.shape[0]
    dt = time.perf_counter() - t0
    print(json.dumps(dict(frames=n, seconds=round(dt, 3), fps=round(n / dt, 2), image_size=S, batch=a.batch)))


if __name__ == "__main__":
    main()
