"""Images in -> images out: per-frame latency / throughput of the whole driver side on the released architecture
(HeadPoseRegressor + ExpressionEmbed + hot path + uint8 packing), eager vs hipGraph replay.  JSON lines.

    python tools/bench_pipeline.py [S=512] [B ...]
"""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import config, random_init  # noqa: E402
from emoportraits_amd import embedders as E  # noqa: E402
from emoportraits_amd.infer import InferenceWrapper  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    Bs = [int(a) for a in sys.argv[2:]] or [1, 16]
    cfg = config.hot_path_config(overrides={"image_size": S})
    ecfg = E.embedder_config()
    sd = random_init.random_state_dict(cfg, seed=0)
    sd.update(E.random_state_dict(E.idt_schema(ecfg), 1))
    sd.update(E.random_state_dict(E.expression_schema(ecfg), 2))
    hp_sd = E.random_state_dict(E.head_pose_schema(), 3)
    hp_sd["fc.weight"] *= 0.05
    hp_sd["fc.bias"] = torch.tensor([1.0, 1.0, 1.0, 0.1, -0.2, 0.05, 0.02, -0.03, 0.01])
    root = tempfile.mkdtemp()
    os.makedirs(os.path.join(root, "logs", "exp", "checkpoints"))
    with open(os.path.join(root, "logs", "exp", "args.txt"), "wt") as f:
        for k, v in {**cfg, **ecfg}.items():
            f.write(f"{k}: {v}\n")
    torch.save(hp_sd, os.path.join(root, "hp.pth"))
    g = torch.Generator().manual_seed(5)
    src = torch.rand(1, 3, S, S, generator=g)
    for use_graphs in (False, True):
        w = InferenceWrapper(experiment_name="exp", model_file_name="x", project_dir=root, folder="logs", state_dict=sd,
                             print_params=False, head_pose_regressor_path=os.path.join(root, "hp.pth"), use_graphs=use_graphs)
        t0 = time.perf_counter()
        w.forward(source_image=src, crop=False, source_mask=torch.ones(1, 1, S, S))
        torch.cuda.synchronize()
        src_ms = (time.perf_counter() - t0) * 1e3
        for B in Bs:
            drv = torch.rand(B, 3, S, S, generator=g).to(w.device)
            for _ in range(3):
                w.forward(driver_image=drv, crop=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 10
            for _ in range(iters):
                imgs, _ = w.forward(driver_image=drv, crop=False)      # includes uint8 packing + D2H + PIL objects
            dt = (time.perf_counter() - t0) / iters
            print(json.dumps(dict(path="forward() per call (PIL out)", S=S, B=B, graphs=use_graphs, ms_per_call=round(dt * 1e3, 3),
                                  fps=round(B / dt, 2), first_source_call_ms=round(src_ms, 1))), flush=True)
            # device-resident frames path (InferenceWrapper.animate_frames): uint8 frames in pinned host memory -> uint8
            # frames in pinned host memory, no .cpu() / host sync inside the loop (D2H ring)
            n_frames = B * 12
            frames = (torch.rand(n_frames, S, S, 3, generator=g) * 255).to(torch.uint8).pin_memory()
            for _ in w.animate_frames(frames[:B * 3], batch_size=B):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = 0
            lat = []
            for first, out in w.animate_frames(frames, batch_size=B):
                got += out.shape[0]
                lat.append(time.perf_counter())
            dt = time.perf_counter() - t0
            assert got == n_frames
            gaps = sorted(b - a for a, b in zip(lat[2:], lat[3:]))
            print(json.dumps(dict(path="animate_frames (uint8 in -> uint8 out, pinned D2H ring)", S=S, B=B, graphs=use_graphs,
                                  fps=round(n_frames / dt, 2), ms_per_batch=round(dt / (n_frames / B) * 1e3, 3),
                                  median_gap_ms=round(gaps[len(gaps) // 2] * 1e3, 3) if gaps else None)), flush=True)


if __name__ == "__main__":
    main()
