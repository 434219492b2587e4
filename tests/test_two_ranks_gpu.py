"""The PRODUCT path under world_size 2 and 8 (notebooks/infer.py:94-105 is how the reference initialises
torch.distributed, it never shards frames): two freshly spawned processes -- on the one GPU of a test box through the gloo
backend (EMO_FORCE_DEVICE=0, EMO_DIST_BACKEND=gloo; RCCL refuses two ranks per device), on two GPUs through RCCL -- each build
an InferenceWrapper(num_gpus=2, use_graphs=True); rank 0 alone runs the source pass, both call share_source(), both run
animate() and animate_frames() on the same 33 driver frames.  The parent asserts that every rank produced exactly its
contiguous shard, that the shards tile the frame range, and that their union equals BIT FOR BIT what ONE rank produces when
it walks the same shards itself (same batches, hence the same launch plans; against the unsharded sweep, whose batches differ,
the frames agree to one uint8 level)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

N_FRAMES = 33

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from emoportraits_amd import parallel
import torch
from notebooks.infer import InferenceWrapper
from test_infer_gpu import _toy_embedders
tiny = torch.load(os.path.join(%(root)r, "tests", "golden", "tiny_hotpath.pt"), weights_only=False)
num_gpus = int(os.environ["WORLD_SIZE"])
w = InferenceWrapper(experiment_name="exp", model_file_name="model.pth", project_dir=%(project)r, folder="logs",
                     print_params=False, num_gpus=num_gpus, use_graphs=True)
assert (w.rank, w.world) == (int(os.environ["RANK"]), num_gpus)
w.embedders.update(_toy_embedders(tiny, w.device))
S = tiny["cfg"]["image_size"]
if w.rank == 0:                      # the source pass runs on ONE rank ...
    w.forward(source_image=tiny["img"], crop=False, source_mask=torch.ones(1, 1, S, S), custome_idt_embed=tiny["idt_embed"],
              custome_source_pose_embed=tiny["source_pose_embed"], custome_source_theta_embed=tiny["theta_src"])
else:
    assert w.target_latent_volume is None
if num_gpus > 1:
    w.share_source(src_rank=0)       # ... and its cache reaches the others by one broadcast
assert w._canonical_cl is not None
N = %(n)d
g = torch.Generator().manual_seed(17)
pose = torch.randn(N, tiny["target_pose_embed"].shape[1], generator=g) * 0.5
srt = (1 + 0.05 * torch.randn(N, 3, generator=g), 0.3 * torch.randn(N, 3, generator=g), 0.05 * torch.randn(N, 3, generator=g))
frames = (torch.rand(N, S, S, 3, generator=g) * 255).to(torch.uint8)
out = {"animate": {}, "animate_frames": {}, "rank": w.rank, "world": w.world}
# EMULATE_WORLD (single-rank reference only): the same contiguous shards, walked one after the other by this one process --
# the batches a frame travels in (and with them the launch plan of every kernel: K split, tile choice) are then the same as
# in the sharded run, which is what makes a BIT-FOR-BIT comparison meaningful; the unsharded sweep is kept beside it
emulate = int(os.environ.get("EMULATE_WORLD", "0"))
spans = [parallel.shard_range(N, r, emulate) for r in range(emulate)] if emulate else [None]
for span in spans:
    sl = slice(None) if span is None else slice(*span)
    off = 0 if span is None else span[0]
    for rep in range(2):                 # second sweep: graph replay
        for b0, u8 in w.animate(pose[sl], [t[sl] for t in srt], batch_size=4):
            for j in range(u8.shape[0]):
                out["animate"][off + b0 + j] = u8[j].cpu()
    for b0, u8 in w.animate_frames(frames[sl], batch_size=4, ring=2):
        for j in range(u8.shape[0]):
            out["animate_frames"][off + b0 + j] = u8[j].clone()
# smooth_pose (notebooks/infer.py:571-581) is a scan over the FRAME ORDER: two chunks of 16 frames, EMA state carried from chunk
# to chunk; every rank must render its shard with the thetas of the GLOBAL scan (SURVEY.md section 8e)
sm = {"frames": {}, "theta": {}}
w.theta = None
for b0, u8 in w.animate_frames([frames[:16], frames[16:32]], batch_size=4, smooth_pose=True, to_host=False):
    th = w.pred_target_theta.cpu()
    for j in range(u8.shape[0]):
        sm["frames"][b0 + j] = u8[j].cpu()
        sm["theta"][b0 + j] = th[j].clone()
out["smooth"] = sm
if w.rank == 0:
    x = frames[:32].permute(0, 3, 1, 2).float().div(255.0).to(w.device)
    out["pred_theta"] = torch.cat([w.embedders["head_pose_regressor"](x[i:i + 4].contiguous(), True)[0].cpu() for i in range(0, 32, 4)])
    out["pose_momentum"] = w.pose_momentum
if emulate:
    out["unsharded"] = {}
    for b0, u8 in w.animate(pose, srt, batch_size=4):
        for j in range(u8.shape[0]):
            out["unsharded"][b0 + j] = u8[j].cpu()
torch.save(out, os.path.join(%(project)r, "rank%%d_of%%d.pt" %% (w.rank, w.world)))
parallel.barrier()
parallel.shutdown()
print("WORKER_OK", w.rank, flush=True)
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _project(tmp_path, golden_dir):
    from emoportraits_amd import config
    tiny = torch.load(os.path.join(golden_dir, "tiny_hotpath.pt"), weights_only=False)
    exp = tmp_path / "logs" / "exp"
    (exp / "checkpoints").mkdir(parents=True)
    cfg = config.hot_path_config(overrides=tiny["cfg"])
    with open(exp / "args.txt", "wt") as f:
        for k, v in cfg.items():
            f.write(f"{k}: {v}\n")
        f.write("experiment_name: exp\nuse_seg: True\n")
    torch.save(tiny["state_dict"], exp / "checkpoints" / "model.pth")
    return str(tmp_path)


def _spawn(world, project, share_gpu, emulate=0):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   EMULATE_WORLD=str(emulate))
        for k in ("EMO_DIST_BACKEND", "EMO_FORCE_DEVICE", "EMO_DIST_FORCE_INIT"):
            env.pop(k, None)
        if share_gpu and world > 1:
            env.update(EMO_FORCE_DEVICE="0", EMO_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % dict(root=ROOT, project=project, n=N_FRAMES)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "WORKER_OK" in o, o[-4000:]
    return [torch.load(os.path.join(project, f"rank{r}_of{world}.pt"), weights_only=False) for r in range(world)]


def _check(world, tmp_path, golden_dir, share_gpu):
    from emoportraits_amd import parallel
    project = _project(tmp_path, golden_dir)
    single = _spawn(1, project, share_gpu, emulate=world)[0]
    assert sorted(single["animate"]) == list(range(N_FRAMES)) == sorted(single["animate_frames"]) == sorted(single["unsharded"])
    # (the unsharded sweep batches the frames differently -- other launch plans, other fp32 rounding: at most one uint8 level)
    worst = max(int((single["animate"][i].int() - single["unsharded"][i].int()).abs().max()) for i in range(N_FRAMES))
    assert worst <= 1, worst
    ranks = _spawn(world, project, share_gpu)
    for kind in ("animate", "animate_frames"):
        covered = []
        for r, out in enumerate(ranks):
            lo, hi = parallel.shard_range(N_FRAMES, r, world)
            assert sorted(out[kind]) == list(range(lo, hi)), (kind, r, sorted(out[kind]))     # its contiguous shard, nothing else
            covered += list(out[kind])
            for i, frame in out[kind].items():
                assert torch.equal(frame, single[kind][i]), f"{kind}: frame {i} of rank {r} differs from the single-rank run"
        assert sorted(covered) == list(range(N_FRAMES))                                       # the shards tile the range
    # smooth_pose: the thetas every rank rendered with are the global scan's -- a replay of the reference's loop
    # (notebooks/infer.py:571-581: `self.theta = pred[i] * m + self.theta * (1 - m)`, state carried across the two chunks), in torch
    # on the CPU, bit for bit -- whatever the world size; the frames equal the 1-rank run's bit for bit when the shards' batches
    # line up with its batches (16-frame chunks, batch 4: world 2 and 4), to one uint8 level otherwise (other launch plans)
    pred, m = single["pred_theta"], single["pose_momentum"]
    state, replay = pred[0].clone(), []
    for i in range(32):
        state = pred[i] * m + state * (1 - m)
        replay.append(state.clone())
    aligned = 16 % (world * 4) == 0
    seen = []
    for r, out in enumerate(ranks):
        spans = [parallel.shard_range(16, r, world) for _ in range(2)]
        want = [c * 16 + i for c, (lo, hi) in enumerate(spans) for i in range(lo, hi)]
        assert sorted(out["smooth"]["frames"]) == want, (r, sorted(out["smooth"]["frames"]))
        seen += want
        for i in want:
            assert torch.equal(out["smooth"]["theta"][i], replay[i]), f"smooth_pose: theta of frame {i} on rank {r} is not the global scan's"
            assert torch.equal(single["smooth"]["theta"][i], replay[i])
            d = int((out["smooth"]["frames"][i].int() - single["smooth"]["frames"][i].int()).abs().max())
            assert d == 0 if aligned else d <= 1, (i, r, d)
    assert sorted(seen) == list(range(32))
    assert not torch.equal(replay[16], pred[16])                      # (the scan does something, and it crosses the chunk boundary)
    print(f"PARITY two-rank product path: world {world}, {N_FRAMES} frames, animate + animate_frames bit-identical to 1 rank; "
          f"smooth_pose thetas = the reference's sequential EMA on every rank, frames {'bit-identical' if aligned else 'within 1 level'}")


def test_two_ranks_share_one_gpu_gloo(tmp_path, golden_dir):
    _check(2, tmp_path, golden_dir, share_gpu=True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_ranks_two_gpus_rccl(tmp_path, golden_dir):
    _check(2, tmp_path, golden_dir, share_gpu=False)


def test_eight_ranks_share_one_gpu_gloo(tmp_path, golden_dir):
    """world 8 with a frame count that 8 does not divide (33 = 5 + 7 x 4: ragged shards, a 1-frame tail batch on rank 0): eight
    co-scheduled processes, each with its own packed weights, graphs and pinned ring"""
    _check(8, tmp_path, golden_dir, share_gpu=True)


@pytest.mark.skipif(torch.cuda.device_count() < 8, reason="needs eight GPUs")
def test_eight_ranks_eight_gpus_rccl(tmp_path, golden_dir):
    _check(8, tmp_path, golden_dir, share_gpu=False)


def _bench_line(args, world, share_gpu=True):
    import json
    env = dict(os.environ)
    for k in ("EMO_DIST_BACKEND", "EMO_FORCE_DEVICE", "EMO_DIST_FORCE_INIT", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if share_gpu and world > 1:
        env.update(EMO_FORCE_DEVICE="0", EMO_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--no-cpu-baseline", "--no-extras"] + args,
                       env=env, capture_output=True, text=True, timeout=1500)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[0])


def test_bench_strong_scaling_line_two_ranks():
    """bench.py --total-frames (BASELINE configs[3]: a fixed clip, contiguous shards of DISTINCT frames, source pass + broadcast +
    D2H inside the timed region) under 2 ranks on the one GPU of the box (gloo): a valid line that says scaling: strong; and the
    weak-scaling form of the same launch carries the strong figure beside its headline.  R256 keeps the test short."""
    rec = _bench_line(["--image-size", "256", "--batch", "4", "--steps", "2", "--warmup", "1", "--total-frames", "21"], 2)
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["unit"] == "frames/s"
    st = rec["strong_scaling"]
    assert st["total_frames"] == 21 and st["frames_per_rank"] == [11, 10] and st["clips"] == 2
    assert abs(rec["value"] - st["frames_per_s"]) < 1e-6 and rec["value"] > 0
    assert rec["weak_scaling"]["scaling"] == "weak" and rec["roofline"]["achieved"] > 0
    rec = _bench_line(["--image-size", "256", "--batch", "4", "--steps", "2", "--warmup", "1", "--strong-frames", "13"], 2)
    assert rec["scaling"] == "weak" and rec["strong_scaling"]["total_frames"] == 13 and rec["strong_scaling"]["frames_per_rank"] == [7, 6]
