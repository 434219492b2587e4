"""Frame-parallel multi-GPU execution -- SURVEY.md section 8(e).

The unit of work is one driver frame; the per-frame warp/decode has no cross-frame dependence, so N driver frames are
split into contiguous per-rank shards (contiguous so that optional scan-like state such as pose EMA stays local) and
there is NO data-path collective.  The only exchange is once per source identity: the rank that ran the source pass
broadcasts {canonical volume [1,96,16,64,64] = 25.2 MB, idt_embed 32 KB, theta_src} over RCCL/xGMI (a flat broadcast
is ~0.17 ms link-bound -- recomputing the 3.9 TFLOP source pass on every rank would cost far more).

One process per GPU; rendezvous through the usual env:// variables (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
The reference initialises torch.distributed the same way (notebooks/infer.py:94-105) but never shards frames.
"""
import os

import torch
import torch.distributed as dist


def local_device_index():
    """GPU of this process: LOCAL_RANK (one process per GPU).  EMO_FORCE_DEVICE overrides it -- a test hook that lets
    a world_size-2 run share the single GPU of a test box (with EMO_DIST_BACKEND=gloo; RCCL refuses two ranks per GPU)."""
    forced = os.environ.get("EMO_FORCE_DEVICE")
    if forced is not None:
        return int(forced)
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """-> (rank, world).  No-op without a WORLD_SIZE > 1 environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("EMO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_device_index())
        dist.init_process_group(backend=backend, init_method="env://")
    return dist.get_rank(), dist.get_world_size()


def shard_range(n_items, rank, world):
    """contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one, earlier ranks get the extra"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_source_cache(cache, shapes, src=0, device=None, world=None, rank=None):
    """Broadcast the per-identity tensors from `src`.  `cache` values may be None on non-source ranks; `shapes`
    gives the (static) shape of each entry.  Returns the dict with every entry present on every rank."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    out = {}
    for name, shape in shapes.items():
        t = cache.get(name)
        if rank == src:
            if t is None:
                raise RuntimeError(f"rank {src} has no '{name}' to broadcast: run the source pass there first")
            t = t.to(device).float().contiguous() if device is not None else t.float().contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"'{name}' has shape {tuple(t.shape)}, expected {tuple(shape)}")
        else:
            t = torch.empty(shape, dtype=torch.float32, device=device if device is not None else "cpu")
        if world > 1:
            dist.broadcast(t, src=src)
        out[name] = t
    return out


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (timing aggregation of bench.py)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
