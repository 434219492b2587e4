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


_MAX_DIMS = 6


def broadcast_source_cache(cache, shapes=None, src=0, device=None, world=None, rank=None, names=None):
    """Broadcast the per-identity tensors from `src` as ONE flat fp32 buffer (one RCCL broadcast of ~25 MB instead of
    one per tensor; on xGMI the collective is latency- then link-bound, so fewer, larger messages).

    cache   name -> tensor; values may be None on non-source ranks.
    names   entries to exchange, in order (default: the keys of `shapes`, else of `cache` -- every rank must pass the same).
    shapes  optional name -> expected shape: checked on the source rank.  Receivers need no shapes: a small int64 header
            [n, ndim_i, dims_i ...] is broadcast first, so e.g. idt_embed may have any idt_output_channels / size.
    Returns the dict with every entry present on every rank (views into one buffer on the receivers)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if names is None:
        names = list(shapes) if shapes is not None else list(cache)
    dev = device if device is not None else "cpu"
    header = torch.zeros(len(names) * (1 + _MAX_DIMS), dtype=torch.int64, device=dev)
    tensors = []
    if rank == src:
        rows = []
        for name in names:
            t = cache.get(name)
            if t is None:
                raise RuntimeError(f"rank {src} has no '{name}' to broadcast: run the source pass there first")
            t = t.to(dev).float().contiguous()
            if shapes is not None and name in shapes and tuple(t.shape) != tuple(shapes[name]):
                raise RuntimeError(f"'{name}' has shape {tuple(t.shape)}, expected {tuple(shapes[name])}")
            if t.dim() > _MAX_DIMS:
                raise RuntimeError(f"'{name}' has more than {_MAX_DIMS} dimensions")
            tensors.append(t)
            rows += [t.dim()] + list(t.shape) + [0] * (_MAX_DIMS - t.dim())
        header.copy_(torch.tensor(rows, dtype=torch.int64))
    if world == 1:
        return dict(zip(names, tensors))
    dist.broadcast(header, src=src)
    h = header.cpu().tolist()
    dims = []
    for i in range(len(names)):
        row = h[i * (1 + _MAX_DIMS):(i + 1) * (1 + _MAX_DIMS)]
        dims.append(tuple(row[1:1 + row[0]]))
    sizes = [int(torch.Size(d).numel()) for d in dims]
    padded = [(n + 3) // 4 * 4 for n in sizes]          # every entry starts 16-byte aligned (the kernels require it)
    flat = torch.zeros(sum(padded), dtype=torch.float32, device=dev)
    if rank == src:
        off = 0
        for t, n, pn in zip(tensors, sizes, padded):
            flat[off:off + n].copy_(t.reshape(-1))
            off += pn
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for name, d, n, pn in zip(names, dims, sizes, padded):
        out[name] = flat[off:off + n].view(d)
        off += pn
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
