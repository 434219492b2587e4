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

# RCCL shares device buffers between the ranks of a node through IPC handles; the host driver of these boxes only supports the
# dmabuf flavour (without it: `hipIpcGetMemHandle: invalid argument` at the first collective).  Must be in the environment
# before the HIP runtime initialises, i.e. before the first CUDA call of the process -- hence at import, as a default.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def local_device_index():
    """GPU of this process: LOCAL_RANK (one process per GPU).  EMO_FORCE_DEVICE overrides it -- a test hook that lets
    a world_size-2 run share the single GPU of a test box (with EMO_DIST_BACKEND=gloo; RCCL refuses two ranks per GPU)."""
    forced = os.environ.get("EMO_FORCE_DEVICE")
    if forced is not None:
        return int(forced)
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None, force=False):
    """-> (rank, world).  One process per GPU: rank r is bound to GPU LOCAL_RANK explicitly, the intra-op CPU thread pool is
    capped to the rank's share of the host cores (unless OMP_NUM_THREADS is set).  Without a WORLD_SIZE > 1 environment this
    is a no-op unless `force` (or EMO_DIST_FORCE_INIT=1): then a 1-rank group is created, so that the very same collective
    calls run through RCCL on a single-GPU box (tests/test_rccl_gpu.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = force or os.environ.get("EMO_DIST_FORCE_INIT") == "1"
    if world <= 1 and not force:
        return 0, 1
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("EMO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device_index())
        if "OMP_NUM_THREADS" not in os.environ:
            torch.set_num_threads(max(1, (os.cpu_count() or 1) // max(1, world)))
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


def broadcast_source_cache(cache, shapes=None, src=0, device=None, world=None, rank=None, names=None, exchange_shapes=True):
    """Broadcast the per-identity tensors from `src` as ONE flat fp32 buffer (one RCCL broadcast of ~25 MB instead of
    one per tensor; on xGMI the collective is latency- then link-bound, so fewer, larger messages).

    cache   name -> tensor; values may be None on non-source ranks.
    names   entries to exchange, in order (default: the keys of `shapes`, else of `cache` -- every rank must pass the same).
    shapes  optional name -> expected shape: checked on the source rank.
    exchange_shapes  True (default): receivers need no shapes, they learn them from a small int64 header [ndim_i, dims_i ...]
            that is broadcast first (one more collective and a device -> host read), so e.g. idt_embed may have any
            idt_output_channels / size.  False: EVERY rank passes `shapes` for every name and they are taken as given --
            ONE collective, no host synchronisation (what bench.py does).  All ranks must pass the same flag.
    Returns the dict with every entry present on every rank (views into one buffer on the receivers).  The collectives run
    whenever a process group exists -- also a 1-rank group (RCCL path on a single GPU)."""
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
    if world == 1 and not dist.is_initialized():
        return dict(zip(names, tensors))
    if not dist.is_initialized():
        raise RuntimeError(f"broadcast_source_cache(world={world}) without a process group: call init_distributed() first")
    if world != dist.get_world_size():
        # e.g. InferenceWrapper(num_gpus=1) under torchrun: a collective the other ranks never join would hang
        raise RuntimeError(f"world={world} does not match the process group's {dist.get_world_size()} ranks")
    if not exchange_shapes:
        if shapes is None or any(n not in shapes for n in names):
            raise RuntimeError("exchange_shapes=False needs the shape of every entry on every rank")
        dims = [tuple(shapes[n]) for n in names]            # known on every rank: no header exchange, no host sync
    else:
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
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    """SUM all-reduce of a python number (bench.py: frames delivered by all ranks)"""
    if not dist.is_initialized():
        return value
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return type(value)(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    """destroy the process group this process initialised (RCCL communicators are released in order, not at interpreter exit)"""
    if dist.is_initialized():
        dist.destroy_process_group()
