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


def _parse_cpulist(text):
    """'0-15,64-79' -> [0, ..., 15, 64, ..., 79] (the sysfs cpulist format)"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


def plan_affinity(local_rank, n_local, gpu_nodes, node_cpus, allowed):
    """Host cores for the rank that drives GPU `local_rank` of a node with `n_local` ranks (pure function: tests run it on
    made-up topologies).  gpu_nodes[i] = NUMA node of GPU i (None / -1: unknown), node_cpus = {node: [cpus]}, allowed = the cpus
    this process may run on.  A rank gets the cores of its GPU's NUMA node, split evenly between the ranks whose GPUs hang off
    the same node (in local-rank order), so that the pinned staging rings and the launch thread sit next to the GPU's PCIe root
    and no two ranks share a core; with an unknown topology the allowed cores are split evenly by local rank instead.
    -> (sorted cpu list, how)"""
    allowed = sorted(allowed)
    node = gpu_nodes[local_rank] if local_rank < len(gpu_nodes) else None
    if node is not None and node >= 0 and node in node_cpus:
        mine = [c for c in node_cpus[node] if c in set(allowed)]
        peers = [r for r in range(n_local) if r < len(gpu_nodes) and gpu_nodes[r] == node]
        if mine and local_rank in peers and len(mine) >= len(peers):
            k, per = peers.index(local_rank), len(mine) // len(peers)
            return mine[k * per:(k + 1) * per], f"numa node {node}: share {k + 1} of {len(peers)}"
    per = len(allowed) // max(1, n_local)
    if per == 0:
        return allowed, "all allowed cores (fewer cores than ranks)"
    return allowed[local_rank * per:(local_rank + 1) * per], f"even split of the allowed cores: share {local_rank + 1} of {n_local}"


def _gpu_numa_nodes():
    """NUMA node of every visible GPU from sysfs (PCI bus id -> /sys/bus/pci/devices/<id>/numa_node); None where unknown"""
    nodes = []
    for i in range(torch.cuda.device_count()):
        node = None
        try:
            pr = torch.cuda.get_device_properties(i)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                node = int(f.read().strip())
        except Exception:
            node = None
        nodes.append(node)
    return nodes


def _node_cpus():
    out = {}
    base = "/sys/devices/system/node"
    try:
        for name in os.listdir(base):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(base, name, "cpulist")) as f:
                    out[int(name[4:])] = _parse_cpulist(f.read())
    except OSError:
        pass
    return out


_affinity = None      # what pin_to_local_cores() did in this process (bench.py puts it into its record)


def pin_to_local_cores(local_rank=None, n_local=None):
    """Bind this process to the host cores next to its GPU (plan_affinity) -- BEFORE the pinned staging buffers are allocated
    (first-touch places their pages on the node of the touching core) and before the intra-op thread pool is sized.  Multi-rank
    runs only; EMO_PIN_CORES=0 turns it off.  -> the record, also kept for affinity_record()."""
    global _affinity
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if n_local is None:
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    rec = {"pinned": False, "local_rank": local_rank, "n_local": n_local}
    if os.environ.get("EMO_PIN_CORES", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        rec["how"] = "off"
    else:
        try:
            allowed = sorted(os.sched_getaffinity(0))
            nodes = _gpu_numa_nodes() if torch.cuda.is_available() else []
            forced = os.environ.get("EMO_FORCE_DEVICE")
            if forced is not None:            # ranks sharing one GPU (test hook): the topology says nothing, split evenly
                nodes = []
            cpus, how = plan_affinity(local_rank, n_local, nodes, _node_cpus(), allowed)
            os.sched_setaffinity(0, cpus)
            rec.update(pinned=True, how=how, n_cpus=len(cpus), cpus=f"{cpus[0]}-{cpus[-1]}" if cpus else "",
                       gpu_numa_node=nodes[local_rank] if local_rank < len(nodes) else None)
        except Exception as e:                # affinity is an optimisation: never fail a run over it
            rec["how"] = f"failed: {e}"
    _affinity = rec
    return rec


def affinity_record():
    return _affinity


def init_distributed(backend=None, force=False):
    """-> (rank, world).  One process per GPU: rank r is bound to GPU LOCAL_RANK explicitly and to the host cores of that GPU's
    NUMA node (pin_to_local_cores), the intra-op CPU thread pool is capped to the rank's share of the host cores (unless
    OMP_NUM_THREADS is set).  Without a WORLD_SIZE > 1 environment this
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
        if world > 1:
            # every rank on the cores next to its GPU, before anything allocates pinned memory (risk 2)
            pin_to_local_cores()
        if "OMP_NUM_THREADS" not in os.environ:
            share = len(os.sched_getaffinity(0)) if (_affinity or {}).get("pinned") else (os.cpu_count() or 1) // max(1, world)
            torch.set_num_threads(max(1, share))
        dist.init_process_group(backend=backend, init_method="env://")
    return dist.get_rank(), dist.get_world_size()


def shard_range(n_items, rank, world):
    """contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one, earlier ranks get the extra"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_shards(local, n_items, rank=None, world=None):
    """Rows of the contiguous shards (shard_range) of every rank -> the whole [n_items, ...] tensor on EVERY rank, in frame
    order.  Used for per-frame SCALARS only (the head-pose thetas of a clip, 16 floats per frame, in front of the smooth_pose
    scan -- SURVEY.md section 8e caveat); frames themselves never travel.  One all_gather of equal-size padded shards."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_items, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, its shard of {n_items} is [{lo}, {hi})")
    if world == 1:
        return local
    if not dist.is_initialized():
        raise RuntimeError(f"gather_shards(world={world}) without a process group: call init_distributed() first")
    if world != dist.get_world_size():
        raise RuntimeError(f"world={world} does not match the process group's {dist.get_world_size()} ranks")
    per = -(-n_items // world)                                  # the largest shard
    mine = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    mine[:hi - lo].copy_(local)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    rows = []
    for r in range(world):
        a, b = shard_range(n_items, r, world)
        rows.append(parts[r][:b - a])
    return torch.cat(rows)


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
