"""CPU tests of the frame-parallel path with the gloo backend, world_size 2."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emoportraits_amd import parallel  # noqa: E402


def test_shard_range_is_a_contiguous_partition():
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # odd sizes on purpose: entries are padded to 16-byte boundaries inside the single flat broadcast
    shapes = dict(canonical=(1, 4, 2, 3, 3), idt_embed=(1, 7, 3, 3), theta_src=(1, 4, 4))
    g = torch.Generator().manual_seed(5)
    full = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
    cache = full if rank == 0 else {k: None for k in shapes}
    # only the source rank knows (and checks) the shapes; receivers learn them from the broadcast header
    got = parallel.broadcast_source_cache(cache, shapes if rank == 0 else None, src=0, world=w, rank=r, names=list(shapes))
    ok = all(torch.equal(got[k], full[k]) and got[k].data_ptr() % 16 == 0 for k in shapes)
    # shapes known on every rank: a single collective, no header
    got2 = parallel.broadcast_source_cache(cache, shapes, src=0, world=w, rank=r, names=list(shapes), exchange_shapes=False)
    ok = ok and all(torch.equal(got2[k], full[k]) for k in shapes)
    # frame sharding: every rank processes its contiguous slice of 11 "frames"; union must be all frames
    lo, hi = parallel.shard_range(11, r, w)
    # per-frame scalars of the shards (the smooth_pose thetas) gathered on every rank in frame order: ragged shards (6 + 5)
    rows = torch.arange(11 * 16, dtype=torch.float32).view(11, 4, 4)
    every = parallel.gather_shards(rows[lo:hi].clone(), 11, r, w)
    ok = ok and torch.equal(every, rows)
    t = parallel.max_over_ranks(float(rank + 1))
    parallel.barrier()
    q.put((rank, ok, lo, hi, t))
    torch.distributed.destroy_process_group()


def test_broadcast_and_sharding_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "broadcast payload differs across ranks"
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 6, 6, 11)
    assert res[0][4] == res[1][4] == 2.0


def test_missing_source_cache_fails_loudly():
    with pytest.raises(RuntimeError, match="run the source pass"):
        parallel.broadcast_source_cache({"canonical": None}, {"canonical": (1, 2)}, src=0, world=1, rank=0)


def test_gather_shards_checks_its_input():
    assert torch.equal(parallel.gather_shards(torch.ones(5, 2), 5, 0, 1), torch.ones(5, 2))      # one rank: the shard is everything
    with pytest.raises(ValueError, match="its shard"):
        parallel.gather_shards(torch.ones(4, 2), 5, 0, 1)
    with pytest.raises(RuntimeError, match="process group"):
        parallel.gather_shards(torch.ones(3, 2), 5, 0, 2)


def test_affinity_plan_follows_the_gpu_numa_nodes():
    """parallel.plan_affinity on made-up topologies: 8 GPUs on 2 sockets (4 per node) -> disjoint quarter-node shares next to each
    GPU; unknown topology -> an even split; a cgroup-restricted cpu set is respected; never an empty set"""
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = list(range(256))
    shares = [parallel.plan_affinity(r, 8, gpu_nodes, node_cpus, allowed)[0] for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert sorted(c for s in shares for c in s) == allowed                       # disjoint, everything used
    for r, s in enumerate(shares):
        assert set(s) <= set(node_cpus[gpu_nodes[r]])                            # next to its own GPU
    # unknown topology (None / -1) or missing sysfs: even split by local rank
    for nodes in ([None] * 8, [-1] * 8, []):
        s3, how = parallel.plan_affinity(3, 8, nodes, node_cpus if nodes else {}, allowed)
        assert s3 == list(range(96, 128)) and "even split" in how
    # a container that may only use 16 cores of node 0
    s1, _ = parallel.plan_affinity(1, 2, [0, 0], node_cpus, list(range(16)))
    assert s1 == list(range(8, 16))
    # more ranks than cores: everybody keeps what is allowed
    assert parallel.plan_affinity(5, 8, [], {}, [0, 1, 2])[0] == [0, 1, 2]
    assert parallel._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
