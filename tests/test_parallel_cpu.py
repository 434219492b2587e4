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
