"""The multi-GPU exchange path through RCCL itself -- backend "nccl" is RCCL on ROCm -- instead of the gloo
stand-in of tests/test_parallel_cpu.py: init_distributed -> broadcast_source_cache -> max_over_ranks -> barrier ->
destroy_process_group in freshly spawned processes (the reference initialises torch.distributed the same way,
notebooks/infer.py:94-105).  World size 1 on any GPU box; world size 2 when two GPUs are visible."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys, time
sys.path.insert(0, %(root)r)
from emoportraits_amd import parallel          # sets HSA_ENABLE_IPC_MODE_LEGACY before the first HIP call
import torch, torch.distributed as dist
rank, world = parallel.init_distributed(force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl", dist.get_backend()
assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
dev = torch.device("cuda", parallel.local_device_index())
assert torch.cuda.current_device() == dev.index
shapes = dict(canonical=(1, 96, 16, 64, 64), idt_embed=(1, 512, 4, 4), theta_src=(1, 4, 4))
g = torch.Generator().manual_seed(5)
full = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
cache = {k: v.to(dev) for k, v in full.items()} if rank == 0 else {k: None for k in shapes}
for known in (True, False):            # receivers that know the shapes (one collective) and receivers that do not (header first)
    got = parallel.broadcast_source_cache(cache, shapes if (known or rank == 0) else None, src=0, device=dev, world=world, rank=rank,
                                          names=list(shapes), exchange_shapes=not known)
    torch.cuda.synchronize()
    for k in shapes:
        assert got[k].is_cuda and torch.equal(got[k].cpu(), full[k]) and got[k].data_ptr() %% 16 == 0, k
parallel.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
parallel.broadcast_source_cache(cache, shapes, src=0, device=dev, world=world, rank=rank, names=list(shapes), exchange_shapes=False)
torch.cuda.synchronize()
ms = parallel.max_over_ranks(time.perf_counter() - t0, device=dev) * 1e3
assert parallel.max_over_ranks(float(rank + 1), device=dev) == float(world)
parallel.barrier()
if rank == 0:
    print("RCCL_OK world=%%d broadcast_ms=%%.3f" %% (world, ms), flush=True)
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("EMO_DIST_BACKEND", None)
        env.pop("EMO_FORCE_DEVICE", None)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % dict(root=ROOT)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    line = [l for l in outs[0].splitlines() if l.startswith("RCCL_OK")]
    assert line, outs[0][-2000:]
    print("PARITY", line[0])


def test_rccl_single_rank_group_runs_the_exchange_path():
    _run(1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_rccl_two_ranks_two_gpus():
    _run(2)


def test_numa_local_cores_of_this_box():
    """parallel.pin_to_local_cores on the real topology of the box (a child process: the affinity is the process's): the GPU's NUMA
    node is found through sysfs, the rank is bound to cores of THAT node (all of them for one rank; its share of them for two
    ranks on GPUs of one node), and the record says so.  On a box whose sysfs does not tell, the even split is what it reports."""
    code = (
        "import os, sys, json\n"
        "sys.path.insert(0, %r)\n"
        "from emoportraits_amd import parallel\n"
        "import torch\n"
        "torch.cuda.init()\n"
        "nodes, cpus = parallel._gpu_numa_nodes(), parallel._node_cpus()\n"
        "before = sorted(os.sched_getaffinity(0))\n"
        "rec = parallel.pin_to_local_cores(local_rank=0, n_local=1)\n"
        "after = sorted(os.sched_getaffinity(0))\n"
        "two = parallel.plan_affinity(1, 2, [nodes[0], nodes[0]], cpus, before)[0]\n"
        "print(json.dumps(dict(nodes=nodes, n_nodes=len(cpus), rec=rec, n_before=len(before), n_after=len(after), after_in_node="
        "bool(nodes and nodes[0] is not None and nodes[0] in cpus and set(after) <= set(cpus[nodes[0]])), two=len(two))))\n") % ROOT
    env = dict(os.environ)
    for k in ("EMO_FORCE_DEVICE", "EMO_PIN_CORES", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    print("PARITY host affinity on this box:", out)
    assert len(out["nodes"]) == torch.cuda.device_count() and out["rec"]["pinned"] and 0 < out["n_after"] <= out["n_before"]
    if out["nodes"][0] is not None and out["nodes"][0] >= 0 and out["n_nodes"] > 1:
        assert out["after_in_node"] and "numa node" in out["rec"]["how"] and out["n_after"] < out["n_before"]
        assert 0 < out["two"] <= out["n_after"] // 2 + 1
