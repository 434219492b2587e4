"""Python-side cost of one driver pass: the host code runs against a STUB of libemoportraits_hip.so (every kernel entry point
returns 0 at once; the pack-info queries go to the real library) on CPU tensors, so what is timed is argument checking, launch
planning, output allocation and the ctypes calls -- no GPU needed.

    python tools/host_overhead.py [512] [16]   ->   ms per driver pass, launches per entry point, cProfile top list
"""
import collections
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import hip  # noqa: E402

QUERIES = ("emo_groupnorm_workspace_bytes", "emo_conv_pack_info", "emo_conv_pack_info_f16", "emo_conv_pack_info_bf16x3",
           "emo_conv_tile_positions", "emo_abi_version", "emo_conv_igemm_ksplit")


class StubLibrary:
    """kernel entry points return EMO_OK without doing anything and are counted; size / layout queries are forwarded"""

    def __init__(self, real):
        self.real, self.calls = real, collections.Counter()

    def __getattr__(self, name):
        if name in QUERIES:
            return getattr(self.real, name)

        def entry(*args):
            self.calls[name] += 1
            return 0
        return entry


def install_stub(monkeypatch_setattr=setattr):
    """route emoportraits_amd through a StubLibrary; returns it.  `monkeypatch_setattr(obj, name, value)` lets a test undo it."""
    from emoportraits_amd import ops
    stub = StubLibrary(hip.load())
    monkeypatch_setattr(hip, "load", lambda: stub)
    monkeypatch_setattr(hip, "require_cuda_f32", lambda *a, **k: None)
    monkeypatch_setattr(hip, "current_stream", lambda: None)
    monkeypatch_setattr(ops, "_lattice", lambda n, idx: torch.linspace(-1, 1, n))
    monkeypatch_setattr(ops, "_ada_views", lambda ag, ab, N, C: 0 if ag is None else ag.stride(0))
    monkeypatch_setattr(torch.cuda, "current_device", lambda: 0)
    return stub


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    stub = install_stub()
    from emoportraits_amd import config, nets, random_init
    cfg = config.hot_path_config(overrides={"image_size": S})
    sd = random_init.trained_like_state_dict(cfg, seed=0, with_source=False)
    hp = nets.HotPath(sd, cfg, "cpu", with_source=False)
    c, d, s = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
    ccl = hp.prepare_canonical(torch.empty(1, c, d, s, s))
    idt = torch.randn(1, cfg["gen_max_channels"], 4, 4)
    pose = torch.randn(B, cfg["lpe_output_channels_expression"])
    theta = torch.eye(4)[None].repeat(B, 1, 1).contiguous()
    for _ in range(2):
        hp.driver_pass(ccl, idt, pose, theta)
    stub.calls.clear()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        hp.driver_pass(ccl, idt, pose, theta)
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"host ms per driver pass (R{S}, {B} frames, conv mode {hp.precision}): {ms:.2f}")
    print("launches per pass:", {k: v // n for k, v in sorted(stub.calls.items())}, "=", sum(stub.calls.values()) // n)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        hp.driver_pass(ccl, idt, pose, theta)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)


if __name__ == "__main__":
    main()
