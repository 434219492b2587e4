#!/usr/bin/env python
"""Benchmark of the MI355X hot path: reenactment frames/sec, 1 source -> N driver frames (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--image-size 512] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the per-frame hot path (a3, a4, a5-uv, a1 x2 with a2 fused, a9, a11) over one batch of
B synthetic driver frames that share one source identity, inputs resident in HBM.  Workload at N=1: released
architecture at 512x512 (BASELINE configs[2]: full HIP path; configs[1]'s 64-driver batch is a sampler parity/bench case
in tests/ + tools/bench_sampler.py).  Random-init weights in the reference key layout (the released checkpoint is not in
the repo), synthetic embeddings/poses.  Multi-GPU: frames shard across ranks, per-GPU batch fixed ("weak"); the source
pass runs on rank 0 and its canonical volume is broadcast over RCCL once per identity, outside the timed region
(reported as source_pass_ms / broadcast_ms).

The JSON line also carries
  roofline      the dominant kernel -- conv_igemm_bf16x3_kernel (fp32 3x3 convolution on the 16-bit matrix pipes: the two-term
                fp16 split with its device-checked range by default, the three-term bf16 split with EMO_CONV_PRECISION=bf16x3)
                or, with EMO_CONV_PRECISION=f32, conv_igemm_kernel (fp32 MFMA): algorithmic FLOPs of its launches in the timed
                region / their duration measured with HIP events on the launch stream (in the default mode the pair of
                launches of a layer: the fp16-split launch and the guarded bf16x3 launch that exits when the range check
                passed), vs 2500 / 3 TF (three fp16 products per fp32 product), 2500 / 6 TF resp. the 157.3 TF fp32 MFMA
                peak; roofline_other_convs: the same for the remaining conv launches (the pointwise fp16-split kernel and the
                fp32 MFMA kernel as classes of their own; the image head, a stream kernel, with "bound": "hbm" against 8 TB/s)
  roofline_sampler  the 3-D grid_sample kernels, algorithmic bytes / event time vs 8 TB/s
  cpu_baseline  the oracle (oracle/restate.py, a port of the reference's PyTorch forward) timed on this box's host cores
                on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from emoportraits_amd import config, graphs, nets, ops, parallel, random_init  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, dense bf16 (v_mfma_f32_32x32x16_bf16); the bf16x3 kernel issues 6 products per fp32 product
PEAK_HBM_GBPS = 8000.0            # HBM3E spec; 6.29 TB/s measured copy


class ConvMeter:
    """Brackets every conv_igemm launch with HIP events on the launch stream and sums algorithmic FLOPs."""

    def __init__(self):
        self.events = []            # (start, stop, kernel = 'f32' | 'bf16x3' | 'f16', algorithmic FLOPs)
        self.flops = 0.0
        self.stream_bytes = 0.0     # algorithmic bytes of the launches of the stream kernel (csrc/conv_head.hip)
        self.launches = 0
        self._orig = None

    def __enter__(self):
        self._orig = ops.conv_igemm
        meter = self

        def wrapped(x, layer, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ret = meter._orig(x, layer, *a, **kw)
            e1.record()
            out = ret[0] if isinstance(ret, tuple) else ret         # (out, tile statistics) with want_stats=True
            positions = out.numel() // layer.cout
            fl = 2.0 * positions * layer.macs_per_position
            kern = layer.last_plan[2]
            if kern == "f16x2" and getattr(layer, "pointwise_split", False):
                kern = "f16x2_pointwise"                 # (csrc/conv_igemm_f16x2_p1.h: a kernel of its own, metered separately)
            meter.events.append((e0, e1, kern, fl))
            meter.flops += fl
            meter.launches += 1
            return ret

        ops.conv_igemm = wrapped
        nets.ops.conv_igemm = wrapped
        self._orig_head = ops.conv_head

        def wrapped_head(x, layer, *a, **kw):
            # the image head as a stream (csrc/conv_head.hip): HBM-bound, metered by its algorithmic bytes.  A launch form the
            # stream kernel does not take goes through ops.conv_igemm inside and is metered there
            layer.last_plan = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = meter._orig_head(x, layer, *a, **kw)
            e1.record()
            if layer.last_plan is not None and layer.last_plan[2] == "stream":
                positions = out.numel() // layer.cout
                meter.events.append((e0, e1, "head_stream", 2.0 * positions * layer.macs_per_position))
                meter.stream_bytes += 4.0 * positions * (layer.cin + layer.cout)
                meter.flops += 2.0 * positions * layer.macs_per_position
                meter.launches += 1
            return out

        ops.conv_head = wrapped_head
        nets.ops.conv_head = wrapped_head
        return self

    def __exit__(self, *exc):
        ops.conv_igemm = self._orig
        nets.ops.conv_igemm = self._orig
        ops.conv_head = self._orig_head
        nets.ops.conv_head = self._orig_head

    def total_ms(self):
        return sum(e[0].elapsed_time(e[1]) for e in self.events)

    def by_kernel(self):
        """{kernel: (ms, algorithmic FLOPs, launches)}"""
        out = {}
        for e0, e1, k, fl in self.events:
            ms, f, n = out.get(k, (0.0, 0.0, 0))
            out[k] = (ms + e0.elapsed_time(e1), f + fl, n + 1)
        return out


class SamplerMeter:
    def __init__(self, frames_per_step=None):
        self.frames_per_step = frames_per_step
        self.events = []
        self.bytes = 0.0
        self._orig = None

    def __enter__(self):
        self._orig = ops.grid_sample3d
        meter = self

        def wrapped(vol, grid=None, theta=None, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = meter._orig(vol, grid, theta, *a, **kw)
            e1.record()
            meter.events.append((e0, e1))
            delta = kw.get("delta")
            grid_bytes = (grid.numel() if grid is not None else 0) * 4 + (delta.numel() if delta is not None else 0) * 4
            # a volume shared by the batch is read once per STEP (amortised over the samples that share it),
            # however many launches the step's frames are split into
            shared = vol.shape[0] == 1 and bool(meter.frames_per_step) and meter.frames_per_step > 1   # also a trailing 1-frame chunk
            vol_bytes = vol.numel() * 4 * (out.shape[0] / meter.frames_per_step if shared and meter.frames_per_step else 1.0)
            meter.bytes += vol_bytes + grid_bytes + out.numel() * 4
            return out

        ops.grid_sample3d = wrapped
        nets.ops.grid_sample3d = wrapped
        return self

    def __exit__(self, *exc):
        ops.grid_sample3d = self._orig
        nets.ops.grid_sample3d = self._orig

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events)


# the statement of config.conv_arithmetic about the default mode's accuracy: the WORST mean-error ratio to the exact-fp32 MFMA
# kernel over the five shapes of tests/test_conv_bf16x3_gpu.py::FP64_SHAPES (192->128 @64^2, 512->512 @64^2 with K = 4608, the
# fused-upsample form, 128->128 @256^2, a 3-D layer), as printed into profiles/r6_parity.txt
F16X2_FP64_RATIO = ("0.49x .. 1.09x the fp32 MFMA kernel's on five layer shapes -- worst 1.09x on 192->128 @64^2 (K = 1728), 0.49x on "
                    "512->512 @64^2 (K = 4608): profiles/r6_parity.txt; test bound 1.25x")


def host_cpu():
    """(physical cores, logical cpus, model string) of this box from /proc/cpuinfo"""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return (len(cores) or logical), logical, model


def cpu_baseline(cfg, sd, inputs):
    """SURVEY.md section 8(d) 'CPU baseline timing': the oracle driver pass (oracle/restate.py -- a restatement of the
    reference's PyTorch forward that oracle/validate_restatement.py pins BIT-EXACTLY, max |delta| = 0.0 on every stage,
    against the reference's own nn.Modules; /root/reference does not exist on the GPU box) on the physical host cores,
    batch 1 per call as the reference does, 1 warm-up + 5 timed frames per thread count (bounded sample), median.  Plus the
    1-thread figure of the 3-D sampler alone: ATen's CPU grid_sampler_3d (what the reference calls, va.py:264-265) does
    not parallelise at N = 1."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import restate as O
    import torch.nn.functional as F
    physical, logical, model = host_cpu()

    def frames_at(threads, n_timed):
        torch.set_num_threads(threads)
        ts = []
        for i in range(n_timed + 1):
            t0 = time.time()
            O.driver_pass(sd, cfg, inputs["canonical"], inputs["idt"], inputs["pose"][i % inputs["pose"].shape[0]][None],
                          inputs["theta"][i % inputs["theta"].shape[0]][None])
            ts.append(time.time() - t0)
        ts = sorted(ts[1:])                     # first call = warm-up
        return ts[len(ts) // 2]

    with torch.no_grad():
        # all physical cores as SURVEY.md 8(d) specifies -- and half of them: on a 2-socket box torch's intra-op pool gets
        # SLOWER past one socket (measured 4.5 s vs 2.2 s per frame); the better one is reported as the baseline
        per_threads = {physical: frames_at(physical, 5)}
        if physical >= 16:
            per_threads[physical // 2] = frames_at(physical // 2, 5)
        threads_used = min(per_threads, key=per_threads.get)
        med = per_threads[threads_used]
        # the sampler alone, 1 thread, the reference's call at its own batch size (N = 1, explicit grid)
        torch.set_num_threads(1)
        c, d, s_ = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
        g = torch.Generator().manual_seed(2)
        lin = lambda n: torch.linspace(-1, 1, n)
        w, v, u = torch.meshgrid(lin(d), lin(s_), lin(s_), indexing="ij")
        grid = (torch.stack([u, v, w], -1)[None] + 0.05 * torch.tanh(torch.randn(1, d, s_, s_, 3, generator=g))).contiguous()
        vol = inputs["canonical"]
        F.grid_sample(vol, grid, padding_mode=cfg["grid_sample_padding_mode"], align_corners=False)
        ts = []
        for _ in range(3):
            t0 = time.time()
            F.grid_sample(vol, grid, padding_mode=cfg["grid_sample_padding_mode"], align_corners=False)
            ts.append(time.time() - t0)
        samp_ms = sorted(ts)[1] * 1e3
        torch.set_num_threads(threads_used)
    ref = None
    try:
        vj = json.load(open(os.path.join(ROOT, "oracle", f"VALIDATION_{cfg['image_size']}.json")))
        ref = {"reference_driver_s_per_frame": round(vj["reference_driver_s"], 3), "threads": vj["threads"],
               "where": "the reference's own nn.Modules timed in the build container by oracle/validate_restatement.py "
                        "(different host CPU; the restatement matched them bit for bit there)"}
    except Exception:
        pass
    return dict(value=round(1.0 / med, 4), unit="frames/s", cores=threads_used, kind="port",
                physical_cores=physical, logical_cpus=logical, cpu_model=model, threads=threads_used,
                s_per_frame_by_threads={str(k): round(v, 4) for k, v in per_threads.items()},
                sample=f"5 driver frames at {cfg['image_size']}x{cfg['image_size']} per thread count, batch 1 per call "
                       f"(1 warm-up call excluded), median; oracle/restate.py = bit-exact restatement of the reference "
                       f"PyTorch forward, torch CPU fp32; torch.set_num_threads(all {physical} physical cores) and "
                       f"({physical // 2}), the faster one is the value",
                s_per_frame=round(med, 4),
                sampler_1thread={"ms_per_call": round(samp_ms, 1), "threads": 1,
                                 "what": f"F.grid_sample (ATen CPU grid_sampler_3d, the reference's call) on "
                                         f"[1,{c},{d},{s_},{s_}] with an explicit [1,{d},{s_},{s_},3] grid, median of 3",
                                 "GBps_algorithmic": round((2 * vol.numel() + grid.numel()) * 4 / (samp_ms * 1e-3) / 1e9, 3)},
                reference_classes=ref)


def _time_loop(fn, seconds=2.0, min_iters=3, max_iters=200, graph=True):
    """median wall time of fn() (device-synchronised), bounded to about `seconds`; fn (device tensors in its closure, a device
    tensor out) is replayed from a hipGraph when it can be captured -- no launch gaps, as in the timed region of the headline"""
    fn()
    torch.cuda.synchronize()
    if graph:
        try:
            dummy = torch.zeros(1, device="cuda")
            eager = fn
            gf = graphs.Graphed(lambda d: eager(), warmup=1, clone_outputs=False)
            gf(dummy)
            fn = lambda: gf(dummy)
        except Exception:
            torch.cuda.synchronize()
    ts, t_end = [], time.perf_counter() + seconds
    while len(ts) < min_iters or (time.perf_counter() < t_end and len(ts) < max_iters):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def extras(cfg, sd, hp, ccl, idt, pose, srt, dev, S):
    """Secondary figures under the driver's clock (rank 0, N = 1, a few seconds each; the headline value is measured before):
    batch-1 latency, R256, stage 2 (BASELINE configs[4]) in exact-fp32 and fp16-operand mode, stage 1 + stage 2, the
    frames-in -> frames-out pipeline (embedders + hot path + uint8 packing) and the reference's emotion-driver hooks
    (forward(custome_target_pose_embed=, custome_target_theta_embed=), notebooks/infer.py:565-566,603-604)."""
    import tempfile
    from emoportraits_amd import embedders as E, stage2
    from emoportraits_amd.infer import InferenceWrapper
    out = {}
    B = pose.shape[0]
    theta = ops.pose_theta(*srt)
    # batch-1 latency of the hot path (what one reference-style forward() call costs on the device)
    t = _time_loop(lambda: ops.pack_rgb8(hp.driver_pass(ccl, idt, pose[:1], theta[:1])))
    out["latency_b1_ms"] = round(t * 1e3, 3)
    # the same step in the other fp32 conv modes (nets.HotPath precision=): exact-fp32 MFMA everywhere, the exact three-term bf16
    # split (round 3's default), the device-checked two-term fp16 split (the default), and the latter WITHOUT its range check and
    # guarded recomputation launches (what the contract costs)
    for mode, key in (("f32", "fp32_mfma_everywhere_fps"), ("bf16x3", "bf16x3_split_fps"), ("f16x2", "f16x2_split_fps")):
        if mode == hp.precision:
            continue
        hpm = nets.HotPath(sd, cfg, dev, with_source=False, precision=mode)
        t = _time_loop(lambda: ops.pack_rgb8(hpm.driver_pass(ccl, idt, pose, theta)))
        out[key] = round(B / t, 2)
        del hpm
    if hp.precision == "f16x2":
        ops.F16X2_GUARD = False
        try:
            t = _time_loop(lambda: ops.pack_rgb8(hp.driver_pass(ccl, idt, pose, theta)))
            out["f16x2_without_range_check_fps"] = round(B / t, 2)
        finally:
            ops.F16X2_GUARD = True
    # stage 2 at 512x512 (notebooks/infer_s2.py:351-376), 8 frames per call
    g = torch.Generator().manual_seed(11)
    s2cfg = stage2.stage2_config(overrides=dict(output_size_s2=512))
    s2sd = stage2.random_state_dict(s2cfg, seed=0)
    img8 = torch.rand(8, 3, 512, 512, generator=g).to(dev)
    m8 = (torch.rand(8, 1, 512, 512, generator=g) > 0.1).float().to(dev)
    f8 = (torch.rand(8, 1, 512, 512, generator=g) > 0.3).float().to(dev)
    s2 = {}
    for prec in ("f32", "bf16x3", "f16x2", "f16"):
        s2[prec] = stage2.Stage2(s2sd, s2cfg, dev, precision=prec)
        t = _time_loop(lambda: s2[prec].refine(img8, m8, f8))
        out[f"stage2_{prec}_fps"] = round(8 / t, 2)
    if S == 512:
        # stage 1 + stage 2 per frame, both stages in the bench's conv mode (the key names it); and BASELINE configs[4]'s
        # reduced-precision reading (fp16 MFMA operands in both stages)
        mask = torch.ones(B, 1, S, S, device=dev)
        both = s2[hp.precision if hp.precision in s2 else "f32"]
        t = _time_loop(lambda: ops.pack_rgb8(both.refine(hp.driver_pass(ccl, idt, pose, theta), mask, mask)))
        out[f"stage1_plus_stage2_{both.precision}_fps"] = round(B / t, 2)
        hp16 = nets.HotPath(sd, cfg, dev, with_source=False, precision="f16")
        t = _time_loop(lambda: ops.pack_rgb8(s2["f16"].refine(hp16.driver_pass(ccl, idt, pose, theta), mask, mask)))
        out["stage1_plus_stage2_f16_operands_fps"] = round(B / t, 2)
        t = _time_loop(lambda: ops.pack_rgb8(hp16.driver_pass(ccl, idt, pose, theta)))
        out["stage1_f16_operands_fps"] = round(B / t, 2)
        del hp16
    del s2
    # BASELINE configs[1] / SURVEY.md section 8(d) "Config 2": 64 drivers sharing one canonical volume randn(1,96,16,64,64) (seed 1);
    # uv call: identity lattice + 0.05 * tanh(randn) (seed 2); rotation call: analytic theta, yaw / pitch / roll ~ U(-0.3, 0.3),
    # scale ~ U(0.9, 1.1), translation ~ U(-0.05, 0.05) (seed 3); BOTH padding modes.  Algorithmic bytes of section 8(d) / time.
    # And the reference-layout operator seam (va.py:264-265: Model.grid_sample(NCDHW volume, explicit grid) -> NCDHW) on the same
    # 64 samples, each with a volume of its own, as a drop-in user who only swaps that attribute gets it
    try:
        c_, d_, s_ = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
        N64 = 64
        vol1 = torch.randn(1, c_, d_, s_, s_, generator=torch.Generator().manual_seed(1)).to(dev)
        ccl64 = hp.prepare_canonical(vol1)
        delta64 = (torch.tanh(torch.randn(N64, 3, d_, s_, s_, generator=torch.Generator().manual_seed(2))) * 0.05).to(dev)
        g3 = torch.Generator().manual_seed(3)
        uni = lambda lo, hi: lo + (hi - lo) * torch.rand(N64, 3, generator=g3)
        rot64, sc64, tr64 = uni(-0.3, 0.3), uni(0.9, 1.1), uni(-0.05, 0.05)
        th64 = ops.pose_theta(sc64.to(dev), rot64.to(dev), tr64.to(dev))[:, :3].contiguous()
        al64 = torch.empty((N64, c_, d_, s_, s_), device=dev)
        vol_b = c_ * d_ * s_ * s_ * 4
        # per frame: uv call = shared volume / N + delta (3 planes) + warped out; rotation call = warped in + aligned out
        byts = N64 * (vol_b / N64 + 3 * d_ * s_ * s_ * 4 + vol_b) + N64 * 2 * vol_b
        res64 = {"frames": N64, "warp": "identity + 0.05 * tanh(randn), seed 2", "theta": "yaw/pitch/roll U(-0.3,0.3), scale U(0.9,1.1), "
                 "translation U(-0.05,0.05), seed 3", "algorithmic_bytes_per_frame": byts / N64}
        for pad in ("zeros", "reflection"):
            def sampler_pair():
                for a0 in range(0, N64, hp.sampler_chunk):
                    w_ = ops.grid_sample3d(ccl64, delta=delta64[a0:a0 + hp.sampler_chunk], padding_mode=pad, in_layout="ndhwc", out_layout="ndhwc")
                    ops.grid_sample3d(w_, theta=th64[a0:a0 + hp.sampler_chunk], padding_mode=pad, in_layout="ndhwc", out_layout="ncdhw",
                                      out=al64[a0:a0 + hp.sampler_chunk])
                return al64
            t = _time_loop(sampler_pair)
            res64[pad] = {"us_per_frame": round(t / N64 * 1e6, 2), "GBps_algorithmic": round(byts / t / 1e9, 1),
                          "frac_of_8TBps": round(byts / t / 1e9 / PEAK_HBM_GBPS, 4)}
        res64.update(res64["zeros"])          # (round-4 key layout: the zeros-padding figures at the top level)
        # the seam: explicit grids [N,16,64,64,3], per-sample NCDHW volumes, NCDHW out, 16 samples per call
        NS = 16
        lin = lambda n: torch.linspace(-1, 1, n)
        wz, vy, ux = torch.meshgrid(lin(d_), lin(s_), lin(s_), indexing="ij")
        gridS = (torch.stack([ux, vy, wz], -1)[None] + delta64[:NS].permute(0, 2, 3, 4, 1).cpu()).contiguous().to(dev)
        volS = torch.randn(NS, c_, d_, s_, s_, generator=torch.Generator().manual_seed(4)).to(dev)
        outS = torch.empty_like(volS)
        t = _time_loop(lambda: ops.grid_sample3d(volS, gridS, padding_mode="zeros", out=outS))
        seam_b = NS * (2 * vol_b + d_ * s_ * s_ * 3 * 4)
        res64["reference_layout_seam_ncdhw_in_out"] = {"samples": NS, "us_per_sample": round(t / NS * 1e6, 2),
                                                       "GBps_algorithmic": round(seam_b / t / 1e9, 1),
                                                       "frac_of_8TBps": round(seam_b / t / 1e9 / PEAK_HBM_GBPS, 4)}
        out["sampler_only_n64"] = res64
        del vol1, ccl64, delta64, al64, gridS, volS, outS
    except Exception as e:
        out["sampler_only_n64"] = {"error": repr(e)}
    # R256 (BASELINE configs[0]/[1] size): same hot path, 32 frames per step
    cfg256 = config.hot_path_config(overrides={"image_size": 256})
    sd256 = random_init.trained_like_state_dict(cfg256, seed=0, with_source=False)
    hp256 = nets.HotPath(sd256, cfg256, dev, with_source=False)
    B2 = 32
    g2 = torch.Generator().manual_seed(12)
    pose2 = torch.randn(B2, cfg256["lpe_output_channels_expression"], generator=g2).to(dev)
    th2 = ops.pose_theta(*[x.to(dev) for x in (1 + 0.05 * torch.randn(B2, 3, generator=g2), 0.3 * torch.randn(B2, 3, generator=g2),
                                               0.05 * torch.randn(B2, 3, generator=g2))])
    t = _time_loop(lambda: ops.pack_rgb8(hp256.driver_pass(ccl, idt, pose2, th2)))
    out["r256_fps"] = round(B2 / t, 2)
    del hp256
    # frames in -> frames out through the wrapper (HeadPoseRegressor + ExpressionEmbed + hot path + uint8 D2H ring) and the
    # emotion-driver hooks, on a wrapper built like the reference's (args.txt + checkpoint layout)
    ecfg = E.embedder_config()
    full = dict(random_init.trained_like_state_dict(cfg, seed=0))
    full.update(E.random_state_dict(E.idt_schema(ecfg), 1))
    full.update(E.random_state_dict(E.expression_schema(ecfg), 2))
    hp_sd = E.random_state_dict(E.head_pose_schema(), 3)
    hp_sd["fc.weight"] *= 0.05
    hp_sd["fc.bias"] = torch.tensor([1.0, 1.0, 1.0, 0.1, -0.2, 0.05, 0.02, -0.03, 0.01])
    root = tempfile.mkdtemp()
    os.makedirs(os.path.join(root, "logs", "exp", "checkpoints"))
    with open(os.path.join(root, "logs", "exp", "args.txt"), "wt") as f:
        for k, v in {**cfg, **ecfg}.items():
            f.write(f"{k}: {v}\n")
    torch.save(hp_sd, os.path.join(root, "hp.pth"))
    w = InferenceWrapper(experiment_name="exp", model_file_name="x", project_dir=root, folder="logs", state_dict=full,
                         print_params=False, head_pose_regressor_path=os.path.join(root, "hp.pth"))
    g3 = torch.Generator().manual_seed(13)
    w.forward(source_image=torch.rand(1, 3, S, S, generator=g3), crop=False, source_mask=torch.ones(1, 1, S, S))
    frames = (torch.rand(B * 6, S, S, 3, generator=g3) * 255).to(torch.uint8).pin_memory()
    for _ in w.animate_frames(frames[:B * 2], batch_size=B):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(o.shape[0] for _, o in w.animate_frames(frames, batch_size=B))
    out["pipeline_frames_in_out_fps"] = round(n / (time.perf_counter() - t0), 2)
    # emotion driver: expression vectors and (scale, rotation, translation) poses fed through the reference's hooks, one
    # forward() call per frame as the reference API has it (PIL image out)
    ge = torch.Generator().manual_seed(14)
    embeds = torch.randn(16, 1, cfg["lpe_output_channels_expression"], generator=ge).to(dev)
    poses = [(1 + 0.05 * torch.randn(1, 3, generator=ge), 0.3 * torch.randn(1, 3, generator=ge), 0.05 * torch.randn(1, 3, generator=ge))
             for _ in range(16)]
    poses = [tuple(x.to(dev) for x in p) for p in poses]

    def emo_loop():
        for e, p in zip(embeds, poses):
            w.forward(custome_target_pose_embed=e, custome_target_theta_embed=p, crop=False)
    t = _time_loop(emo_loop, seconds=3.0)
    out["emotion_driver_forward_fps"] = round(16 / t, 2)
    out["what"] = ("latency_b1_ms: one driver frame through the hot path; fp32_mfma_everywhere_fps / bf16x3_split_fps / f16x2_split_fps: "
                   "the bench step in the fp32 conv modes that are not the headline's (exact-fp32 MFMA everywhere; exact 3-term bf16 split; "
                   "device-checked 2-term fp16 split); f16x2_without_range_check_fps: the headline mode without its overflow words and "
                   "guarded recomputation launches; stage2_*: Stage2.refine at 512x512, 8 frames per call (f32 / bf16x3 / f16x2: fp32 "
                   "modes as above, f16 = reduced precision, fp16 operands, configs[4]); sampler_only_n64: BASELINE configs[1], the two "
                   "3-D grid_sample calls of the driver pass on 64 frames; "
                   "stage1_plus_stage2_<mode>_fps: driver pass + refinement + uint8 pack in the named conv mode, B frames per call; r256_fps: R256 driver pass, 32 frames "
                   "per call; pipeline_frames_in_out_fps: InferenceWrapper.animate_frames (uint8 in, embedders, hot path, uint8 out); "
                   "emotion_driver_forward_fps: forward(custome_target_pose_embed=, custome_target_theta_embed=) per frame, PIL out")
    return out


class StrongClip:
    """BASELINE configs[3] / SURVEY.md section 8(e): ONE clip of `total` driver frames of one source identity, the unit a user
    animates.  Per clip, in the timed region:
        rank 0   source pass (LocalEncoder, VPN, xy WarpGenerator, 2 samplers, Unet3D) on the clip's source image
        all      ONE flat broadcast of {canonical volume 25.2 MB, idt_embed, theta_src} (RCCL over xGMI; shapes known: no header)
        rank r   its contiguous shard [lo, hi) of DISTINCT frames (parallel.shard_range) in batches of B: pose theta, driver pass,
                 uint8 pack -- the full batches replayed from one hipGraph, a ragged tail batch from a second one -- and every
                 batch's frames D2H into a ring of pinned buffers on a copy stream (the host waits only for a slot it reuses)
    There is no data-path collective.  The frames' inputs (expression vectors, scale / rotation / translation) are resident in
    HBM before the clock starts; the frames leave as uint8 in pinned host memory."""

    def __init__(self, hp, cfg, dev, rank, world, total, B, S, seed, graph=True, ring=3):
        self.hp, self.cfg, self.dev, self.rank, self.world, self.total, self.B, self.S = hp, cfg, dev, rank, world, total, B, S
        c, d, s = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
        self.shapes = dict(canonical=(1, c, d, s, s), idt_embed=(1, cfg["gen_max_channels"], 4, 4), theta_src=(1, 4, 4))
        g = torch.Generator().manual_seed(seed + 7)
        E = cfg["lpe_output_channels_expression"]
        if rank == 0:
            self.src = dict(img=torch.rand(1, 3, S, S, generator=g).to(dev), idt=torch.randn(1, cfg["gen_max_channels"], 4, 4, generator=g).to(dev),
                            pose=torch.randn(1, E, generator=g).to(dev))
            self.src["theta"] = ops.pose_theta(*[t.to(dev) for t in (1 + 0.05 * torch.randn(1, 3, generator=g),
                                                                     0.3 * torch.randn(1, 3, generator=g), 0.05 * torch.randn(1, 3, generator=g))])
        # every rank draws the WHOLE clip's inputs from the same seed and keeps its shard: distinct frames, no exchange
        gf = torch.Generator().manual_seed(seed + 1000)
        pose = torch.randn(total, E, generator=gf)
        srt = (1 + 0.05 * torch.randn(total, 3, generator=gf), 0.3 * torch.randn(total, 3, generator=gf), 0.05 * torch.randn(total, 3, generator=gf))
        self.lo, self.hi = parallel.shard_range(total, rank, world)
        self.pose = pose[self.lo:self.hi].to(dev)
        self.srt = [t[self.lo:self.hi].to(dev) for t in srt]
        # static per-identity buffers the (graph-captured) step reads: overwritten by every clip's broadcast
        self.ccl = torch.zeros((1, d, s, s, c), device=dev)
        self.idt = torch.zeros(self.shapes["idt_embed"], device=dev)
        self.ring = [torch.empty((B, S, S, 3), dtype=torch.uint8).pin_memory() for _ in range(ring)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.checksum = 0

        def step_fn(pose_, s0, s1, s2):
            return ops.pack_rgb8(hp.driver_pass(self.ccl, self.idt, pose_, ops.pose_theta(s0, s1, s2)))
        self.step = graphs.Graphed(step_fn, warmup=1, clone_outputs=False) if graph else step_fn
        # (the source pass is ~300 dependent launches for 13 ms of GPU time: replayed from a graph as well)
        self.source = (graphs.Graphed(lambda *t: hp.source_pass(*t), warmup=1, clone_outputs=False) if graph else hp.source_pass) \
            if rank == 0 else None

    def run(self):
        """one clip; returns the number of frames this rank delivered to the host"""
        hp, dev = self.hp, self.dev
        cache = {"canonical": None, "idt_embed": None, "theta_src": None}
        if self.rank == 0:
            canonical = self.source(self.src["img"], self.src["idt"], self.src["pose"], self.src["theta"])
            cache = {"canonical": canonical, "idt_embed": self.src["idt"], "theta_src": self.src["theta"]}
        cache = parallel.broadcast_source_cache(cache, self.shapes, src=0, device=dev, world=self.world, rank=self.rank,
                                                names=list(self.shapes), exchange_shapes=False)
        self.ccl.copy_(hp.prepare_canonical(cache["canonical"]))
        self.idt.copy_(cache["idt_embed"])
        events = [None] * len(self.ring)
        done = 0
        n = self.hi - self.lo
        for k, b0 in enumerate(range(0, n, self.B)):
            b1 = min(n, b0 + self.B)
            u8 = self.step(self.pose[b0:b1], *[t[b0:b1] for t in self.srt]).clone()     # (the graph's static output is rewritten by the next replay)
            ready = torch.cuda.Event()
            ready.record()
            slot = k % len(self.ring)
            if events[slot] is not None:
                events[slot].synchronize()               # the host has "consumed" the slot's previous batch
                self.checksum += int(self.ring[slot][0, 0, 0, 0])
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ready)
                self.ring[slot][:b1 - b0].copy_(u8, non_blocking=True)
                u8.record_stream(self.copy_stream)
                ev = torch.cuda.Event()
                ev.record()
            events[slot] = ev
            done += b1 - b0
        for ev in events:
            if ev is not None:
                ev.synchronize()
        return done


def strong_scaling(hp, cfg, dev, rank, world, total, B, S, seed, clips, warm, graph):
    """-> record of the strong-scaling measurement (rank 0) or None: `clips` timed clips behind `warm` untimed ones, barrier +
    device synchronisation on both sides, MAX over ranks"""
    clip = StrongClip(hp, cfg, dev, rank, world, total, B, S, seed, graph=graph)
    for _ in range(warm):
        clip.run()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    delivered = 0
    for _ in range(clips):
        delivered += clip.run()
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device=dev)
    frames_all = parallel.sum_over_ranks(delivered, device=dev)
    if frames_all != total * clips:
        raise SystemExit(f"strong scaling: the ranks delivered {frames_all} frames, the clips hold {total * clips}")
    if rank != 0:
        return None
    return {"scaling": "strong", "total_frames": total, "clips": clips, "warmup_clips": warm, "frames_per_s": round(total * clips / elapsed, 3),
            "ms_per_clip": round(elapsed / clips * 1e3, 3), "n_gpus": world, "frames_per_rank": [parallel.shard_range(total, r, world)[1] -
                                                                                               parallel.shard_range(total, r, world)[0] for r in range(world)],
            "batch": B, "in_timed_region": "rank 0's source pass, one flat broadcast of its cache, every rank's contiguous shard of distinct "
                                           "frames (pose theta, driver pass, uint8 pack; hipGraph replay per batch), uint8 D2H into a pinned ring"}


def sustained_mfma(dev, seconds=1.2):
    """What the POWER-MANAGED chip sustains on a kernel that does nothing but fp16 MFMAs (emo_mfma_stream_f16: one wave per SIMD on
    every CU, v_mfma_f32_32x32x16_f16 back to back on pseudo-random operands), measured behind the timed region on the same
    box, about `seconds` of GPU time per form after a warm-up of the same length (the clock settles within ~0.3 s).  The data
    sheet's 2.5 PF assumes 2.4 GHz; under a matrix stream this part runs 1.5-1.7 GHz.  A reader can then tell kernel quality
    (roofline.frac_of_sustained) from clock (roofline.frac)."""
    sink = torch.empty(256 * ops.device_cu_count(), device=dev, dtype=torch.float32)
    out = {}
    for key, lds in (("bare", False), ("with_fragment_reads", True)):
        iters = 4000
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_mfma = ops.mfma_stream(iters, lds, sink)
        torch.cuda.synchronize()
        per_launch = max(1e-4, time.perf_counter() - t0)
        reps = max(2, int(seconds / per_launch))
        for _ in range(reps):                                   # warm-up: the clock under THIS load
            ops.mfma_stream(iters, lds, sink)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.mfma_stream(iters, lds, sink)
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3
        tf = n_mfma * reps * 32768 / sec / 1e12
        out[key] = {"fp16_tflops": round(tf, 1), "launches": reps, "seconds": round(sec, 3),
                    # 1024 lanes-wide MFMA throughput per CU and clock: 4 SIMDs x 32768 flop / 32 cycles
                    "implied_clock_ghz": round(tf * 1e12 / (ops.device_cu_count() * 4 * 32768 / 32.0) / 1e9, 3)}
    out["what"] = ("emo_mfma_stream_f16 (csrc/api.hip): bare v_mfma_f32_32x32x16_f16 stream, one wave per SIMD, all CUs, pseudo-random "
                   "operands; with_fragment_reads adds 8 ds_read_b128 per 12 MFMAs (a step of the split kernels' K loop); measured in "
                   "this run, behind the timed region")
    return out


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` with N > 1 and no rendezvous environment: start N ranks ourselves (one process per GPU,
    RCCL backend) exactly as the driver would -- python -m torch.distributed.run on 127.0.0.1 -- and relay its output."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)                                # (HSA_ENABLE_IPC_MODE_LEGACY=0 is defaulted by emoportraits_amd.parallel)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=16, help="driver frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-source-pass", action="store_true",
                    help="profiling aid: use a synthetic canonical volume instead of running the source pass, so that "
                         "a rocprofv3 trace of this command contains driver-pass launches only")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--raw-weights", action="store_true", help="plain seeded initialisation instead of the trained-like checkpoint")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (extras)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the sustained-MFMA-rate measurement (roofline.sustained_peak)")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of a hipGraph replay of the step")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="STRONG scaling (BASELINE configs[3]): a step is one whole clip of this many driver frames of one identity, "
                         "sharded contiguously over the ranks -- rank 0's source pass, the broadcast of its cache, every rank's shard "
                         "in batches of --batch and the uint8 D2H ring all inside the timed region; the line says scaling: strong")
    ap.add_argument("--strong-frames", type=int, default=512,
                    help="weak-scaling runs also report one strong-scaling figure (strong_scaling in the line) on a clip of this "
                         "many frames; 0 skips it")
    a = ap.parse_args()

    if a.total_frames and a.no_source_pass:          # (every rank sees the same flags: no rank is left waiting in a collective)
        raise SystemExit("--total-frames times the source pass of every clip: it cannot be combined with --no-source-pass")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        visible = torch.cuda.device_count()
        if visible < a.gpus and os.environ.get("EMO_FORCE_DEVICE") is None:
            raise SystemExit(f"--gpus {a.gpus} but only {visible} GPU(s) are visible (EMO_FORCE_DEVICE=0 + "
                             f"EMO_DIST_BACKEND=gloo shares one GPU between ranks: a functional test, not a measurement)")
        relaunch_under_torchrun(a.gpus)
    rank, world = parallel.init_distributed()
    if world != max(1, a.gpus):
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus {a.gpus} does it itself when no WORLD_SIZE is set)")
    dev = torch.device("cuda", parallel.local_device_index())
    torch.cuda.set_device(dev)

    cfg = config.hot_path_config(overrides={"image_size": a.image_size})
    # seeded "trained-like" checkpoint (random_init.trained_like_state_dict): spectral norms ~1, predicted warps within one
    # voxel of the identity, unsaturated image -- the statistics of a trained model, which the released one (not obtainable
    # here) has; --raw-weights times the plain seeded initialisation of rounds 1-2 (uv warp of +-30 voxels) instead
    sd = random_init.random_state_dict(cfg, seed=a.seed) if a.raw_weights else random_init.trained_like_state_dict(cfg, seed=a.seed)
    hp = nets.HotPath(sd, cfg, dev, with_source=(rank == 0 and not a.no_source_pass))
    c, d, s = cfg["latent_volume_channels"], cfg["latent_volume_depth"], cfg["latent_volume_size"]
    S, B = a.image_size, a.batch

    # ---- per identity: source pass on rank 0 (synthetic masked image + embeddings), RCCL broadcast ----
    g = torch.Generator().manual_seed(a.seed + 1)
    idt_cpu = torch.randn(1, cfg["gen_max_channels"], 4, 4, generator=g)
    cache = {"canonical": None, "idt_embed": None, "theta_src": None}
    source_ms = None
    if rank == 0 and a.no_source_pass:
        cache = {"canonical": torch.randn(1, c, d, s, s, generator=g).to(dev), "idt_embed": idt_cpu.to(dev),
                 "theta_src": torch.eye(4)[None].to(dev)}
    elif rank == 0:
        img = torch.rand(1, 3, S, S, generator=g).to(dev)
        pose_s = torch.randn(1, cfg["lpe_output_channels_expression"], generator=g).to(dev)
        srt_s = [t.to(dev) for t in (1 + 0.05 * torch.randn(1, 3, generator=g), 0.3 * torch.randn(1, 3, generator=g),
                                    0.05 * torch.randn(1, 3, generator=g))]
        th_s = ops.pose_theta(*srt_s)
        for _ in range(2):                                          # warm-up: lazy weight packing and allocator (first call), and a
            hp.source_pass(img, idt_cpu.to(dev), pose_s, th_s)      # GPU that is awake (the first call is mostly host work)
        torch.cuda.synchronize()
        t0 = time.time()
        canonical = hp.source_pass(img, idt_cpu.to(dev), pose_s, th_s)
        torch.cuda.synchronize()
        source_ms = (time.time() - t0) * 1e3
        cache = {"canonical": canonical, "idt_embed": idt_cpu.to(dev), "theta_src": th_s}
    shapes = dict(canonical=(1, c, d, s, s), idt_embed=(1, cfg["gen_max_channels"], 4, 4), theta_src=(1, 4, 4))
    names = list(shapes)
    # every rank knows the shapes: one flat RCCL broadcast, no header exchange, no host synchronisation
    bc = lambda: parallel.broadcast_source_cache(cache, shapes, src=0, device=dev, world=world, rank=rank, names=names,
                                                 exchange_shapes=False)
    broadcast_ms = None
    if world > 1:
        bc()                               # first collective: communicator set-up, not the steady-state cost
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out_cache = bc()
        torch.cuda.synchronize()
        broadcast_ms = parallel.max_over_ranks(time.perf_counter() - t0, device=dev) * 1e3
        cache = out_cache
    else:
        cache = bc()
    ccl = hp.prepare_canonical(cache["canonical"])
    idt = cache["idt_embed"]

    # ---- this rank's shard of synthetic driver frames (weak scaling: B per GPU per step) ----
    gd = torch.Generator().manual_seed(a.seed + 100 + rank)
    pose = torch.randn(B, cfg["lpe_output_channels_expression"], generator=gd).to(dev)
    srt = [t.to(dev) for t in (1 + 0.05 * torch.randn(B, 3, generator=gd), 0.3 * torch.randn(B, 3, generator=gd),
                              0.05 * torch.randn(B, 3, generator=gd))]

    def step_fn(pose_, s0, s1, s2):
        theta = ops.pose_theta(s0, s1, s2)                                  # a3
        img = hp.driver_pass(ccl, idt, pose_, theta)                        # a4, a5, a1 x2 (+a2), a9
        return ops.pack_rgb8(img)                                           # a11 (device-side uint8 packing)

    # The step is ~150 launches.  Launched eagerly they leave the GPU idle for 2-6 ms of an 85 ms step (gaps between dependent
    # launches; measured 175.8-183.8 frames/s eager against 190.4-192.7 replayed, same kernels, same 84.8 ms of kernel time), so
    # the timed region replays the step from a hipGraph (one launch per step; graphs.Graphed, the same mechanism as
    # InferenceWrapper(use_graphs=True)); --no-graph times the eager launches instead.
    step_launch = "eager"
    step = lambda: step_fn(pose, *srt)
    if not a.no_graph:
        try:
            gstep = graphs.Graphed(step_fn, warmup=1, clone_outputs=False)
            gstep(pose, *srt)
            step = lambda: gstep(pose, *srt)
            step_launch = "hipgraph"
        except Exception as e:                                              # capture is an optimisation, never a requirement
            sys.stderr.write(f"bench: hipGraph capture failed ({e!r}); timing eager launches\n")
            torch.cuda.synchronize()

    strong = None
    if a.total_frames:
        # STRONG scaling is the headline of this invocation: a step = one clip (class StrongClip), K = --steps clips behind W =
        # --warmup untimed ones; the weak-scaling step below is still run (briefly) for the per-kernel figures of the line
        strong = strong_scaling(hp, cfg, dev, rank, world, a.total_frames, B, S, a.seed, a.steps, a.warmup, not a.no_graph)
    for _ in range(a.warmup):
        step()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device=dev)
    strong_extra = None
    if not a.total_frames and a.strong_frames and not a.no_source_pass:
        # one strong-scaling figure beside the weak headline (2 clips behind 1 untimed), so that a plain `--gpus N` sweep also
        # yields the fixed-total-work curve of BASELINE configs[3]
        strong_extra = strong_scaling(hp, cfg, dev, rank, world, a.strong_frames, B, S, a.seed, 2, 1, not a.no_graph)

    # the same K steps once more, launched eagerly with HIP events around every conv / sampler launch: the per-kernel figures.
    # An event interval is a kernel's duration only while the host is AHEAD of the GPU (otherwise it contains the wait for the
    # launch -- and the first eager step after the capture allocates its intermediates afresh: the warm-up ran on the capture's
    # side stream, whose cached blocks the launch stream cannot reuse), so each metered step is enqueued behind graph replays
    # of the step: real work during which the host enqueues the eager launches, at the clocks of a sustained run.
    # How many replays the host needs as head start depends on the HOST (measured on one box: the metered step took 65 ms of GPU
    # time against 43.5 replayed -- event intervals with launch gaps in them, every per-kernel figure a third too slow -- where
    # two replays had been enough on every box before).  So: the host-side enqueue time of one metered step is measured first,
    # the number of blockers follows from it, and the pass is repeated with twice as many while its GPU time per step still
    # exceeds the replayed step's by more than 6 % (the meters' own event records cost about 1 %).
    def metered_pass(blockers):
        cm, sm = ConvMeter(), SamplerMeter(frames_per_step=B)
        torch.cuda.synchronize()
        sp = []
        for _ in range(a.steps):
            if step_launch == "hipgraph":
                for _b in range(blockers):
                    step()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            with cm, sm:
                step_fn(pose, *srt)
            s1.record()
            sp.append((s0, s1))
        torch.cuda.synchronize()
        return cm, sm, sum(x.elapsed_time(y) for x, y in sp) * 1e-3

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with ConvMeter(), SamplerMeter(frames_per_step=B):
        step_fn(pose, *srt)                                   # (GPU idle: the wall time of this call is the host's enqueue time)
    host_enqueue_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    step_ms = elapsed / a.steps * 1e3
    blockers = max(2, int(1.5 * host_enqueue_ms / step_ms) + 1)
    while True:
        conv_meter, samp_meter, elapsed_metered = metered_pass(blockers)
        if step_launch != "hipgraph" or elapsed_metered / a.steps * 1e3 <= 1.06 * step_ms or blockers >= 16:
            break
        blockers *= 2

    # fp16-split layers whose device-side range check fired during the last step (their guarded bf16x3 launch then recomputed
    # them: correct, but the step was not the fp16 split's): none on the bench checkpoint
    recomputed = sorted(v for v in hp.overflow_events().values() if v) if hp.precision == "f16x2" else []
    # host cores of every rank (parallel.pin_to_local_cores: next to the rank's GPU), for the record
    affinity = [parallel.affinity_record()]
    if world > 1:
        try:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, affinity[0])
            affinity = gathered
        except Exception as e:                                    # a record, never a reason to lose the line
            affinity = [affinity[0], f"all_gather_object failed: {e!r}"]
    if rank != 0:
        return
    sustained = None
    if not a.no_sustained:
        try:
            sustained = sustained_mfma(dev)
        except Exception as e:                                    # a diagnostic must never cost the headline line
            sustained = {"error": repr(e)}
    frames = world * B * a.steps
    fps = frames / elapsed
    weak = {"scaling": "weak", "frames_per_s": round(fps, 3), "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "frames_per_gpu_per_step": B, "what": "the same step with the per-GPU batch fixed and the inputs replayed (source pass and "
                                                  "broadcast outside the timed region)"}
    if strong is not None:
        fps, elapsed = strong["frames_per_s"], strong["ms_per_clip"] * 1e-3 * a.steps
    conv_ms = conv_meter.total_ms()
    samp_ms = samp_meter.total_ms()
    conv_tflops = conv_meter.flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    samp_gbps = samp_meter.bytes / (samp_ms * 1e-3) / 1e9 if samp_ms > 0 else 0.0
    by_k = conv_meter.by_kernel()
    dom = max(by_k, key=lambda k: by_k[k][0])                   # the kernel the step spends most of its time in

    def conv_roofline(k):
        ms, fl, n = by_k[k]
        def pmc_traffic(pat):
            """HBM bytes per launch from a counter profile of this command (separate rocprofv3 --pmc passes: it cannot be measured
            inside the run) -- quoted only from a profile taken on THESE kernel sources (the file records their hash,
            tools/kernel_source_hash.py); with another hash, or none, the field is null"""
            found = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", pat)))[-1:]
            if not found:
                return None, "", None
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from kernel_source_hash import kernel_source_hash
                pj = json.load(open(found[0]))
                if pj.get("kernel_source_sha16") == kernel_source_hash():
                    return pj.get("hbm_bytes_per_launch"), found[0], None
                return None, "", os.path.relpath(found[0], ROOT)
            except Exception:
                return None, "", None

        def traffic_source(pmc, pmc_path, stale):
            return (f"{os.path.relpath(pmc_path, ROOT)}: rocprofv3 --pmc passes of this command (guide-corrected HBM bytes per launch) on "
                    "these kernel sources, NOT measured in this run" if pmc is not None else
                    (f"{stale} was taken on other kernel sources: not quoted" if stale else None))

        if k == "head_stream":
            gbps = conv_meter.stream_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            pmc, pmc_path, stale = pmc_traffic("r*_pmc_conv_head_traffic.json")
            return {"bound": "hbm", "kernel": "conv_head_kernel<3> (the image head 128 -> 3, 1x1, GroupNorm affine + ReLU on the way in, "
                                              "sigmoid on the way out, as a stream: no LDS, fp32 FMAs)",
                    "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                    "traffic": pmc, "traffic_source": traffic_source(pmc, pmc_path, stale),
                    "launches_per_step": n // max(1, a.steps), "avg_launch_ms": round(ms / max(1, n), 4),
                    "share_of_step": round(ms / (elapsed_metered * 1e3), 3),
                    "note": "achieved = 4 bytes x (Cin + Cout) x positions / event time"}
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        pmc, pmc_path, stale = pmc_traffic({"bf16x3": "r*_pmc_conv_bf16x3_traffic.json", "f16x2": "r*_pmc_conv_f16x2_traffic.json",
                                            "f16x2_pointwise": "r*_pmc_conv_f16x2_p1_traffic.json"}.get(k, "r*_pmc_conv_traffic.json"))
        if k == "f16x2":
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
            name = ("conv_igemm_bf16x3_ct2_kernel + conv_igemm_bf16x3_kernel<SPLIT = 2> (fp32 3x3 conv on the fp16 matrix pipes: scaled "
                    "operands as two fp16 terms, 3 v_mfma_f32_32x32x16_f16 products per fp32 product, fp32 accumulation; two 64-channel "
                    "output tiles per work item on one converted patch where a layer has pairs of them, the single-tile kernel on an "
                    "odd last tile; operand range checked on the device, guarded bf16x3 recomputation launch behind every layer)")
            note = ("achieved = algorithmic fp32 FLOPs / event time of a LAYER's launches (two-tile launch, single-tile launch of an odd "
                    "last tile, the guarded, normally skipped, bf16x3 launch); peak = 2500 TF dense fp16 / 3 products")
        elif k == "f16x2_pointwise":
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
            name = ("conv_igemm_bf16x3_p1_kernel (1x1 convolutions on the fp16 matrix pipes: the two-term split, 32-channel stages, two "
                    "channel tiles per work item; guarded fp32 MFMA recomputation launch behind every layer)")
            note = "achieved = algorithmic fp32 FLOPs / event time of the launch pair; peak = 2500 TF dense fp16 / 3 products"
        elif k == "bf16x3":
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0
            name = ("conv_igemm_bf16x3_kernel (fp32 3x3 conv on the bf16 matrix pipes: exact 3-way operand split, 6 "
                    "v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulation)")
            note = ("achieved = algorithmic fp32 FLOPs / event time; peak = 2500 TF dense bf16 / 6 products.  A bare stream of this "
                    "MFMA sustains 1.49-1.72 PF on this chip (clocks fall to 1.4-1.7 GHz under it: tools/microbench/mfma_stream.hip, "
                    "archive/profiles/r3_mfma_stream.jsonl) = 249-287 TF fp32-equivalent")
        elif k == "f16w8":
            peak = PEAK_BF16_MFMA_TFLOPS
            name = ("conv_igemm_f16x2_w8_kernel<NPROD = 1> (plain fp16 operands, fp32 accumulation: two 64-channel tiles per work item, "
                    "eight waves -- two per SIMD; opt-in precision 'f16', BASELINE configs[4])")
            note = "achieved = algorithmic FLOPs / event time; peak = 2500 TF dense fp16"
        else:
            peak = PEAK_FP32_MFMA_TFLOPS if k == "f32" else PEAK_BF16_MFMA_TFLOPS
            name = ("conv_igemm_kernel (fp32 32x32x2 MFMA implicit-GEMM conv, all instantiations)" if k == "f32" else
                    "conv_igemm_f16_kernel (fp16 operands)")
            note = None
        r = {"bound": "mfma", "kernel": name, "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
             "frac": round(tf / peak, 4), "traffic": pmc, "traffic_source": traffic_source(pmc, pmc_path, stale),
             "launches_per_step": n // max(1, a.steps), "avg_launch_ms": round(ms / max(1, n), 4),
             "share_of_step": round(ms / (elapsed_metered * 1e3), 3)}
        if note:
            r["note"] = note
        products = {"f16x2": 3, "f16x2_pointwise": 3, "bf16x3": 6, "f16": 1, "f16w8": 1}.get(k)
        if products and sustained and "bare" in sustained:
            # the same fraction against what the chip SUSTAINS on a pure matrix stream at its power-managed clock (measured in this
            # run): kernel quality apart from clock.  `frac` stays the fraction of the data-sheet peak
            sp = sustained["bare"]["fp16_tflops"] / products
            r["sustained_peak"] = round(sp, 1)
            r["frac_of_sustained"] = round(tf / sp, 4) if sp > 0 else None
            r["sustained_peak_note"] = (f"bare fp16 MFMA stream measured behind the timed region: {sustained['bare']['fp16_tflops']} TF "
                                        f"(implied clock {sustained['bare']['implied_clock_ghz']} GHz) / {products} products")
        return r

    rec = {
        "metric": "reenactment frames/sec @512x512, 1-src->N-driver" if S == 512 else f"reenactment frames/sec @{S}x{S}, 1-src->N-driver",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong is not None else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "dtype_note": f"tensors, norms, epilogue and accumulation fp32; conv operand form: {hp.precision} (config.conv_arithmetic)",
        "config": {"workload": (f"released stage-1 architecture R{S}, full HIP driver pass (pose theta, warp embed, uv WarpGenerator, "
                                f"2x 3-D grid_sample, decoder, uint8 pack), 1 source identity, {B} driver frames per GPU per step"
                                if strong is None else
                                f"released stage-1 architecture R{S}: one clip of {a.total_frames} driver frames of one source identity per "
                                f"step, sharded contiguously over {world} rank(s) -- source pass on rank 0, one broadcast of its cache, "
                                f"full HIP driver pass in batches of {B}, uint8 frames D2H into a pinned ring, all inside the timed region"),
                   "image_size": S, "frames_per_gpu_per_step": B,
                   "weights": ("seeded random, reference key layout" if a.raw_weights else
                               "seeded trained-like (spectral norms ~1, |uv delta| < ~1 voxel, unsaturated image), reference key layout"),
                   "conv_arithmetic": {"bf16x3": "fp32 tensors and accumulation; 3x3 / 3x3x3 convs of the decoder and the WarpGenerator that a 256-position tile of the split kernel fits: every fp32 operand split exactly "
                                                 "into 3 bf16 terms, 6 partial products on the bf16 matrix pipes (error vs fp64 <= "
                                                 "the fp32 MFMA kernel's, tests/test_conv_bf16x3_gpu.py); other convs: fp32 MFMA",
                                       "f16x2": "fp32 tensors and accumulation; 3x3 / 3x3x3 convs of the decoder and the WarpGenerator that a 256-position tile of the split kernel fits: every scaled fp32 operand as two "
                                                "fp16 terms (2^-24 relative), 3 partial products on the fp16 matrix pipes (mean error vs fp64 "
                                                + F16X2_FP64_RATIO + "); the operand range is checked on the device by every "
                                                "launch and a guarded bf16x3 launch recomputes a layer that left it "
                                                "(tests/test_conv_bf16x3_gpu.py); the decoder's four 1x1 layers: the same split on "
                                                "the pointwise kernel (guarded fp32 MFMA recomputation); other convs: fp32 MFMA",
                                       "f32": "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) in every convolution"}[hp.precision],
                   "step_launch": step_launch,
                   "f16x2_layers_recomputed_after_range_check": recomputed,
                   "metered_pass": (f"roofline figures: the same {a.steps} steps launched eagerly with HIP events around every conv / "
                                    f"sampler launch, each behind {blockers} graph replays so that the host is ahead of the GPU (host "
                                    f"enqueue time of a metered step: {host_enqueue_ms:.1f} ms); "
                                    f"{elapsed_metered / a.steps * 1e3:.2f} ms of GPU time per metered step"),
                   "parallelism": f"frame-parallel x{world}" if world > 1 else "single GPU"},
        "roofline": conv_roofline(dom),
        "roofline_sampler": {"bound": "hbm", "kernel": "gs3d_cl_v2 / gs3d_cl2ncdhw_v2 (3-D grid_sample, channels-last)",
                             "achieved": round(samp_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": round(samp_gbps / PEAK_HBM_GBPS, 4),
                             "avg_launch_ms": round(samp_ms / max(1, len(samp_meter.events)), 4)},
        "roofline_other_convs": {k: conv_roofline(k) for k in by_k if k != dom},
        "sustained_mfma": sustained,
        "host_affinity": affinity,
        "strong_scaling": strong if strong is not None else strong_extra,
        "weak_scaling": weak if strong is not None else None,
        "source_pass_ms": None if source_ms is None else round(source_ms, 2),
        "broadcast_ms": None if broadcast_ms is None else round(broadcast_ms, 3),   # one flat RCCL broadcast of the source cache, max over ranks
    }
    if world == 1 and not a.no_extras:
        try:
            rec["extras"] = extras(cfg, sd, hp, ccl, idt, pose, srt, dev, S)
        except Exception as e:                                    # secondary figures must never cost the headline line
            rec["extras"] = {"error": repr(e)}
    if world == 1 and not a.no_cpu_baseline:
        inputs = dict(canonical=cache["canonical"].cpu(), idt=idt_cpu, pose=pose.cpu(), theta=ops.pose_theta(*srt).cpu())
        rec["cpu_baseline"] = cpu_baseline(cfg, sd, inputs)
    else:
        rec["cpu_baseline"] = None
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        parallel.shutdown()
