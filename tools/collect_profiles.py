"""Copies the summaries of the last GPU run from gpurun_out/ (scratch) into profiles/ (tracked) and derives
profiles/<tag>_pmc_conv_traffic.json = per-launch HBM traffic + MFMA utilisation of the dominant kernel."""
import csv
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
g, p = "gpurun_out/", "profiles/"
shutil.copy(g + f"{tag}_kernel_stats.csv", p + f"{tag}_bench_kernel_stats.csv")
import os
for src, dst in ((f"{tag}_bench.json", f"{tag}_bench_n1.json"), (f"{tag}_bench256.json", f"{tag}_bench_n1_r256.json"),
                 (f"{tag}_bench_2ranks_1gpu.json", f"{tag}_bench_2ranks_on_1gpu_gloo.json"),
                 (f"{tag}_stage2.jsonl", f"{tag}_stage2_bench.jsonl"), (f"{tag}_sampler.jsonl", f"{tag}_sampler_microbench.jsonl"),
                 (f"{tag}_conv.jsonl", f"{tag}_conv_microbench.jsonl"), (f"{tag}_driver512.jsonl", f"{tag}_driver_breakdown_r512.jsonl"),
                 (f"{tag}_pipeline.jsonl", f"{tag}_pipeline_images_in_out.jsonl"), (f"{tag}_embedders.txt", f"{tag}_embedders.txt"),
                 (f"{tag}_smoke.log", f"{tag}_smoke.txt"),
                 (f"{tag}_ndhwc_pmc_sampler.json", f"{tag}_pmc_sampler_ndhwc_warp0.05.json"),
                 (f"{tag}_ndhwc_small_pmc_sampler.json", f"{tag}_pmc_sampler_ndhwc_warp0.02.json"),
                 (f"{tag}_p4tile_pmc_sampler.json", f"{tag}_pmc_sampler_lds_tile_warp0.03.json"),
                 (f"{tag}_sampler_tile.jsonl", f"{tag}_sampler_tile_sweep.jsonl"), (f"{tag}_sampler_pair.jsonl", f"{tag}_sampler_tile_pair.jsonl"),
                 (f"{tag}_mem_ceilings.jsonl", f"{tag}_mem_ceilings.jsonl"),
                 (f"{tag}_f16_kernel_stats.csv", f"{tag}_f16_stage2_kernel_stats.csv"),
                 (f"{tag}_f16_pmc_mfma.json", f"{tag}_f16_stage2_pmc_mfma.json"),
                 (f"{tag}_f16_pmc_fetch.json", f"{tag}_f16_stage2_pmc_fetch.json"),
                 (f"{tag}_f16_pmc_write.json", f"{tag}_f16_stage2_pmc_write.json"),
                 (f"{tag}_f16_conv.jsonl", f"{tag}_f16_conv_microbench.jsonl"),
                 (f"{tag}_f16_driver512.jsonl", f"{tag}_f16_driver_breakdown_r512.jsonl"),
                 (f"{tag}_f16_512c_pmc_conv.json", f"{tag}_f16_pmc_sq_conv_512to512_at64.json"),
                 (f"{tag}_f16_128c_pmc_conv.json", f"{tag}_f16_pmc_sq_conv_128to128_at512.json"),
                 (f"{tag}_conv_overhead_fit.jsonl", f"{tag}_conv_overhead_fit.jsonl"),
                 (f"{tag}_vmem_rate.jsonl", f"{tag}_vmem_rate_microbench.jsonl"),
                 (f"{tag}_mfma_stream.jsonl", f"{tag}_mfma_stream.jsonl"),
                 (f"{tag}_bf16x3_conv.jsonl", f"{tag}_bf16x3_conv_microbench.jsonl"),
                 (f"{tag}_bf16x3_512c_pmc_conv.json", f"{tag}_bf16x3_pmc_sq_conv_512to512_at64.json"),
                 (f"{tag}_bf16x3_128c_pmc_conv.json", f"{tag}_bf16x3_pmc_sq_conv_128to128_at512.json"),
                 (f"{tag}_driver512_f32.jsonl", f"{tag}_driver_breakdown_r512_fp32_mfma.jsonl"),
                 (f"{tag}_driver512_f16x2.jsonl", f"{tag}_driver_breakdown_r512_f16x2.jsonl"),
                 (f"{tag}_driver512_bf16x3.jsonl", f"{tag}_driver_breakdown_r512_bf16x3.jsonl"),
                 (f"{tag}_bench_bf16x3.json", f"{tag}_bench_n1_bf16x3.json")):
    if os.path.exists(g + src):
        if dst.endswith(".json") and "bench" in dst:      # keep the JSON line only (gloo prints a banner to stdout)
            lines = [l for l in open(g + src, errors="replace") if l.lstrip().startswith("{")]
            open(p + dst, "w").writelines(lines[-1:] if lines else [])
        else:
            shutil.copy(g + src, p + dst)
if os.path.exists(g + f"{tag}_pytest.log"):
    lines = [l for l in open(g + f"{tag}_pytest.log", errors="replace") if "PARITY" in l or "passed" in l or "failed" in l]
    open(p + f"{tag}_parity.txt", "w").writelines(lines)
f, w, m = (json.load(open(g + f"{tag}_pmc_{n}.json")) for n in ("fetch", "write", "mfma"))
for n, d in (("fetch", f), ("write", w), ("mfma", m)):
    json.dump(d, open(p + f"{tag}_pmc_{n}.json", "w"), indent=1, sort_keys=True)
rows = list(csv.DictReader(open(p + f"{tag}_bench_kernel_stats.csv")))


def _ksh():
    sys.path.insert(0, "tools")
    from kernel_source_hash import kernel_source_hash
    return kernel_source_hash()


def traffic(prefix, label, dst, suffix=None):
    """per-launch HBM traffic + MFMA utilisation of the kernels whose name starts with `prefix` (all instantiations; `suffix`
    narrows them to one template argument list ending, e.g. the SPLIT = 2 instantiations of the split kernel)"""
    pick = lambda k: k.startswith(prefix) and (suffix is None or suffix in k)
    conv = [k for k in f if pick(k)]
    if not conv:
        return None
    n = sum(f[k]["FETCH_SIZE"]["launches"] for k in conv)
    tf = sum(f[k]["FETCH_SIZE"]["sum"] for k in conv)
    tw = sum(w[k]["WRITE_SIZE"]["sum"] for k in conv)
    mf = sum(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] for k in conv)
    gui = sum(m[k]["GRBM_GUI_ACTIVE"]["sum"] for k in conv)
    ct = sum(float(r["TotalDurationNs"]) for r in rows if pick(r["Name"]))
    cc = sum(int(r["Calls"]) for r in rows if pick(r["Name"]))
    out = {
        "kernel": label,
        "pmc_run": "rocprofv3 --pmc <one counter set> (separate passes for FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) "
                   "-- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-source-pass (batch 16, as the bench)",
        "git_head": os.popen("git rev-parse --short HEAD 2>/dev/null").read().strip(),
        # (bench.py quotes the file as `traffic` only while the kernel sources still hash to this: tools/kernel_source_hash.py)
        "kernel_source_sha16": _ksh(),
        "launches": n, "fetch_size_kib_per_launch": tf / n, "write_size_kib_per_launch": tw / n,
        "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated (it equals the output tensor bytes exactly on the large layers)",
        "hbm_bytes_per_launch": (2 * tf + tw) * 1024 / n,
        "mfma_util": mf / (gui / 8 * 1024),
        "mfma_util_formula": "sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); GRBM_GUI_ACTIVE sums the 8 XCDs",
        "kernel_trace_avg_launch_ms": ct / cc / 1e6 if cc else None, "kernel_trace_launches": cc,
    }
    json.dump(out, open(p + dst, "w"), indent=1)
    print(json.dumps(out, indent=1))
    return out


traffic("conv_igemm_kernel", "conv_igemm_kernel (fp32 MFMA, all instantiations)", f"{tag}_pmc_conv_traffic.json")
# round 6: the same passes hold the pointwise kernel and the image head's stream kernel
traffic("conv_igemm_bf16x3_p1_kernel", "conv_igemm_bf16x3_p1_kernel (1x1 layers on the fp16 split)", f"{tag}_pmc_conv_f16x2_p1_traffic.json")
traffic("conv_head_kernel", "conv_head_kernel (the image head as a stream)", f"{tag}_pmc_conv_head_traffic.json")
# the split kernel's two operand modes are instantiations of one template: <TR, TW, UPS, SPLIT>.  A default (f16x2) run also holds
# the guarded bf16x3 launches, which leave at once: they would dilute a per-launch figure, so only a bf16x3 run's trace is used
if any(k.startswith("conv_igemm_bf16x3_ct2_kernel") or __import__("re").match(r"conv_igemm_bf16x3_kernel<\d+, \d+, (?:true|false), 2", k) for k in f):
    # (default mode, round 5: a layer is up to two kernels + its guarded launch -- tools/collect_traffic.py sums them per LAYER
    # launch and records the hash of the kernel sources, which bench.py checks before quoting the file)
    os.system(f"{sys.executable} tools/collect_traffic.py {tag}")
else:
    traffic("conv_igemm_bf16x3_kernel", "conv_igemm_bf16x3_kernel (fp32 3x3 conv on the bf16 matrix pipes, all instantiations)",
            f"{tag}_pmc_conv_bf16x3_traffic.json", suffix=", 3")
