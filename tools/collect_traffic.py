"""profiles/<tag>_pmc_conv_f16x2_traffic.json from the rocprofv3 passes of tools/profile_bench.sh <tag> (default conv mode): HBM bytes
and matrix-pipe utilisation per LAYER LAUNCH of the fp16 split -- since round 5 a layer is up to two kernels (the two-tile kernel on
its channel-tile pairs, the single-tile kernel on an odd last tile) plus the guarded bf16x3 launch that normally exits at once;
the figures sum the fp16-split kernels of the run and divide by the number of layer launches (= guarded launches: one per layer).
Records the hash of the kernel sources (tools/kernel_source_hash.py): bench.py quotes the file only while they are unchanged.

    python tools/collect_traffic.py <tag>      # reads gpurun_out/<tag>_pmc_{fetch,write,mfma}.json, gpurun_out/<tag>_kernel_stats.csv
"""
import csv
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_source_hash import kernel_source_hash  # noqa: E402

tag = sys.argv[1]
g, p = "gpurun_out/", "profiles/"
f, w, m = (json.load(open(g + f"{tag}_pmc_{n}.json")) for n in ("fetch", "write", "mfma"))
for n, d in (("fetch", f), ("write", w), ("mfma", m)):
    json.dump(d, open(p + f"{tag}_pmc_{n}.json", "w"), indent=1, sort_keys=True)
shutil.copy(g + f"{tag}_kernel_stats.csv", p + f"{tag}_bench_kernel_stats.csv")
rows = list(csv.DictReader(open(p + f"{tag}_bench_kernel_stats.csv")))
import re
_mode = lambda k: (re.match(r"conv_igemm_bf16x3_kernel<\d+, \d+, (?:true|false), (\d)", k) or [None, None])[1]   # <TR, TW, UPS, SPLIT[, BMT]>
split = lambda k: k.startswith("conv_igemm_bf16x3_ct2_kernel") or _mode(k) == "2"
guard = lambda k: _mode(k) == "3"
ks = [k for k in f if split(k)]
layers = sum(f[k]["FETCH_SIZE"]["launches"] for k in f if guard(k))
tf = sum(f[k]["FETCH_SIZE"]["sum"] for k in ks)
tw = sum(w[k]["WRITE_SIZE"]["sum"] for k in ks)
mf = sum(m[k]["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] for k in ks)
gui = sum(m[k]["GRBM_GUI_ACTIVE"]["sum"] for k in ks)
ct = sum(float(r["TotalDurationNs"]) for r in rows if split(r["Name"]) or guard(r["Name"]))
cc = sum(int(r["Calls"]) for r in rows if guard(r["Name"]))
out = {
    "kernel": "fp16 split of the 3x3 layers: conv_igemm_bf16x3_ct2_kernel (two channel tiles per item) + conv_igemm_bf16x3_kernel<SPLIT = 2> "
              "(odd last tile, small launches), per LAYER launch",
    "pmc_run": "rocprofv3 --pmc <one counter set> (separate passes for FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE) "
               "-- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-source-pass --no-extras --no-graph (batch 16, as the bench)",
    "kernel_source_sha16": kernel_source_hash(),
    "git_head": os.popen("git rev-parse --short HEAD 2>/dev/null").read().strip(),
    "layer_launches": layers, "kernel_launches": {k[:60]: f[k]["FETCH_SIZE"]["launches"] for k in ks},
    "fetch_size_kib_per_launch": tf / layers, "write_size_kib_per_launch": tw / layers,
    "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated (it equals the output tensor bytes exactly on the large layers)",
    "hbm_bytes_per_launch": (2 * tf + tw) * 1024 / layers,
    "mfma_util": mf / (gui / 8 * 1024),
    "mfma_util_formula": "sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs) over the fp16-split kernels; GRBM_GUI_ACTIVE sums the 8 XCDs",
    "kernel_trace_avg_layer_ms": ct / cc / 1e6 if cc else None, "kernel_trace_layer_launches": cc,
}
json.dump(out, open(p + f"{tag}_pmc_conv_f16x2_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
