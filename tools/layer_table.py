"""Markdown table of DESIGN.md section 3.0 ("Measured") from the per-layer microbenchmark:

    python tools/bench_conv.py 16 --bf16x3 --f16x2 > gpurun_out/r4_conv.jsonl        (on the GPU box)
    python tools/layer_table.py gpurun_out/r4_conv.jsonl                              (anywhere)

One row per convolution shape of the R512 driver pass: the exact-fp32 MFMA kernel as the planner launches it, the bf16 and the
fp16 split where the layer is eligible (pack.supports_bf16x3 / bf16x3_launch_fits), fp32-equivalent TFLOP/s at 16 frames."""
import json
import sys

PEAK_F16X2 = 2500.0 / 3.0       # three fp16 products per fp32 product (bench.py)


def main(path):
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    rows = [r for r in rows if r.get("cin")]
    print("| layer (16 frames) | fp32 MFMA kernel (planner's launch) | bf16 split | fp16 split | fp16 split / (2500/3) | runs on |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        dims = "×".join(str(d) for d in r["dims"])
        kind = "1×1" if r["k"] == 1 else ("3×3×3" if len(r["dims"]) == 3 else "3×3")
        name = f"{r['cin']} → {r['cout']}, {kind}{', ×2 upsample fused' if r['ups'] else ''} @ {dims}"
        f, b, p = r.get("f16x2_tflops"), r.get("bf16x3_tflops"), r.get("planner", {}).get("tflops")
        on = "fp16 split" if f else "fp32 MFMA"
        print(f"| {name} | {p} | {b if b else '—'} | {f if f else '—'} | {('%.2f' % (f / PEAK_F16X2)) if f else '—'} | {on} |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r4_conv_microbench.jsonl")
