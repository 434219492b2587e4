#!/bin/bash
# round 5, call 1: the advisor fixes and the multi-GPU additions on the GPU (new tests), the strong-scaling line under 8 ranks on
# the one GPU of the box (functional), and a short N = 1 bench of the unchanged kernels for this round's box calibration
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
F() { grep -v amdgpu.ids; }
timeout 900 python -m pytest tests/test_conv_bf16x3_gpu.py tests/test_two_ranks_gpu.py -m gpu -q -x 2>&1 | F | tail -6
echo "--- bench N=1 (short)"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r5_c1_bench_n1.json 2> gpurun_out/r5_c1_bench_n1.err; tail -c 600 gpurun_out/r5_c1_bench_n1.err | F
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_c1_bench_n1.json").read().strip().splitlines()[-1])
print("N1", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("strong_scaling"))
PY
echo "--- 8 ranks on one GPU (gloo), weak line + strong figure"
EMO_FORCE_DEVICE=0 EMO_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --strong-frames 77 > gpurun_out/r5_c1_bench_8ranks_weak.json 2> gpurun_out/r5_c1_bench_8ranks_weak.err; tail -c 800 gpurun_out/r5_c1_bench_8ranks_weak.err | F
echo "--- 8 ranks on one GPU (gloo), strong line"
EMO_FORCE_DEVICE=0 EMO_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --total-frames 77 > gpurun_out/r5_c1_bench_8ranks_strong.json 2> gpurun_out/r5_c1_bench_8ranks_strong.err; tail -c 800 gpurun_out/r5_c1_bench_8ranks_strong.err | F
python - <<'PY'
import json
for f in ("weak","strong"):
    try:
        d=json.loads(open(f"gpurun_out/r5_c1_bench_8ranks_{f}.json").read().strip().splitlines()[-1])
        print(f, d["scaling"], d["n_gpus"], d["value"], d["ms_per_step"], d.get("strong_scaling"), d.get("broadcast_ms"))
    except Exception as e:
        print(f, "FAILED", e)
PY
nvidia-smi 2>/dev/null; rocm-smi --showmeminfo vram 2>/dev/null | F | tail -4
