#!/bin/bash
# 16 x 16 position tiles for the 16-wide 3-D maps (split kernel instead of the fp32 MFMA kernel on two WarpGenerator layers):
# parity, the two layers timed in both kernels, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_conv_bf16x3_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-200
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import math, sys, torch
sys.path.insert(0, ".")
from emoportraits_amd import ops, pack
DEV = "cuda:0"
for cin, cout in ((256, 128), (128, 128)):
    x = torch.randn(16, cin, 16, 16, 16, device=DEV)
    w = torch.randn(cout, cin, 3, 3, 3) / math.sqrt(cin * 27)
    sc, sh = torch.rand(16, cin, device=DEV) + 0.5, torch.randn(16, cin, device=DEV) * 0.1
    flops = 2.0 * 16 * cout * cin * 27 * 4096
    for prec in ("f32", "bf16x3", "f16x2"):
        layer = pack.PackedConv("t", w, None, DEV, precision=prec)
        out = ops.conv_igemm(x, layer, sc, sh, relu_in=True)
        for _ in range(3): ops.conv_igemm(x, layer, sc, sh, relu_in=True, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.conv_igemm(x, layer, sc, sh, relu_in=True, out=out)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(cin, cout, "16^3", prec, f"{ms:.3f} ms, {flops / ms / 1e9:.1f} TF")
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c32_bench.err | tee gpurun_out/r4_c32_bench.json | cut -c1-160
timeout 300 python -m pytest tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3
