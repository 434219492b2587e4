#!/bin/bash
# trilinear x2 upsample, one thread per block of outputs that share their inputs: parity (bit-exact vs ATen CPU), timing on the WarpGenerator's shapes, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "upsample or avgpool" 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-200
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, sys
sys.path.insert(0, ".")
from emoportraits_amd import ops
for shape, f in (((16, 256, 8, 8, 8), (2, 2, 2)), ((16, 128, 16, 16, 16), (2, 2, 2)), ((16, 64, 32, 32, 32), (1, 2, 2)), ((16, 32, 32, 32, 32), (1, 2, 2)), ((16, 64, 16, 32, 32), (2, 2, 2))):
    x = torch.randn(*shape, device="cuda:0")
    for _ in range(3): y = ops.upsample_trilinear(x, f)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): y = ops.upsample_trilinear(x, f)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(shape, f, f"{ms*1e3:.1f} us, {gb/ms*1e3/1e3:.2f} TB/s of its bytes")
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-source-pass 2> gpurun_out/r4_c31_bench.err | tee gpurun_out/r4_c31_bench.json | cut -c1-160
timeout 300 python -m pytest tests/test_nets_gpu.py tests/test_bench_config_parity_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -3
