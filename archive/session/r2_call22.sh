#!/bin/bash
# quad staging of the fp32 1x1 kernels: 1x1 layer timings (vs MIOpen), the full GPU parity suite, the bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r2}
mkdir -p $R/gpurun_out; cd $R
python - <<'PY' 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c22_conv1x1.jsonl
import json, math, sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from emoportraits_amd import ops, pack
def timeit(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
torch.backends.cudnn.benchmark = True
for cin, cout, hw in ((1536, 512, 64), (512, 320, 64), (320, 192, 128), (192, 128, 256)):
    x = torch.randn(16, cin, hw, hw, device="cuda:0")
    w = torch.randn(cout, cin, 1, 1) / math.sqrt(cin)
    sc = torch.rand(16, cin, device="cuda:0") + 0.5; sh = torch.randn(16, cin, device="cuda:0") * 0.1
    flops = 2.0 * 16 * cout * cin * hw * hw
    rec = dict(cin=cin, cout=cout, hw=hw)
    wd = w.to("cuda:0")
    rec["torch_tflops"] = round(flops / timeit(lambda: F.conv2d(x, wd)) / 1e9, 1)
    for cfg in (0, 1):
        layer = pack.PackedConv("b", w, None, "cuda:0", cfg=cfg)
        out = ops.conv_igemm(x, layer, sc, sh, relu_in=True)
        ms = timeit(lambda: ops.conv_igemm(x, layer, sc, sh, relu_in=True, out=out))
        rec[f"hip_cfg{cfg}_tflops"] = round(flops / ms / 1e9, 1)
        ref = F.conv2d(F.relu(x * sc[:, :, None, None] + sh[:, :, None, None]), wd)
        rec[f"err_cfg{cfg}"] = float((out - ref).abs().max() / ref.abs().max())
    print(json.dumps(rec), flush=True)
PY
cat gpurun_out/r2c22_conv1x1.jsonl
bash tools/r2_rerun_tests.sh $T
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2c22_bench.json; cut -c1-140 gpurun_out/r2c22_bench.json
