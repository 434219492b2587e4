"""Per-block fixed cost vs per-stage cost of the conv kernels: times one 3x3 layer (Cout, H x W, batch fixed) for several
Cin and fits  t = t_fixed + stages * t_stage  (stages = Cin / KC).   python tools/fit_conv_overhead.py [f16|f32] [HW] [Cout] [B]"""
import json, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import ops, pack
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
DEV = "cuda:0"


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


rows = []
for cin in (32, 64, 128, 256, 512):
    x = torch.randn(B, cin, S, S, device=DEV)
    w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    scale = torch.rand(B, cin, device=DEV) + 0.5
    shift = torch.randn(B, cin, device=DEV) * 0.1
    layer = pack.PackedConv("b", w, None, DEV, precision=prec)
    out = ops.conv_igemm(x, layer, scale, shift, relu_in=True)
    out = out[0] if isinstance(out, tuple) else out
    ms = timeit(lambda: ops.conv_igemm(x, layer, scale, shift, relu_in=True, out=out))
    flops = 2.0 * B * cout * cin * 9 * S * S
    rows.append((cin, ms, flops / ms / 1e9))
    print(json.dumps(dict(prec=prec, cin=cin, cout=cout, hw=S, B=B, ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1))), flush=True)
# least squares  ms = a + b * cin
n = len(rows)
sx = sum(r[0] for r in rows); sy = sum(r[1] for r in rows)
sxx = sum(r[0] ** 2 for r in rows); sxy = sum(r[0] * r[1] for r in rows)
b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
a = (sy - b * sx) / n
print(json.dumps(dict(fit_ms_fixed=round(a, 4), fit_ms_per_channel=round(b, 6), fixed_in_channels=round(a / b, 1),
                      asymptotic_tflops=round(2.0 * B * cout * 9 * S * S / b / 1e9, 1),
                      hbm_floor_ms_out_only=round(B * cout * S * S * 4 / 5.5e9, 4))))
