"""1x1 layers of stage 2 in the fp16-operand mode: the older fp16-operand kernel against the fp16-split pointwise kernel (p1)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from emoportraits_amd import ops, pack
DEV = "cuda:0"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (B, cin, cout, H) in ((8, 128, 256, 512), (8, 256, 512, 256), (8, 512, 1024, 128), (8, 1024, 512, 128), (8, 512, 256, 256), (8, 256, 128, 512), (16, 512, 320, 128), (16, 320, 192, 256), (16, 192, 128, 512)):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    x = torch.randn(B, cin, H, H, generator=g).to(DEV)
    rec = dict(B=B, cin=cin, cout=cout, H=H, gbytes=round((x.numel() + B * cout * H * H) * 4 / 1e9, 2))
    for prec in ("f16", "f16x2", "f32"):
        try:
            lay = pack.PackedConv("p", w, None, DEV, precision=prec)
            ms = timeit(lambda: ops.conv_igemm(x, lay))
            rec[prec] = dict(ms=round(ms, 3), plan=str(lay.last_plan), tbps=round(rec["gbytes"] / ms, 2))
        except Exception as e:
            rec[prec] = str(e)[:80]
    print(json.dumps(rec), flush=True)
    del x
