"""launch one conv layer a few times (PMC target):  python tools/one_conv.py cin cout H W [ups] [f16|f32] [B]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import ops, pack
cin, cout, H, W = (int(v) for v in sys.argv[1:5])
ups = len(sys.argv) > 5 and sys.argv[5] == "1"
prec = sys.argv[6] if len(sys.argv) > 6 else "f16"
B = int(sys.argv[7]) if len(sys.argv) > 7 else 16
DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
x = torch.randn(B, cin, H, W, generator=g).to(DEV)
w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
scale = (torch.rand(B, cin, generator=g) + 0.5).to(DEV)
shift = (torch.randn(B, cin, generator=g) * 0.1).to(DEV)
layer = pack.PackedConv("b", w, None, DEV, precision=prec)
out, _ = ops.conv_igemm(x, layer, scale, shift, relu_in=True, ups=ups), None
out = out[0] if isinstance(out, tuple) else out
for _ in range(6):
    ops.conv_igemm(x, layer, scale, shift, relu_in=True, ups=ups, out=out)
torch.cuda.synchronize()
