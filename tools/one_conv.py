"""launch one released-shape conv a few times (PMC target):  python tools/one_conv.py [B] [cfg] [cin] [cout] [hw]"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd import ops, pack
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
B, cfg, cin, cout, hw = arg(1, 4), arg(2, 0), arg(3, 128), arg(4, 128), arg(5, 512)
x = torch.randn(B, cin, hw, hw, device="cuda:0")
w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
layer = pack.PackedConv("one", w, None, "cuda:0", cfg=cfg)
scale = torch.rand(B, cin, device="cuda:0") + 0.5
shift = torch.randn(B, cin, device="cuda:0") * 0.1
out = ops.conv_igemm(x, layer, scale, shift, relu_in=True)
for _ in range(3):
    ops.conv_igemm(x, layer, scale, shift, relu_in=True, out=out)
torch.cuda.synchronize()
