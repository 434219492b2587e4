"""debug helper: one conv case on the library EMO_HIP_LIB selects; where the output differs from torch"""
import math, sys, os
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from emoportraits_amd import ops, pack
from test_kernels_gpu import run_conv
from test_conv_bf16x3_gpu import CASES

which = [int(a) for a in sys.argv[1:]] or [1]
for ci in which:
    case = CASES[ci]
    for prec in ("f16x2", "bf16x3"):
        for rep in range(2):
            pack.clear_overflow_flags(torch.device("cuda:0"))
            e, got, ref = run_conv(seed=21, precision=prec, **case)
            d = (got.cpu() - ref).abs()
            flags = pack.overflow_events(torch.device("cuda:0"))
            bad = d > 1e-3
            print(f"case {ci} {prec} rep {rep}: err {e:.2e}, bad {int(bad.sum())} of {bad.numel()}, flags {flags}")
            if bad.any():
                n, c = bad.shape[0], bad.shape[1]
                per_n = bad.flatten(1).sum(1).tolist()
                per_c = bad.transpose(0, 1).flatten(1).sum(1)
                print("   per sample:", per_n, " channels with errors:", [int(i) for i in torch.nonzero(per_c).flatten()[:40]])
                sp = bad.any(1)                      # [N, (D,) H, W]
                sp2 = sp.reshape(n, -1, sp.shape[-2], sp.shape[-1]).any(1)
                rows = [int(i) for i in torch.nonzero(sp2.any(0).any(1)).flatten()]
                cols = [int(i) for i in torch.nonzero(sp2.any(0).any(0)).flatten()]
                print("   rows", rows[:70], " cols", cols[:70])
                print("   nan", int(torch.isnan(got).sum()), "max|got|", float(got.abs().max()))
