"""CPU experiment behind csrc/conv_igemm_bf16x3.h: how far the six-product bf16 split is from an fp64 convolution, next to a
plain fp32 convolution.  Products of bf16 values are exact in fp32, so the only approximation of the split itself is the three
dropped cross terms; everything else is accumulation rounding, which both kernels have.

    python tools/split_accuracy.py   ->   JSON: relative mean / max error of each variant on a decoder-like layer
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emoportraits_amd.pack import split_bf16x3  # noqa: E402


def main():
    torch.manual_seed(0)
    N, C, H, W, Co = 1, 128, 64, 64, 128
    x = torch.relu(torch.randn(N, C, H, W) * 3 + 0.5)
    w = torch.randn(Co, C, 3, 3) / (C * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xs = [t.float() for t in split_bf16x3(x)]
    ws = [t.float() for t in split_bf16x3(w)]
    assert torch.equal(xs[0] + xs[1] + xs[2], x) and torch.equal(ws[0] + ws[1] + ws[2], w), "the split is exact"

    def conv64(a, b):
        return F.conv2d(a.double(), b.double(), padding=1)      # exact products, (nearly) exact sums: isolates the dropped terms

    pairs6 = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]   # (activation plane, weight plane), smallest first
    six = sum(conv64(xs[a], ws[b]) for a, b in pairs6)
    three = sum(conv64(xs[a], ws[b]) for a, b in pairs6[3:])
    six32 = sum(F.conv2d(xs[a], ws[b], padding=1) for a, b in pairs6)
    scale = ref.abs().mean()
    out = {}
    for name, y in (("fp32_direct", F.conv2d(x, w, padding=1).double()), ("six_products_exact_accumulation", six),
                    ("six_products_fp32_accumulation", six32.double()), ("three_products_exact_accumulation", three)):
        e = (y - ref).abs()
        out[name] = dict(rel_mean=float(e.mean() / scale), rel_max=float(e.max() / scale))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
