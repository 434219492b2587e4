"""Read / write / copy ceilings of the GPU as a CU-driven kernel sees them (HIP events, torch elementwise kernels):
what a kernel that must WRITE 25 MB per frame (every sampler call does) can reach at best.  JSON lines."""
import json

import torch

DEV = "cuda:0"


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


for mb in (25, 100, 400, 1600):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device=DEV)
    y = torch.empty(n, device=DEV)
    x.normal_()
    rec = dict(MB=mb)
    rec["fill_GBps"] = round(mb / 1e3 / t(lambda: y.fill_(1.5)), 1)                      # write only
    rec["copy_GBps_rw"] = round(2 * mb / 1e3 / t(lambda: y.copy_(x)), 1)                 # read + write
    rec["sum_GBps"] = round(mb / 1e3 / t(lambda: x.sum()), 1)                            # read only
    rec["add_inplace_GBps_rw"] = round(2 * mb / 1e3 / t(lambda: x.add_(1.0)), 1)         # read + write, same buffer
    print(json.dumps(rec), flush=True)
