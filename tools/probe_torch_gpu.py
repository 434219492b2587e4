"""Measurement probe (not product, not shipped): how fast does stock PyTorch-ROCm (MIOpen/rocBLAS) run the hot-path
layer shapes on this GPU?  Gives the numbers the hand-written kernels have to beat.  Prints JSON lines."""
import json
import sys
import torch
import torch.nn.functional as F

DEV = "cuda:0"


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.backends.cudnn.benchmark = True
    shapes2d = [  # (Cin, Cout, H, k)
        (1536, 512, 64, 1), (512, 512, 64, 3), (512, 320, 128, 3), (320, 320, 128, 3), (512, 320, 128, 1),
        (320, 192, 256, 3), (192, 192, 256, 3), (192, 128, 512, 3), (128, 128, 512, 3), (192, 128, 512, 1)]
    for cin, cout, h, k in shapes2d:
        x = torch.randn(B, cin, h, h, device=DEV)
        w = torch.randn(cout, cin, k, k, device=DEV) * 0.01
        ms = timeit(lambda: F.conv2d(x, w, padding=k // 2))
        fl = 2.0 * B * cout * cin * k * k * h * h
        print(json.dumps(dict(op="conv2d", B=B, cin=cin, cout=cout, hw=h, k=k, ms=round(ms, 3), tflops=round(fl / ms / 1e9, 1))), flush=True)
    shapes3d = [(512, 256, (8, 8, 8), 3), (256, 256, (8, 8, 8), 3), (256, 128, (16, 16, 16), 3), (128, 128, (16, 16, 16), 3),
                (128, 64, (32, 32, 32), 3), (64, 64, (32, 32, 32), 3), (64, 32, (32, 64, 64), 3), (32, 32, (32, 64, 64), 3)]
    for cin, cout, dhw, k in shapes3d:
        x = torch.randn(B, cin, *dhw, device=DEV)
        w = torch.randn(cout, cin, k, k, k, device=DEV) * 0.01
        ms = timeit(lambda: F.conv3d(x, w, padding=1))
        fl = 2.0 * B * cout * cin * 27 * dhw[0] * dhw[1] * dhw[2]
        print(json.dumps(dict(op="conv3d", B=B, cin=cin, cout=cout, dhw=dhw, ms=round(ms, 3), tflops=round(fl / ms / 1e9, 1))), flush=True)
    for c, h in [(512, 64), (320, 128), (192, 256), (128, 512)]:
        x = torch.randn(B, c, h, h, device=DEV)
        wt, bs = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        ms = timeit(lambda: F.relu(F.group_norm(x, 32, wt, bs)))
        print(json.dumps(dict(op="gn+relu", B=B, c=c, hw=h, ms=round(ms, 3), GBps=round(x.numel() * 4 * 2 / ms / 1e6, 1))), flush=True)
    # fp32 GEMM ceiling via rocBLAS/hipBLASLt for reference
    for n in (4096, 8192):
        a, b = torch.randn(n, n, device=DEV), torch.randn(n, n, device=DEV)
        ms = timeit(lambda: a @ b)
        print(json.dumps(dict(op="sgemm", n=n, ms=round(ms, 3), tflops=round(2 * n ** 3 / ms / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
