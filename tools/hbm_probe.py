#!/usr/bin/env python3
"""Achievable HBM bandwidth of this device with plain streaming kernels (torch copy / add / sum), as the yardstick
for the algorithmic-bytes rates of the hot-path kernels."""
import torch

def rate(fn, nbytes, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return nbytes * iters / (a.elapsed_time(b) * 1e-3) / 1e9

for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    print(f"{mb:5d} MB  copy {rate(lambda: y.copy_(x), 8 * n):7.0f} GB/s   "
          f"read (sum) {rate(lambda: x.sum(), 4 * n):7.0f} GB/s   write (fill) {rate(lambda: y.fill_(1.0), 4 * n):7.0f} GB/s   "
          f"add {rate(lambda: torch.add(x, y, out=y), 12 * n):7.0f} GB/s")
