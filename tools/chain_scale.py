#!/usr/bin/env python3
"""ffn_chain_kernel: time against the hidden width at a fixed token count (24 cells): the slope is the cost of a
64-unit chunk (two chained products + GELU), the intercept the row load / LayerNorm / store phases.
usage (GPU box): python tools/chain_scale.py [cells]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scmomer_amd import ops
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T, D = cells * 16906, 200
dev, bf, f32 = "cuda", torch.bfloat16, torch.float32


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


x = torch.randn(T, D, device=dev)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
rows = []
for F in (64, 128, 256, 512, 800, 1600):
    w1 = (torch.randn(F, D, device=dev) * 0.1).to(bf); w2 = (torch.randn(D, F, device=dev) * 0.1).to(bf)
    b1 = torch.randn(F, device=dev) * 0.1; b2 = torch.randn(D, device=dev) * 0.1
    w2c = ops.chain_copy(w2)
    t = timeit(lambda: ops.ffn_chain_fwd(x, gamma, beta, w1, b1, w2c, b2, save=False, next_ln=(gamma, beta)))
    rows.append((F, t))
    print(f"F {F:5d}: {t * 1e6:8.1f} us   {4.0 * T * D * F / t / 1e12:7.1f} TFLOP/s   {T * D * 10 / t / 1e9:7.0f} GB/s of rows")
(f0, t0), (f1, t1) = rows[0], rows[-1]
slope = (t1 - t0) / ((f1 - f0) / 64)
print(f"per 64-unit chunk: {slope * 1e6:.2f} us over the launch; intercept {1e6 * (t0 - slope * f0 / 64):.1f} us")
