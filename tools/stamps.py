#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock totals of the attention tile loop from a -DSCM_STAMPS build.
usage: SCMOMER_HIP_LIB=scmomer_amd/libscmomer_hip_stamps.so python tools/stamps.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scmomer_amd import ops
from scmomer_amd._lib import LIB_PATH
raw = C.CDLL(LIB_PATH)
B, N, H = 3, 16906, 10
dev = "cuda"
qkv = (torch.randn(B, N, 3, H, 64, device=dev) * 0.5).to(torch.bfloat16)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
omega = ops.prep_omega(torch.randn(266, 64, device=dev))
o = torch.empty(B, N, H, 64, dtype=torch.bfloat16, device=dev); oc = torch.empty_like(o)
do = (torch.randn(B, N, H, 64, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
def run():
    sv = ops.favor_fwd(q, k, v, omega, o.permute(0, 2, 1, 3), oc.permute(0, 2, 1, 3))
    ops.favor_bwd(q, k, v, oc.permute(0, 2, 1, 3), do.permute(0, 2, 1, 3), omega, sv, dq, dk, dv)
for _ in range(3): run()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
raw.scm_debug_stamps(buf, 1)
reps = 5
for _ in range(reps): run()
torch.cuda.synchronize()
raw.scm_debug_stamps(buf, 0)
vals = list(buf)
tiles, wgs = vals[5], vals[7]
print("tiles stamped", tiles, "workgroup-launches", wgs)
names = ["scores+G products (issue)", "request transposed Omega frags", "exponentials / dA", "request dC^T, next rows, dz",
         "dk products (issue)"]
tot = 0
for n_, c in zip(names, vals[:5]):
    print(f"  {n_:34s} {c / max(tiles, 1):8.1f} cycles per tile")
    tot += c / max(tiles, 1)
print(f"  sum {tot:8.1f}; whole kernel per wave-0: {vals[6] / max(wgs, 1):10.0f} ticks; tiles per wave {tiles / max(wgs, 1):.1f} "
      f"-> {vals[6] / max(tiles, 1):8.1f} ticks per tile incl. everything")
