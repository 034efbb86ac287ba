#!/usr/bin/env python3
"""Phase times of the weight-gradient kernel from in-kernel stamps: prologue, stage loop (cycles per product, clock the
chip held), slab epilogue.  Needs a library built with the stamps compiled in:
    tools/build_variant.sh stamp "-DSCM_STAMPS"
    SCMOMER_HIP_LIB=$PWD/scmomer_amd/libscmomer_hip_stamp.so python tools/wgrad_stamps.py      (GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scmomer_amd import ops
from scmomer_amd._lib import lib, check
T = 3 * 16906; dev = "cuda"; bf = torch.bfloat16
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
for P, slices in ((1920, 17), (800, 36), (640, 51)):
    pm, qm = rnd(T, P), rnd(T, 200); g = torch.zeros(P, 200, device=dev)
    ptiles = (P + 127) // 128; nwg = ptiles * slices
    nfl = lib.scm_wgrad_ws_floats(P, slices)
    ws = torch.zeros(nfl + nwg * 4 * 8 * 2 + 64, dtype=torch.float32, device=dev)
    def call():
        check(lib.scm_wgrad(pm.data_ptr(), P, qm.data_ptr(), 200, 0, T, P, 200, slices, ws.data_ptr(), g.data_ptr(), None, 0, 0,
                            torch.cuda.current_stream().cuda_stream), "wgrad")
    t_end = time.time() + 2.0
    while time.time() < t_end:
        for _ in range(50): call()
        torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    d = ws[nfl:nfl + nwg * 64].view(torch.int64).view(nwg * 4, 8).cpu().double()
    pro, loop, epi, real, tiles = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
    full = tiles == tiles.max()
    clk = (loop[full] / real[full] * 100).median().item()
    span = (d[:, 6].max() - d[:, 5].min()).item() / 100.0
    print(f"P {P}: prologue {pro.median():.0f} cyc, loop {loop[full].median():.0f} cyc over {tiles.max():.0f} stages = "
          f"{(loop[full] / tiles[full] / 28).median():.1f} cyc/MFMA, epilogue {epi.median():.0f} cyc, clock {clk:.0f} MHz, "
          f"first start to last loop end {span:.1f} us, loop {loop[full].median() / clk:.1f} us")
