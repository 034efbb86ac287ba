#!/usr/bin/env python3
"""Per-op microbenchmark at the S1 step's shapes (B=3, N=16906): time, TFLOP/s and GB/s of algorithmic bytes.
usage (GPU box): python tools/kbench.py [filter]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scmomer_amd import ops
T, D, H, F = int(os.environ.get("KB_CELLS", "3")) * 16906, 200, 10, 800
INNER = 640
dev = "cuda"
bf, f32 = torch.bfloat16, torch.float32

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

def rnd(*shape, dtype=bf): return (torch.randn(*shape, device=dev) * 0.5).to(dtype)
cases = []
def gemm_case(name, epi, a_dtype, K, N, out_dtype, extra=None, nbytes=None):
    a = rnd(T, K, dtype=a_dtype); w = rnd(N, K); out0 = torch.empty(T, N, dtype=out_dtype, device=dev)
    kw = dict(out0=out0)
    by = T * K * a.element_size() + T * N * out0.element_size()
    if extra == "gelu": kw.update(out1=torch.empty(T, N, dtype=bf, device=dev), bias=rnd(N, dtype=f32)); by += T * N * 2
    if extra == "dgelu": kw.update(pre=rnd(T, N)); by += T * N * 2
    if extra == "res": kw.update(bias=rnd(N, dtype=f32), resid=rnd(T, N, dtype=f32)); by += T * N * 4
    cases.append((name, lambda: ops.gemm_nt(epi, a, w, **kw), 2.0 * T * K * N, by))
gemm_case("qkv       K200 N1920 bf16", ops.EPI_BF16, bf, 200, 1920, bf)
gemm_case("ffn1+gelu K200 N800  bf16x2", ops.EPI_GELU_DUAL, bf, 200, 800, bf, "gelu")
gemm_case("dO        K200 N640  f32A", ops.EPI_BF16, f32, 200, 640, bf)
gemm_case("dgelu     K200 N800  f32A", ops.EPI_DGELU, f32, 200, 800, bf, "dgelu")
gemm_case("outproj   K640 N200  f32+res", ops.EPI_F32_BIAS_RES, bf, 640, 200, f32, "res")
gemm_case("ffn2      K800 N200  f32+res", ops.EPI_F32_BIAS_RES, bf, 800, 200, f32, "res")
gemm_case("dh(qkv)   K1920 N200 f32", ops.EPI_F32, bf, 1920, 200, f32)
gemm_case("dh(ffn)   K800 N200  f32", ops.EPI_F32, bf, 800, 200, f32)
def wgrad_case(name, P, qdtype, trans, bias):
    pm = rnd(T, P); qm = rnd(T, 200, dtype=qdtype)
    g = torch.zeros((200, P) if trans else (P, 200), device=dev); b = torch.zeros(P, device=dev) if bias else None
    cases.append((name, lambda: ops.wgrad(pm, qm, g, b, trans, False), 2.0 * T * P * 200, T * P * 2 + T * 200 * qm.element_size()))
wgrad_case("wgrad P1920 bf16", 1920, bf, False, False)
wgrad_case("wgrad P800 bf16 +bias", 800, bf, False, True)
wgrad_case("wgrad P800 bf16 trans", 800, bf, True, False)
wgrad_case("wgrad P640 bf16 trans", 640, bf, True, False)
def ln_cases():
    x = rnd(T, D, dtype=f32); dy = rnd(T, D, dtype=f32); dres = rnd(T, D, dtype=f32)
    g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    h, mean, rstd = ops.ln_fwd(x, g, b)
    dg, db, dc = (torch.zeros(D, device=dev) for _ in range(3))
    jobs = []
    cases.append(("ln_fwd    f32 -> bf16", lambda: ops.ln_fwd(x, g, b), 0.0, T * D * 6))
    def bwd():
        ops.ln_bwd(dy, x, mean, rstd, g, dres, dg, db, dc, False, True, jobs); jobs.clear()
    cases.append(("ln_bwd    dy, x, dres -> dx f32 + bf16", bwd, 0.0, T * D * 18))
ln_cases()
filt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, fn, flops, by in cases:
    if filt and filt not in name: continue
    dt = timeit(fn)
    if name.startswith("ln_"):   # short kernels: the wrapper's host time exceeds the launch; take the kernel's own time
        ops.trace_collect(); ops.trace_enable("all")
        for _ in range(20): fn()
        torch.cuda.synchronize(); ops.trace_enable(None)
        tr = ops.trace_collect()
        k = max(tr, key=lambda kk: tr[kk][0])
        dt = tr[k][0] / tr[k][2] * 1e-3
    print(f"{name:32s} {dt*1e6:8.1f} us  {flops/dt/1e12:7.1f} TFLOP/s  {by/dt/1e12:5.2f} TB/s")

# ---- FAVOR+ attention passes (per-kernel times from the library's event trace)
if not filt or "favor" in filt:
    B, N = 3, 16906
    qkv = (torch.randn(B, N, 3, H, 64, device=dev) * 0.5).to(bf)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    omega = ops.prep_omega(torch.randn(266, 64, device=dev) * 1.0)
    o = torch.empty(B, N, H, 64, dtype=bf, device=dev); oc = torch.empty_like(o)
    do = (torch.randn(B, N, H, 64, device=dev) * 0.1).to(bf)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    def run():
        sv = ops.favor_fwd(q, k, v, omega, o.permute(0, 2, 1, 3), oc.permute(0, 2, 1, 3))
        ops.favor_bwd(q, k, v, oc.permute(0, 2, 1, 3), do.permute(0, 2, 1, 3), omega, sv, dq, dk, dv)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ops.trace_collect(); ops.trace_enable("all")
    reps = 10
    for _ in range(reps): run()
    torch.cuda.synchronize(); ops.trace_enable(None)
    for name, (ms, fl, cnt, by) in sorted(ops.trace_collect().items(), key=lambda kv: -kv[1][0]):
        print(f"{name:32s} {ms / cnt * 1e3:8.1f} us  {fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s  {by / (ms * 1e-3) / 1e12:5.2f} TB/s")

# ---- weight-gradient sweep: P width and token slices (filter "wsweep"), kernel time without the slab reduction
if filt == "wsweep":
    from scmomer_amd._lib import lib, check
    for P in (128, 256, 640, 800, 1920):
        pm = rnd(T, P); qm = rnd(T, 200)
        g = torch.zeros(P, 200, device=dev)
        ptiles = (P + 127) // 128
        for slices in sorted({max(1, 128 // ptiles), max(1, 256 // ptiles), max(1, 512 // ptiles), max(1, 768 // ptiles)}):
            ws = torch.empty(lib.scm_wgrad_ws_floats(P, slices), dtype=f32, device=dev)
            def call():
                check(lib.scm_wgrad(pm.data_ptr(), P, qm.data_ptr(), 200, 0, T, P, 200, slices, ws.data_ptr(), g.data_ptr(),
                                    None, 0, 0, torch.cuda.current_stream().cuda_stream), "wgrad")
            for _ in range(3): call()
            torch.cuda.synchronize(); ops.trace_collect(); ops.trace_enable("all")
            for _ in range(10): call()
            torch.cuda.synchronize(); ops.trace_enable(None)
            tr = ops.trace_collect()
            by = T * (P + 200) * 2
            wg, rd = tr["wgrad"][0] / 10, tr["wgrad_reduce"][0] / 10
            print(f"P {P:5d} slices {slices:4d} WGs {ptiles * slices:4d}  wgrad {wg * 1e3:7.1f} us ({by / wg / 1e9:5.2f} TB/s)  "
                  f"reduce {rd * 1e3:6.1f} us  slab {ws.numel() * 4 / 1e6:6.1f} MB")

# ---- do two independent weight-gradient products overlap when issued on two streams? (filter "wpair")
if filt == "wpair":
    pm1 = rnd(T, 800); qm1 = rnd(T, 200); g1 = torch.zeros(800, 200, device=dev); b1 = torch.zeros(800, device=dev)
    pm2 = rnd(T, 800); qm2 = rnd(T, 200); g2 = torch.zeros(200, 800, device=dev)
    pm3 = rnd(T, 1920); g3 = torch.zeros(1920, 200, device=dev)
    pm4 = rnd(T, 640); g4 = torch.zeros(200, 640, device=dev)
    s1_, s2_ = torch.cuda.Stream(), torch.cuda.Stream()
    def seq(a, b):
        a(); b()
    def par(a, b):
        cur = torch.cuda.current_stream()
        s1_.wait_stream(cur); s2_.wait_stream(cur)
        with torch.cuda.stream(s1_): a()
        with torch.cuda.stream(s2_): b()
        cur.wait_stream(s1_); cur.wait_stream(s2_)
    A = lambda: ops.wgrad(pm1, qm1, g1, b1, False, False)
    B = lambda: ops.wgrad(pm2, qm2, g2, None, True, False)
    Cq = lambda: ops.wgrad(pm3, qm1, g3, None, False, False)
    Do = lambda: ops.wgrad(pm4, qm2, g4, None, True, False)
    for name, a, b in (("ffn pair (w1+bias, w2)", A, B), ("attn pair (qkv, wo)", Cq, Do)):
        print(f"{name:28s} sequential {timeit(lambda: seq(a, b)) * 1e6:7.1f} us   two streams {timeit(lambda: par(a, b)) * 1e6:7.1f} us")

# ---- FeedForward sublayer forward in one launch against the kernels it replaces (filter "chain")
if filt == "chain":
    x = rnd(T, D, dtype=f32); gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
    w1 = rnd(F, D); w2 = rnd(D, F); b1 = rnd(F, dtype=f32); b2 = rnd(D, dtype=f32)
    w2c = ops.chain_copy(w2)
    def sep_fwd():
        h, mean, rstd = ops.ln_fwd(x, gamma, beta)
        pre = ops.empty_rows(T, F, bf, dev); act = ops.empty_rows(T, F, bf, dev)
        ops.gemm_nt(ops.EPI_GELU_DUAL, h, w1, out0=pre, out1=act, bias=b1)
        y = torch.empty_like(x)
        ops.gemm_nt(ops.EPI_F32_BIAS_RES, act, w2, out0=y, bias=b2, resid=x)
        return ops.ln_fwd(y, gamma, beta)
    print(f"separate (ln, ffn1, ffn2, next ln)          {timeit(sep_fwd) * 1e6:7.1f} us")
    print(f"one launch (saving tape, next ln)           {timeit(lambda: ops.ffn_chain_fwd(x, gamma, beta, w1, b1, w2c, b2, save=True, next_ln=(gamma, beta))) * 1e6:7.1f} us")
    print(f"one launch (no tape, next ln)               {timeit(lambda: ops.ffn_chain_fwd(x, gamma, beta, w1, b1, w2c, b2, save=False, next_ln=(gamma, beta))) * 1e6:7.1f} us")
    print(f"one launch (no tape)                        {timeit(lambda: ops.ffn_chain_fwd(x, gamma, beta, w1, b1, w2c, b2, save=False)) * 1e6:7.1f} us")

# ---- data-gradient product + LayerNorm backward: one launch against the two it replaces (filter "lnfuse")
if filt == "lnfuse":
    x = rnd(T, D, dtype=f32); gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
    dres = rnd(T, D, dtype=f32)
    _, mean, rstd = ops.ln_fwd(x, gamma, beta)
    dg, db, dc = (torch.zeros(D, device=dev) for _ in range(3))
    for K in (1920, 800):
        a = rnd(T, K); w = rnd(D, K)
        dh = torch.empty(T, D, dtype=f32, device=dev)
        jobs = []
        def sep():
            ops.gemm_nt(ops.EPI_F32, a, w, out0=dh)
            ops.ln_bwd(dh, x, mean, rstd, gamma, dres, dg, db, dc, False, True, jobs); jobs.clear()
        def one():
            ops.gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres, dg, db, dc, False, True, jobs); jobs.clear()
        for name, fn in (("separate", sep), ("one launch", one)):
            for _ in range(3): fn()
            torch.cuda.synchronize(); ops.trace_collect(); ops.trace_enable("all")
            for _ in range(20): fn()
            torch.cuda.synchronize(); ops.trace_enable(None)
            tr = ops.trace_collect()
            print(f"K {K:5d} {name:11s} " + "  ".join(f"{k} {v[0] / v[2] * 1e3:6.1f} us" for k, v in tr.items()) +
                  f"   total {sum(v[0] / v[2] for v in tr.values()) * 1e3:6.1f} us")

# ---- residual product + next LayerNorm forward: one launch against the two it replaces (filter "resln")
if filt == "resln":
    gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev); bias = rnd(D, dtype=f32)
    resid = rnd(T, D, dtype=f32)
    for K in (640, 800):
        a = rnd(T, K); w = rnd(D, K)
        y = torch.empty(T, D, dtype=f32, device=dev)
        def sep():
            ops.gemm_nt(ops.EPI_F32_BIAS_RES, a, w, out0=y, bias=bias, resid=resid)
            ops.ln_fwd(y, gamma, beta)
        def one():
            ops.gemm_res_ln(a, w, bias, resid, gamma, beta)
        for name, fn in (("separate", sep), ("one launch", one)):
            for _ in range(3): fn()
            torch.cuda.synchronize(); ops.trace_collect(); ops.trace_enable("all")
            for _ in range(20): fn()
            torch.cuda.synchronize(); ops.trace_enable(None)
            tr = ops.trace_collect()
            print(f"K {K:5d} {name:11s} " + "  ".join(f"{k} {v[0] / v[2] * 1e3:6.1f} us" for k, v in tr.items()) +
                  f"   total {sum(v[0] / v[2] for v in tr.values()) * 1e3:6.1f} us")

# ---- two weight gradients in one launch against two launches, reductions included (filter "wpair2")
if filt == "wpair2":
    for (pa, pb, bias) in ((800, 800, True), (640, 1920, False)):
        pm0, pm1, q0, q1 = rnd(T, pa), rnd(T, pb), rnd(T, 200), rnd(T, 200)
        g0 = torch.zeros(200, pa, device=dev); g1 = torch.zeros(pb, 200, device=dev)
        b1 = torch.zeros(pb, device=dev) if bias else None
        def two():
            jobs = []
            ops.wgrad(pm0, q0, g0, None, True, False, jobs); ops.wgrad(pm1, q1, g1, b1, False, False, jobs); ops.reduce_batch(jobs)
        def one():
            jobs = []
            ops.wgrad_multi([(pm0, q0, g0, None, True), (pm1, q1, g1, b1, False)], False, jobs); ops.reduce_batch(jobs)
        for name, fn in (("two launches", two), ("one launch", one)):
            for _ in range(3): fn()
            torch.cuda.synchronize(); ops.trace_collect(); ops.trace_enable("all")
            for _ in range(20): fn()
            torch.cuda.synchronize(); ops.trace_enable(None)
            tr = ops.trace_collect()
            print(f"P {pa}+{pb} {name:12s} " + "  ".join(f"{k} {v[0] / 20 * 1e3:6.1f} us" for k, v in tr.items()) +
                  f"   total {sum(v[0] / 20 for v in tr.values()) * 1e3:6.1f} us")

# ---- a layer's four weight gradients: two paired launches against one (filter "wquad")
if filt == "wquad":
    pa, pb, pc, pd = rnd(T, 800), rnd(T, 800), rnd(T, 640), rnd(T, 1920)
    qs = [rnd(T, 200) for _ in range(4)]
    ga, gb_, gc, gd = torch.zeros(200, 800, device=dev), torch.zeros(800, 200, device=dev), torch.zeros(200, 640, device=dev), torch.zeros(1920, 200, device=dev)
    bb = torch.zeros(800, device=dev)
    A, B, Cc, Dd = (pa, qs[0], ga, None, True), (pb, qs[1], gb_, bb, False), (pc, qs[2], gc, None, True), (pd, qs[3], gd, None, False)
    def pairs():
        jobs = []
        ops.wgrad_multi([A, B], False, jobs); ops.wgrad_multi([Cc, Dd], False, jobs); ops.reduce_batch(jobs)
    def quad():
        jobs = []
        ops.wgrad_multi([A, B, Cc, Dd], False, jobs); ops.reduce_batch(jobs)
    for name, fn in (("two pairs", pairs), ("one launch", quad)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); ops.trace_collect(); ops.trace_enable("all")
        for _ in range(20): fn()
        torch.cuda.synchronize(); ops.trace_enable(None)
        tr = ops.trace_collect()
        print(f"{name:12s} " + "  ".join(f"{k} {v[0] / 20 * 1e3:6.1f} us" for k, v in tr.items()) +
              f"   total {sum(v[0] / 20 for v in tr.values()) * 1e3:6.1f} us")
