"""The allocation-free LayerNorm backward (scm_ln_bwd_noalloc: column partials in an array built into the library
instead of a caller workspace) against scm_ln_bwd, through the C ABI: bit-identical outputs on two calls in a row,
the second with a smaller input (the array keeps the first call's contents), and a clear rejection of sizes the
array cannot hold."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(t, d, seed):
    gen = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    x = mk(t, d) * 1.7 + 0.3
    mean = x.mean(-1).contiguous()
    rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
    return dict(dy=mk(t, d), x=x, mean=mean, rstd=rstd, gamma=(1 + 0.1 * mk(d)).contiguous(), dres=mk(t, d))


def _run(lib, check, inp, t, d, noalloc, with_dres=True, accumulate=0):
    f32 = dict(dtype=torch.float32, device=DEV)
    dx = torch.full((t, d), float("nan"), **f32)
    dxb = torch.zeros((t, d), dtype=torch.bfloat16, device=DEV)
    dg, db, dc = (torch.full((d,), 0.25, **f32) for _ in range(3))
    s = torch.cuda.current_stream().cuda_stream
    p = lambda z: z.data_ptr()
    dres = p(inp["dres"]) if with_dres else None
    if noalloc:
        check(lib.scm_ln_bwd_noalloc(p(inp["dy"]), p(inp["x"]), p(inp["mean"]), p(inp["rstd"]), p(inp["gamma"]), dres,
                                     t, d, p(dx), p(dxb), p(dg), p(db), p(dc) if with_dres else None, accumulate, s),
              "scm_ln_bwd_noalloc")
    else:
        ws = torch.full((lib.scm_ln_bwd_ws_floats(t, d),), float("nan"), **f32)
        check(lib.scm_ln_bwd(p(inp["dy"]), p(inp["x"]), p(inp["mean"]), p(inp["rstd"]), p(inp["gamma"]), dres, t, d,
                             p(dx), p(dxb), p(ws), p(dg), p(db), p(dc) if with_dres else None, accumulate, s),
              "scm_ln_bwd")
    torch.cuda.synchronize()
    return dx, dxb, dg, db, dc


def test_ln_bwd_noalloc_equals_ln_bwd_on_consecutive_calls():
    from scmomer_amd._lib import check, lib
    # large first (fills all 1024 partial rows), then smaller ones that use a prefix of the array
    for (t, d, seed, with_dres, acc) in [(70000, 200, 1, True, 0), (333, 200, 2, True, 0), (64, 8, 3, False, 1),
                                         (5000, 256, 4, True, 1), (1, 200, 5, True, 0)]:
        inp = _inputs(t, d, seed)
        got = _run(lib, check, inp, t, d, True, with_dres, acc)
        want = _run(lib, check, inp, t, d, False, with_dres, acc)
        for name, a, b in zip(("dx", "dx_bf16", "dgamma", "dbeta", "dres_colsum"), got, want):
            assert torch.equal(a, b), f"T={t} D={d}: {name} differs"
        assert torch.isfinite(got[0]).all()
    # the plain gradient itself: against autograd of torch's layer_norm (f32, same statistics)
    t, d = 333, 200
    inp = _inputs(t, d, 2)
    dx, _, dg, db, dc = _run(lib, check, inp, t, d, True)
    xr = inp["x"].clone().requires_grad_()
    g = inp["gamma"].clone().requires_grad_()
    b0 = torch.zeros(d, device=DEV, requires_grad=True)
    y = torch.nn.functional.layer_norm(xr, (d,), g, b0, 1e-5)
    (y * inp["dy"]).sum().backward()
    torch.testing.assert_close(dx, xr.grad + inp["dres"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg, g.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db, b0.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dc, inp["dres"].sum(0), rtol=1e-4, atol=1e-4)


def test_ln_bwd_noalloc_rejects_what_the_array_cannot_hold():
    from scmomer_amd._lib import lib
    inp = _inputs(16, 264, 7)
    p = lambda z: z.data_ptr()
    out = torch.zeros((16, 264), device=DEV)
    vec = torch.zeros(264, device=DEV)
    rc = lib.scm_ln_bwd_noalloc(p(inp["dy"]), p(inp["x"]), p(inp["mean"]), p(inp["rstd"]), p(inp["gamma"]), None, 16,
                                264, p(out), None, p(vec), p(vec), None, 0, 0)
    assert rc < 0 and b"D <= 256" in lib.scm_last_error()
    rc = lib.scm_ln_bwd_noalloc(p(inp["dy"]), p(inp["x"]), p(inp["mean"]), p(inp["rstd"]), p(inp["gamma"]), None, 0,
                                200, p(out), None, p(vec), p(vec), None, 0, 0)
    assert rc < 0
