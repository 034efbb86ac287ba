"""GPU parity tests of every C-ABI op against the CPU oracle (float64 on the same bf16-rounded
operands, so what is measured is the kernel's own arithmetic, not the operand quantisation).

Tolerances: matrix products take bf16 operands and accumulate in f32; outputs stored as bf16
carry 2^-9 relative rounding.  Stated per test below.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import performer_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from scmomer_amd import ops as _ops
    return _ops


DEV = "cuda"


def r16(t):
    """round to bf16 and back (float32)"""
    return t.to(torch.bfloat16).to(torch.float32)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# --------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(300, 200, 640), (257, 1920, 200), (130, 800, 200), (64, 640, 200),
                                   (513, 200, 1920), (100, 96, 72),
                                   # the A-in-registers kernel at its other contraction length, a partial last column
                                   # chunk and row counts that split quads of rows
                                   (257, 640, 208), (190, 328, 208), (131, 264, 200)])
@pytest.mark.parametrize("a_f32", [False, True])
def test_gemm_nt_plain(ops, m, n, k, a_f32):
    a = rnd(m, k, seed=1)
    w = rnd(n, k, seed=2) * 0.1
    ref = r16(a).double() @ r16(w).double().t()
    ad = (a if a_f32 else a.to(torch.bfloat16)).to(DEV)
    wd = w.to(torch.bfloat16).to(DEV)
    out32 = torch.empty((m, n), dtype=torch.float32, device=DEV)
    ops.gemm_nt(ops.EPI_F32, ad, wd, out0=out32)
    assert rel_err(out32, ref) < 2e-5          # f32 accumulation of exact bf16 products
    out16 = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(ops.EPI_BF16, ad, wd, out0=out16)
    assert rel_err(out16.float(), ref) < 5e-3  # + bf16 output rounding (2^-9)


def test_gemm_nt_epilogues(ops):
    m, n, k = 333, 800, 200
    a, w = rnd(m, k, seed=3), rnd(n, k, seed=4) * 0.1
    bias, resid = rnd(n, seed=5), rnd(m, n, seed=6)
    acc = r16(a).double() @ r16(w).double().t()
    ad, wd = a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV)
    # bias + residual, f32 out
    out = torch.empty((m, n), dtype=torch.float32, device=DEV)
    ops.gemm_nt(ops.EPI_F32_BIAS_RES, ad, wd, out0=out, bias=bias.to(DEV), resid=resid.to(DEV))
    assert rel_err(out, acc + bias.double() + resid.double()) < 2e-5
    # gelu dual
    pre = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
    act = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(ops.EPI_GELU_DUAL, ad, wd, out0=pre, out1=act, bias=bias.to(DEV))
    p64 = acc + bias.double()
    assert rel_err(pre.float(), p64) < 5e-3
    assert rel_err(act.float(), O.gelu_erf(p64)) < 5e-3
    # dgelu: out = acc * gelu'(pre)
    pre_in = r16(rnd(m, n, seed=7))
    out16 = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(ops.EPI_DGELU, ad, wd, out0=out16, pre=pre_in.to(torch.bfloat16).to(DEV))
    x = pre_in.double().requires_grad_()
    O.gelu_erf(x).sum().backward()
    assert rel_err(out16.float(), acc * x.grad) < 5e-3
    # bf16 + bias
    ops.gemm_nt(ops.EPI_BF16_BIAS, ad, wd, out0=out16, bias=bias.to(DEV))
    assert rel_err(out16.float(), p64) < 5e-3


@pytest.mark.parametrize("t,p", [(1000, 640), (64, 160), (4097, 1920), (130, 800)])
@pytest.mark.parametrize("q_f32,trans,with_bias", [(False, False, True), (True, True, False), (False, True, False),
                                                   (True, False, True)])
def test_wgrad(ops, t, p, q_f32, trans, with_bias):
    q = 200
    pm, qm = rnd(t, p, seed=8), rnd(t, q, seed=9)
    ref = r16(pm).double().t() @ r16(qm).double()
    pmd = pm.to(torch.bfloat16).to(DEV)
    qmd = (qm if q_f32 else qm.to(torch.bfloat16)).to(DEV)
    base = rnd(q, p, seed=10) if trans else rnd(p, q, seed=10)
    grad = base.clone().to(DEV)
    bg0 = rnd(p, seed=11)
    bgrad = bg0.clone().to(DEV) if with_bias else None
    ops.wgrad(pmd, qmd, grad, bgrad, trans, True)
    want = base.double() + (ref.t() if trans else ref)
    assert rel_err(grad, want) < 3e-5
    if with_bias:
        assert rel_err(bgrad, bg0.double() + r16(pm).double().sum(0)) < 3e-5
    # overwrite mode
    ops.wgrad(pmd, qmd, grad, bgrad, trans, False)
    assert rel_err(grad, (ref.t() if trans else ref)) < 3e-5


@pytest.mark.parametrize("t,p0,p1", [(1000, 640, 1920), (50718, 800, 800), (130, 160, 640), (4097, 1920, 800)])
def test_wgrad_multi_equals_separate_calls(ops, t, p0, p1):
    """several products in one launch against scm_wgrad per product: the same kernel code per product; only the token
    slices (so the grouping of the f32 partial sums) differ"""
    q = 200
    pm0 = (rnd(t, p0, seed=51) * 0.5).to(torch.bfloat16).to(DEV)
    pm1 = (rnd(t, p1, seed=52) * 0.5).to(torch.bfloat16).to(DEV)
    qm0 = (rnd(t, q, seed=53) * 0.5).to(torch.bfloat16).to(DEV)
    qm1 = (rnd(t, q, seed=54) * 0.5).to(torch.bfloat16).to(DEV)
    assert ops.wgrad_multi_supported(pm0, qm0, pm1, qm1)
    g0r, g1r = torch.zeros(q, p0, device=DEV), torch.zeros(p1, q, device=DEV)
    b1r = torch.zeros(p1, device=DEV)
    ops.wgrad(pm0, qm0, g0r, None, True, False)
    ops.wgrad(pm1, qm1, g1r, b1r, False, False)
    g0, g1, b1 = torch.full_like(g0r, 3.0), torch.full_like(g1r, 3.0), torch.full_like(b1r, 3.0)
    ops.wgrad_multi([(pm0, qm0, g0, None, True), (pm1, qm1, g1, b1, False)], False)
    assert rel_err(g0, g0r) < 2e-5 and rel_err(g1, g1r) < 2e-5 and rel_err(b1, b1r) < 2e-5
    want = pm1.double().t() @ qm1.double()
    assert rel_err(g1, want) < 3e-5
    jobs = []
    ops.wgrad_multi([(pm0, qm0, g0, None, True), (pm1, qm1, g1, b1, False)], True, jobs)
    ops.reduce_batch(jobs)
    assert rel_err(g0, 2 * g0r) < 2e-5 and rel_err(b1, 2 * b1r) < 2e-5
    # four products (a whole layer's weight gradients), the third and fourth repeating the first two with swapped roles
    g2, b2, g3 = torch.zeros(p0, q, device=DEV), torch.zeros(p0, device=DEV), torch.zeros(q, p1, device=DEV)
    g0.zero_(); g1.zero_(); b1.zero_()
    ops.wgrad_multi([(pm0, qm0, g0, None, True), (pm1, qm1, g1, b1, False), (pm0, qm1, g2, b2, False),
                     (pm1, qm0, g3, None, True)], False)
    assert rel_err(g0, g0r) < 2e-5 and rel_err(g1, g1r) < 2e-5 and rel_err(b1, b1r) < 2e-5
    assert rel_err(g2, pm0.double().t() @ qm1.double()) < 3e-5 and rel_err(b2, pm0.double().sum(0)) < 3e-5
    assert rel_err(g3, qm0.double().t() @ pm1.double()) < 3e-5


# --------------------------------------------------------------------------------------- rows
@pytest.mark.parametrize("t", [1, 77, 5000])
def test_ln_fwd_bwd(ops, t):
    d = 200
    x = rnd(t, d, seed=12) * 2 + 0.5
    gamma, beta = 1 + 0.1 * rnd(d, seed=13), 0.1 * rnd(d, seed=14)
    dy, dres = rnd(t, d, seed=15), rnd(t, d, seed=16)
    x64 = x.double().requires_grad_()
    g64, b64 = gamma.double().requires_grad_(), beta.double().requires_grad_()
    y64 = O.layer_norm(x64, g64, b64)
    (y64 * dy.double()).sum().backward()
    y, mean, rstd = ops.ln_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV))
    assert rel_err(y.float(), y64) < 5e-3  # bf16 output
    dg = torch.zeros(d, device=DEV)
    db = torch.zeros(d, device=DEV)
    dc = torch.zeros(d, device=DEV)
    dx = ops.ln_bwd(dy.to(DEV), x.to(DEV), mean, rstd, gamma.to(DEV), dres.to(DEV), dg, db, dc, False)
    assert rel_err(dx, x64.grad + dres.double()) < 2e-5
    assert rel_err(dg, g64.grad) < 2e-5
    assert rel_err(db, b64.grad) < 2e-5
    assert rel_err(dc, dres.double().sum(0)) < 2e-5


@pytest.mark.parametrize("t,k,with_res,with_b", [(1, 640, True, True), (257, 1920, True, True), (5000, 800, False, False),
                                                 (50718, 1920, True, True), (7000, 136, True, False)])
def test_gemm_ln_bwd_equals_product_then_ln_bwd(ops, t, k, with_res, with_b):
    """the one-launch form (product rows kept on the chip) against scm_gemm_nt(F32) + scm_ln_bwd on the same operands:
    the same accumulators go through the same arithmetic, so dx agrees to rounding of the fused multiply-adds; the
    column sums are grouped differently (per workgroup of the product instead of per 64 rows)"""
    d = 200
    a = (rnd(t, k, seed=31) * 0.5).to(torch.bfloat16).to(DEV)
    w = (rnd(d, k, seed=32) * 0.1).to(torch.bfloat16).to(DEV)
    x = (rnd(t, d, seed=33) * 2 + 0.5).to(DEV)
    gamma, beta = (1 + 0.1 * rnd(d, seed=34)).to(DEV), (0.1 * rnd(d, seed=35)).to(DEV)
    dres = rnd(t, d, seed=36).to(DEV) if with_res else None
    assert ops.gemm_ln_bwd_supported(a, w)
    _, mean, rstd = ops.ln_fwd(x, gamma, beta)
    dh = torch.empty(t, d, device=DEV)
    ops.gemm_nt(ops.EPI_F32, a, w, out0=dh)
    dg0, db0, dc0 = (torch.zeros(d, device=DEV) for _ in range(3))
    ref = ops.ln_bwd(dh, x, mean, rstd, gamma, dres, dg0, db0, dc0 if with_res else None, False, with_b)
    dg1, db1, dc1 = (torch.full((d,), 7.0, device=DEV) for _ in range(3))
    got = ops.gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres, dg1, db1, dc1 if with_res else None, False, with_b)
    if with_b:
        assert rel_err(got[0], ref[0]) < 2e-6
        assert rel_err(got[1].float(), ref[1].float()) < 4e-3     # a few elements round to the neighbouring bf16 value
        assert (got[1] != ref[1]).float().mean() < 1e-3
    else:
        assert rel_err(got, ref) < 2e-6
    assert rel_err(dg1, dg0) < 2e-5 and rel_err(db1, db0) < 2e-5
    if with_res:
        assert rel_err(dc1, dc0) < 2e-5
    # deferred form: the column partials are summed by reduce_batch, accumulating into the outputs
    jobs = []
    dg2, db2 = dg1.clone(), db1.clone()
    got2 = ops.gemm_ln_bwd(a, w, x, mean, rstd, gamma, dres, dg2, db2, None, True, with_b, jobs)
    ops.reduce_batch(jobs)
    assert torch.equal(got2[0] if with_b else got2, got[0] if with_b else got)
    assert rel_err(dg2, 2 * dg1) < 2e-5 and rel_err(db2, 2 * db1) < 2e-5


@pytest.mark.parametrize("t,k,with_res", [(1, 640, True), (257, 800, True), (5000, 640, False), (50718, 800, True)])
def test_gemm_res_ln_equals_product_then_ln_fwd(ops, t, k, with_res):
    """one launch against scm_gemm_nt(F32_BIAS_RES) + scm_ln_fwd: y is the same accumulator, and the epilogue repeats
    the LayerNorm kernel's arithmetic operation by operation, so every output has the same bits"""
    d = 200
    a = (rnd(t, k, seed=41) * 0.5).to(torch.bfloat16).to(DEV)
    w = (rnd(d, k, seed=42) * 0.1).to(torch.bfloat16).to(DEV)
    bias = rnd(d, seed=43).to(DEV)
    resid = (rnd(t, d, seed=44) * 2).to(DEV) if with_res else None
    gamma, beta = (1 + 0.1 * rnd(d, seed=45)).to(DEV), (0.1 * rnd(d, seed=46)).to(DEV)
    assert ops.gemm_res_ln_supported(a, w)
    y0 = torch.empty(t, d, device=DEV)
    ops.gemm_nt(ops.EPI_F32_BIAS_RES, a, w, out0=y0, bias=bias, resid=resid)
    h0, mean0, rstd0 = ops.ln_fwd(y0, gamma, beta)
    y1, (h1, mean1, rstd1) = ops.gemm_res_ln(a, w, bias, resid, gamma, beta)
    assert torch.equal(y1, y0)
    assert torch.equal(mean1, mean0) and torch.equal(rstd1, rstd0)
    assert torch.equal(h1, h0)


def test_embed_fwd_bwd_bit_exact(ops):
    b, n, d, v = 3, 501, 200, 7
    p = O.make_params(5, depth=0, max_seq_len=n)
    tok = torch.from_numpy(np.random.default_rng(1).integers(0, v, (b, n)).astype(np.int64))
    ref = O.embed(tok, p)
    out = ops.embed_fwd(tok.to(DEV), p["token_emb.weight"].to(DEV), p["pos_emb.emb.weight"].to(DEV))
    assert torch.equal(out.cpu(), ref)  # same f64 add, one rounding: bit exact
    assert int(out._scm_err.item()) == 0
    dx = rnd(b, n, d, seed=17)
    grad = torch.zeros(v, d, device=DEV)
    ops.embed_bwd(tok.to(DEV), dx.to(DEV), v, grad, False)
    want = torch.zeros(v, d, dtype=torch.float64).index_add_(0, tok.reshape(-1), dx.reshape(-1, d).double())
    assert rel_err(grad, want) < 1e-5
    bad = tok.clone()
    bad[0, 0] = 9
    out = ops.embed_fwd(bad.to(DEV), p["token_emb.weight"].to(DEV), None)
    assert int(out._scm_err.item()) == 1


@pytest.mark.parametrize("c,relu", [(7, False), (1, True)])
def test_head_fwd_bwd(ops, c, relu):
    t, d = 1234, 200
    x = rnd(t, d, seed=18) + 0.3
    gamma, beta = 1 + 0.1 * rnd(d, seed=19), 0.1 * rnd(d, seed=20)
    w, bias = rnd(c, d, seed=21) * 0.1, rnd(c, seed=22) * 0.1
    dout, denc = rnd(t, c, seed=23), rnd(t, d, seed=24)
    leaves = [z.double().requires_grad_() for z in (x, gamma, beta, w, bias)]
    x64, g64, b64, w64, bi64 = leaves
    enc64 = O.layer_norm(x64, g64, b64)
    out64 = enc64 @ w64.t() + bi64
    if relu:
        out64 = torch.relu(out64)
    ((out64 * dout.double()).sum() + (enc64 * denc.double()).sum()).backward()
    dev = lambda z: z.to(DEV)
    enc, out, mean, rstd = ops.head_fwd(dev(x), dev(gamma), dev(beta), dev(w), dev(bias), relu, True, True)
    assert rel_err(enc, enc64) < 2e-5
    assert rel_err(out, out64) < 2e-5
    dg, db_, dw, dbias = (torch.zeros_like(dev(z)) for z in (gamma, beta, w, bias))
    dx = ops.head_bwd(dev(dout), dev(denc), out, dev(x), mean, rstd, dev(gamma), dev(beta), dev(w), relu, dg, db_, dw,
                      dbias, False)
    assert rel_err(dx, x64.grad) < 5e-5
    assert rel_err(dg, g64.grad) < 5e-5
    assert rel_err(db_, b64.grad) < 5e-5
    assert rel_err(dw, w64.grad) < 5e-5
    assert rel_err(dbias, bi64.grad) < 5e-5


# --------------------------------------------------------------------------------------- FAVOR+
def _favor_case(ops, b, h, n, scale, seed):
    gen = torch.Generator().manual_seed(seed)
    omega = O.gaussian_orthogonal_random_matrix(266, 64, gen)
    q, k, v, cot = O.attention_inputs(seed, (b, h, n, 64), scale)
    k = k * scale
    q, k, v, cot = r16(q), r16(k), r16(v), r16(cot)
    om16 = r16(omega)
    q64, k64, v64 = (z.double().requires_grad_() for z in (q, k, v))
    o64 = O.favor_attention(q64, k64, v64, om16.double())
    # layout as in the fused projection buffer: [B, N, 3, H, 64]
    buf = torch.stack([q, k, v], 0).permute(1, 3, 0, 2, 4).contiguous().to(torch.bfloat16).to(DEV)  # [B,N,3,H,64]
    qd, kd, vd = (buf[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    out = torch.zeros((b, n, h, 64), dtype=torch.bfloat16, device=DEV)
    outv = out.permute(0, 2, 1, 3)
    outc = torch.zeros((b, n, h, 64), dtype=torch.bfloat16, device=DEV).permute(0, 2, 1, 3)
    omb = ops.prep_omega(omega.to(DEV))
    saved = ops.favor_fwd(qd, kd, vd, omb, outv, outc)
    return dict(q=qd, k=kd, v=vd, out=outv, outc=outc, omb=omb, saved=saved, o64=o64, leaves=(q64, k64, v64), cot=cot,
                om16=om16)


@pytest.mark.parametrize("b,h,n,scale", [(2, 3, 300, 1.0), (1, 10, 131, 1.5), (3, 2, 1000, 0.5), (1, 1, 33, 1.0)])
def test_favor_forward(ops, b, h, n, scale):
    c = _favor_case(ops, b, h, n, scale, 31)
    o = c["out"].float()
    # bf16 exponentials / context / output roundings: ~3 independent 2^-9 relative errors
    assert rel_err(o, c["o64"]) < 1.5e-2
    assert rel_l2(o, c["o64"]) < 4e-3
    # saved stabilisers are the exact maxima of the bf16-operand scores
    q64, k64, _ = c["leaves"]
    s_q = (q64.detach() @ c["om16"].double().t())
    assert rel_err(c["saved"].rowmax.reshape(b, h, n), s_q.amax(-1)) < 1e-5
    s_k = (k64.detach() @ c["om16"].double().t())
    assert abs(float(c["saved"].kmax.max()) - float(s_k.max())) < 1e-4 * max(1.0, abs(float(s_k.max())))


def test_favor_forward_without_tape_writes_the_same_output(ops):
    """out_centered=None (no backward will follow): the centred output and the per-token statistics are skipped, the
    attention output is the same to the bit."""
    c = _favor_case(ops, 2, 3, 300, 1.0, 33)
    out2 = torch.zeros_like(c["out"].permute(0, 2, 1, 3).contiguous()).permute(0, 2, 1, 3)
    saved = ops.favor_fwd(c["q"], c["k"], c["v"], c["omb"], out2, None)
    torch.cuda.synchronize()
    assert saved.rowmax is None and saved.den is None
    assert torch.equal(out2, c["out"])


@pytest.mark.parametrize("b,h,n,scale", [(2, 3, 300, 1.0), (1, 10, 131, 1.5), (3, 2, 1000, 0.5)])
def test_favor_backward(ops, b, h, n, scale):
    c = _favor_case(ops, b, h, n, scale, 32)
    q64, k64, v64 = c["leaves"]
    (c["o64"] * c["cot"].double()).sum().backward()
    dout = c["cot"].permute(0, 2, 1, 3).contiguous().to(torch.bfloat16).to(DEV).permute(0, 2, 1, 3)
    dbuf = torch.zeros((b, n, 3, h, 64), dtype=torch.bfloat16, device=DEV)
    dq, dk, dv = (dbuf[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ops.favor_bwd(c["q"], c["k"], c["v"], c["outc"], dout, c["omb"], c["saved"], dq, dk, dv)
    # Gradient tolerances.  dq and dk are differences of two bf16-rounded products
    # (numerator path minus denominator path) that cancel almost completely when the attention
    # is near uniform, so their error is bounded against the magnitude of the cancelling term
    # ("natural scale"), not only against the (possibly tiny) exact gradient.
    om = c["om16"].double()
    qd_, kd_, vd_ = (z.detach() for z in (q64, k64, v64))
    cn = 64 ** -0.25
    eq = torch.exp(cn * qd_ @ om.t() - (qd_ ** 2).sum(-1, keepdim=True) / 2 * cn ** 2
                   - (cn * qd_ @ om.t()).amax(-1, keepdim=True))
    ek = torch.exp(cn * kd_ @ om.t() - (kd_ ** 2).sum(-1, keepdim=True) / 2 * cn ** 2 - (cn * kd_ @ om.t()).max())
    ctx = (ek + 1e-4).transpose(-1, -2) @ vd_
    zz = (ek + 1e-4).sum(-2)
    den = (eq + 1e-4) @ zz.unsqueeze(-1)
    dnum = c["cot"].double() / den
    nat_q = (cn * (eq * (dnum @ ctx.transpose(-1, -2))) @ om).norm()
    dctx = (eq + 1e-4).transpose(-1, -2) @ dnum
    nat_k = (cn * (ek * (vd_ @ dctx.transpose(-1, -2))) @ om).norm()
    for name, got, want, nat in (("dv", dv, v64.grad, 0.0), ("dq", dq, q64.grad, float(nat_q)),
                                 ("dk", dk, k64.grad, float(nat_k))):
        err = float((got.float().cpu().double() - want).norm())
        ref = max(float(want.norm()), nat)
        print(f"{name}: |err| {err:.3e} |want| {float(want.norm()):.3e} natural {nat:.3e} ratio {err / ref:.3e}")
        assert err <= 1e-2 * ref, name


# --------------------------------------------------------------------------------------- deferred reductions
def test_reduce_batch_equals_the_immediate_reductions(ops):
    """scm_reduce_batch over collected jobs (weight-gradient slabs with and without a bias column, transposed output,
    accumulation into an existing gradient, LayerNorm / embedding column partials) gives the bits of the immediate
    reductions."""
    g = torch.Generator().manual_seed(77)
    t, d = 5000, 200
    pm = (torch.randn(t, 640, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    pm2 = (torch.randn(t, 800, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    qm = (torch.randn(t, d, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    x = torch.randn(t, d, generator=g).to(DEV)
    dy = (torch.randn(t, d, generator=g) * 0.1).to(DEV)
    tok = torch.randint(0, 7, (t,), generator=g).to(DEV)
    h, mean, rstd = ops.ln_fwd(x, torch.ones(d, device=DEV), torch.zeros(d, device=DEV))
    gamma = torch.ones(d, device=DEV)

    def run(defer):
        out = {}
        out["w_t"] = torch.full((d, 640), 0.5, device=DEV)            # accumulate into an existing gradient, transposed
        ops.wgrad(pm, qm, out["w_t"], None, True, True, defer)
        out["w"] = torch.empty((800, d), device=DEV)
        out["b"] = torch.empty(800, device=DEV)
        ops.wgrad(pm2, qm, out["w"], out["b"], False, False, defer)
        out["g"], out["be"], out["c"] = (torch.empty(d, device=DEV) for _ in range(3))
        out["dx"] = ops.ln_bwd(dy, x, mean, rstd, gamma, dy, out["g"], out["be"], out["c"], False, False, defer)
        out["e"] = torch.empty((7, d), device=DEV)
        ops.embed_bwd(tok, dy, 7, out["e"], False, defer)
        return out

    now = run(None)
    jobs = []
    later = run(jobs)
    assert len(jobs) == 4
    ops.reduce_batch(jobs)
    torch.cuda.synchronize()
    assert not jobs
    for k in now:
        assert torch.equal(now[k], later[k]), k
