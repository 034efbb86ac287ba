"""GPU parity of the one-launch FeedForward sublayer (scm_ffn_chain_fwd / scm_ffn_chain_bwd) through the C ABI:
against the separate kernels it replaces (LayerNorm, the two Linear products, LayerNorm backward) and against an
fp32 torch restatement of PreLayerNorm(Chunk(FeedForward)) + residual (performer_pytorch.py:289-331,
reversible.py:145)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

D, F = 200, 800


def _setup(m, seed=0):
    from scmomer_amd import ops
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    x = (torch.randn(m, D, generator=g) * 1.3 + 0.2).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    w1 = (torch.randn(F, D, generator=g) * 0.07).to(dev)
    b1 = (torch.randn(F, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(D, F, generator=g) * 0.05).to(dev)
    b2 = (torch.randn(D, generator=g) * 0.1).to(dev)
    w1b, w1t = ops.cast_weight(w1)
    w2b, w2t = ops.cast_weight(w2)
    w2c = ops.chain_copy(w2b)
    w1tc = ops.chain_copy(w1t)
    return dict(x=x, gamma=gamma, beta=beta, w1=w1, b1=b1, w2=w2, b2=b2, w1b=w1b, w1t=w1t, w2b=w2b, w2t=w2t, w2c=w2c,
                w1tc=w1tc)


def _separate_fwd(p):
    from scmomer_amd import ops
    h, mean, rstd = ops.ln_fwd(p["x"], p["gamma"], p["beta"])
    m = p["x"].shape[0]
    pre = ops.empty_rows(m, F, torch.bfloat16, "cuda")
    act = ops.empty_rows(m, F, torch.bfloat16, "cuda")
    ops.gemm_nt(ops.EPI_GELU_DUAL, h, p["w1b"], out0=pre, out1=act, bias=p["b1"])
    y = torch.empty_like(p["x"])
    ops.gemm_nt(ops.EPI_F32_BIAS_RES, act, p["w2b"], out0=y, bias=p["b2"], resid=p["x"])
    return h, mean, rstd, pre, act, y


@pytest.mark.parametrize("m", [1, 31, 200, 257, 5000, 50718])
def test_forward_matches_separate_kernels_and_fp32(m):
    from scmomer_amd import ops
    p = _setup(m, seed=m)
    gn = (1 + 0.1 * torch.randn(D)).cuda()
    bn = (0.1 * torch.randn(D)).cuda()
    y, tape, nxt = ops.ffn_chain_fwd(p["x"], p["gamma"], p["beta"], p["w1b"], p["b1"], p["w2c"], p["b2"], save=True,
                                     next_ln=(gn, bn))
    h, mean, rstd, pre, act = tape
    h0, mean0, rstd0, pre0, act0, y0 = _separate_fwd(p)
    torch.cuda.synchronize()
    assert torch.allclose(mean, mean0, atol=1e-6, rtol=1e-6) and torch.allclose(rstd, rstd0, atol=0, rtol=1e-5)
    # the LayerNorm sums run in another order: a value may land on the neighbouring bf16
    assert (h.float() - h0.float()).abs().max() <= 2 ** -7 * h0.float().abs().max()
    assert (pre.float() - pre0.float()).abs().max() <= 2e-2 * max(1.0, pre0.float().abs().max().item())
    assert (act.float() - act0.float()).abs().max() <= 2e-2 * max(1.0, act0.float().abs().max().item())
    assert torch.allclose(y, y0, atol=3e-2, rtol=0), (y - y0).abs().max()
    # fp32 restatement
    xr = p["x"].double()
    hr = torch.nn.functional.layer_norm(xr, (D,), p["gamma"].double(), p["beta"].double(), 1e-5)
    prer = hr @ p["w1"].double().t() + p["b1"].double()
    yr = xr + torch.nn.functional.gelu(prer) @ p["w2"].double().t() + p["b2"].double()
    err = (y.double() - yr).abs().max().item()
    ref = (yr - xr).abs().max().item()
    assert err <= 8e-3 * max(ref, 1.0), (err, ref)
    # the next sublayer's LayerNorm of y
    hn, mn, rn = nxt
    hn0, mn0, rn0 = ops.ln_fwd(y, gn, bn)
    assert torch.allclose(mn, mn0, atol=1e-6, rtol=1e-6) and torch.allclose(rn, rn0, atol=0, rtol=1e-5)
    assert (hn.float() - hn0.float()).abs().max() <= 2 ** -7 * hn0.float().abs().max()


def test_forward_without_tape_writes_only_y():
    from scmomer_amd import ops
    p = _setup(777, seed=5)
    y, tape, nxt = ops.ffn_chain_fwd(p["x"], p["gamma"], p["beta"], p["w1b"], p["b1"], p["w2c"], p["b2"], save=False)
    assert tape is None and nxt is None
    y1, _, _ = ops.ffn_chain_fwd(p["x"], p["gamma"], p["beta"], p["w1b"], p["b1"], p["w2c"], p["b2"], save=True)
    assert torch.equal(y, y1)


@pytest.mark.parametrize("m", [1, 31, 257, 5000, 50718])
def test_backward_matches_separate_kernels_and_fp32(m):
    from scmomer_amd import ops
    p = _setup(m, seed=100 + m)
    h, mean, rstd, pre, act, y = _separate_fwd(p)
    dy = (torch.randn(m, D, generator=torch.Generator().manual_seed(m)) * 0.02).cuda()
    dyb = dy.to(torch.bfloat16)
    # separate kernels
    dpre0 = ops.empty_rows(m, F, torch.bfloat16, "cuda")
    ops.gemm_nt(ops.EPI_DGELU, dyb, p["w2t"], out0=dpre0, pre=pre)
    dh0 = torch.empty(m, D, device="cuda")
    ops.gemm_nt(ops.EPI_F32, dpre0, p["w1t"], out0=dh0)
    g0, b0, c0 = (torch.empty(D, device="cuda") for _ in range(3))
    dx0, dxb0 = ops.ln_bwd(dh0, p["x"], mean, rstd, p["gamma"], dy, g0, b0, c0, False, True)
    # one launch
    g1, b1, c1 = (torch.empty(D, device="cuda") for _ in range(3))
    dpre, dx, dxb = ops.ffn_chain_bwd(dyb, p["w2t"], p["w1tc"], pre, p["x"], mean, rstd, p["gamma"], dy, g1, b1, c1, False)
    torch.cuda.synchronize()
    assert torch.equal(dpre, dpre0)                      # same products, same order
    scale = dx0.abs().max().item()
    assert (dx - dx0).abs().max().item() <= 2e-3 * scale, ((dx - dx0).abs().max().item(), scale)
    assert (dxb.float() - dx).abs().max().item() <= 2 ** -8 * scale
    for a_, b_ in ((g1, g0), (b1, b0), (c1, c0)):
        assert torch.allclose(a_, b_, atol=2e-3 * max(b_.abs().max().item(), 1e-3), rtol=0), (a_ - b_).abs().max()
    # fp32/fp64 autograd restatement of the data path (dy treated exactly)
    xr = p["x"].double().requires_grad_(True)
    gam = p["gamma"].double().requires_grad_(True)
    bet = p["beta"].double().requires_grad_(True)
    hr = torch.nn.functional.layer_norm(xr, (D,), gam, bet, 1e-5)
    yr = xr + torch.nn.functional.gelu(hr @ p["w1"].double().t() + p["b1"].double()) @ p["w2"].double().t()
    yr.backward(dy.double())
    rel = ((dx.double() - xr.grad).norm() / xr.grad.norm()).item()
    assert rel <= 6e-3, rel
    assert ((g1.double() - gam.grad).norm() / gam.grad.norm()).item() <= 2e-2
    assert ((b1.double() - bet.grad).norm() / bet.grad.norm()).item() <= 2e-2
    assert torch.allclose(c1.double(), dy.double().sum(0), atol=1e-4 * m ** 0.5 + 1e-5, rtol=1e-4)
