"""GPU parity of the one-launch FeedForward sublayer forward (scm_ffn_chain_fwd) through the C ABI:
against the separate kernels it replaces (LayerNorm, the two Linear products) and against an
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
    return dict(x=x, gamma=gamma, beta=beta, w1=w1, b1=b1, w2=w2, b2=b2, w1b=w1b, w1t=w1t, w2b=w2b, w2t=w2t, w2c=w2c)


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


def _small_model(n=700, depth=2):
    import numpy as np
    import scmomer_amd as S
    from oracle import performer_oracle as O
    p = O.make_params(3, depth=depth, max_seq_len=n)
    m = S.PerformerLM(num_tokens=7, dim=200, depth=depth, max_seq_len=n, heads=10, local_attn_heads=0)
    m.load_state_dict(p)
    m = m.cuda().eval()
    tokens = torch.from_numpy(np.random.default_rng(4).integers(0, 6, (2, n)).astype("int64")).cuda()
    return m, tokens


def test_model_without_tape_matches_the_taped_forward():
    """no_grad runs the FeedForward sublayers through the one-launch kernel (and its fused next LayerNorm); with
    gradients enabled the separate kernels run.  Same weights, same inputs: the outputs agree to bf16 rounding."""
    m, tokens = _small_model()
    with torch.no_grad():
        a = m(tokens)
        ea = m(tokens, return_encodings=True)
    with torch.enable_grad():
        b = m(tokens).detach()
        eb = m(tokens, return_encodings=True).detach()
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() <= 4e-3 * b.abs().max().item(), (a - b).abs().max().item()
    assert (ea - eb).abs().max().item() <= 4e-3 * eb.abs().max().item()


def test_weight_refresh_rewrites_the_chain_copy():
    from scmomer_amd import ops, stack
    m, tokens = _small_model(n=300, depth=1)
    with torch.no_grad():
        m(tokens)                                   # creates the copies
        stack.refresh_weights(m)                    # builds the one-launch refresh table
        w2 = m.performer.net.layers[0][1].fn.fn.w2.weight
        w2.add_(0.01 * torch.randn_like(w2))        # bumps the version, as an optimizer step does
        stack.refresh_weights(m)
        cache = stack._cache_of(m).store[("f", 0)][1]
        torch.cuda.synchronize()
        assert torch.equal(cache["w2_c"], ops.chain_copy(w2.to(torch.bfloat16)))
        assert torch.equal(cache["w2"], w2.to(torch.bfloat16))
