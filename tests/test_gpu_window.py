"""Several forward calls of the reference run as ONE launch chain: the micro-batches of a gradient-accumulation
window (s1_pretrain_rna.py:294-305) stacked along dim 0, with what the reference computes per call kept per call --
the key stabiliser of the attention (one max per call, performer_pytorch.py:89-90), the mean of the loss
(s1_pretrain_rna.py:297) and the mask draws (s1_pretrain_rna.py:113-121)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from scmomer_amd import ops, s1, train
    return ops, s1, train


def _favor(ops, q, k, v, omega_b, dout, group):
    b, h, n, _ = q.shape
    o, oc = torch.empty_like(q), torch.empty_like(q)
    sv = ops.favor_fwd(q, k, v, omega_b, o, oc, group)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    ops.favor_bwd(q, k, v, oc, dout, omega_b, sv, dq, dk, dv)
    return o, dq, dk, dv, sv


def test_grouped_attention_equals_one_call_per_group(mods):
    """7 cells in groups of 3 (3 + 3 + 1): output and the three input gradients of every group carry the bits of a
    call over that group alone (the stabiliser gradient's single token aside: its sum over workgroup partials runs
    in another order), and differ from the ungrouped call, whose stabiliser is the max over all 7 cells."""
    ops, s1, train = mods
    torch.manual_seed(3)
    b, h, n, g = 7, 10, 1333, 3
    scale = torch.tensor([1.0, 1.0, 1.0, 1.6, 1.6, 1.6, 0.7], device=DEV).view(b, 1, 1, 1)
    q = torch.randn((b, h, n, 64), device=DEV).to(BF16)
    k = (torch.randn((b, h, n, 64), device=DEV) * scale).to(BF16)
    v = torch.randn((b, h, n, 64), device=DEV).to(BF16)
    dout = torch.randn((b, h, n, 64), device=DEV).to(BF16)
    omega_b = ops.prep_omega(torch.randn((266, 64), device=DEV) * 1.0)
    o, dq, dk, dv, sv = _favor(ops, q, k, v, omega_b, dout, g)
    stab = sv.sums[:, 65].view(b, h)
    for lo in range(0, b, g):
        sl = slice(lo, min(b, lo + g))
        o1, dq1, dk1, dv1, sv1 = _favor(ops, q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), omega_b,
                                        dout[sl].contiguous(), None)
        assert torch.equal(stab[sl], sv1.sums[:, 65].view(-1, h))
        assert float(stab[sl].max()) == float(stab[sl].min())            # one stabiliser per group
        if sl.stop - sl.start < g:
            # the short last group splits its tokens like the full ones (8 chunks), a call over one cell alone into
            # 21: the context sums add up in another order, and a few outputs round to the neighbouring bf16 value
            for got, ref in ((o[sl], o1), (dq[sl], dq1), (dk[sl], dk1), (dv[sl], dv1)):
                err = (got.float() - ref.float()).abs()
                assert float(err.max()) <= 2e-2 * float(ref.float().abs().max())
                assert float((err > 0).float().mean()) < 1e-2
            continue
        assert torch.equal(o[sl], o1)
        assert torch.equal(dq[sl], dq1)
        assert torch.equal(dv[sl], dv1)
        diff = (dk[sl].float() - dk1.float()).abs().amax(dim=-1)          # [cells, h, n]
        assert int((diff > 0).sum()) <= 1, "only the stabiliser's arg-max token may differ"
        assert float(diff.max()) <= 2e-2 * float(dk1.float().abs().max())
    assert len({float(stab[i, 0]) for i in (0, 3, 6)}) == 3               # the groups have different maxima
    o_all = _favor(ops, q, k, v, omega_b, dout, None)[0]
    assert not torch.equal(o_all[:3], o[:3])                              # ungrouped: another stabiliser


def test_grouped_cross_entropy_is_one_mean_per_group(mods):
    ops, s1, train = mods
    torch.manual_seed(4)
    b, n, c, g = 7, 501, 7, 3
    logits = torch.randn((b, n, c), device=DEV, requires_grad=True)
    labels = torch.randint(0, 7, (b, n), device=DEV)
    loss = s1.masked_cross_entropy(logits, labels, 6, group_cells=g)
    assert tuple(loss.shape) == (3,)
    seed = torch.tensor([0.2, 0.5, 1.0], device=DEV)
    (grad,) = torch.autograd.grad(loss, logits, grad_outputs=seed)
    for i, lo in enumerate(range(0, b, g)):
        sl = slice(lo, min(b, lo + g))
        lg = logits[sl].detach().clone().requires_grad_()
        one = s1.masked_cross_entropy(lg, labels[sl].contiguous(), 6)
        assert torch.equal(one, loss[i])
        (g1,) = torch.autograd.grad(one, lg, grad_outputs=seed[i])
        assert torch.equal(grad[sl], g1)
        ref = torch.nn.functional.cross_entropy(lg.detach().transpose(1, 2), labels[sl], ignore_index=6)
        torch.testing.assert_close(one, ref, rtol=1e-5, atol=1e-6)


def _model(n, depth=2):
    import scmomer_amd as S
    torch.manual_seed(5)
    m = S.PerformerLM(num_tokens=7, dim=200, depth=depth, max_seq_len=n, heads=10, local_attn_heads=0,
                      g2v_position_emb=True)
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        m.pos_emb.emb.weight[:n] = torch.randn((n, 200), generator=gen, dtype=torch.float64)
    m.fix_projection_matrices_()
    return m.to(DEV)


@pytest.mark.parametrize("use_graph", [False, True])
def test_window_of_five_equals_five_micro_steps(mods, use_graph):
    """grad_acc 5, batch 3: ten micro-batches as ten micro-steps against two windows of five (15 cells per launch
    chain), both starting from the same generator state.  In the first window (same weights) the losses and
    accuracies of every micro-batch are identical to the bit: same masks, and the forward of every cell carries the
    same bits.  The accumulated gradients differ by the order in which the token sums of the weight gradients add
    up in f32 (next test: relative L2 1.4e-7 per tensor); Adam's first updates are ~ lr * sign(g), so elements
    whose gradient is within that noise of zero move by +-lr either way: the weight updates of the two runs differ
    by 1.3e-3 relative L2 at worst (measured on MI355X), and the second window's losses by a few 1e-6."""
    ops, s1, train = mods
    n, b, G, lr = 517, 3, 5, 2e-3
    rng = np.random.default_rng(12)
    toks = [torch.from_numpy(np.minimum((rng.random((b, n)) < 0.3) * rng.gamma(2.0, 1.2, (b, n)), 5).astype(np.int64)).to(DEV)
            for _ in range(2 * G)]
    out = {}
    for mode in ("steps", "window"):
        m = _model(n).train()
        tr = train.S1Trainer(m, learning_rate=lr, grad_acc=G, use_graph=use_graph, window=G if mode == "window" else 1)
        tr.scheduler.step(), tr.scheduler.step()
        if use_graph:
            tr.prepare(toks[0]) if mode == "steps" else tr.prepare(torch.cat(toks[:G]), G)
        torch.manual_seed(77)                                 # both runs consume the same draws
        losses, accs = [], []
        for w in range(2):
            batch = toks[w * G:(w + 1) * G]
            if mode == "steps":
                for t in batch:
                    l_, a_ = tr.micro_step(t)
                    losses.append(l_.reshape(1)), accs.append(a_.reshape(1))
            else:
                l_, a_ = tr.window_step(torch.cat(batch))
                losses.append(l_), accs.append(a_)
        assert tr.index == 2 * G
        out[mode] = (torch.cat(losses).cpu(), torch.cat(accs).cpu(), {k: v.cpu() for k, v in m.state_dict().items()})
    (l_s, a_s, sd_s), (l_w, a_w, sd_w) = out["steps"], out["window"]
    assert torch.isfinite(l_s).all() and torch.isfinite(a_s).all()
    print("losses", l_s.tolist(), l_w.tolist(), "acc", a_s.tolist(), a_w.tolist())
    assert torch.equal(l_s[:G], l_w[:G]) and torch.equal(a_s[:G], a_w[:G])      # first window: the same weights
    torch.testing.assert_close(l_w, l_s, rtol=1e-4, atol=0)                  # second: weights differ by the noise above
    assert float((a_s - a_w).abs().max()) <= 1.0 / 40
    fresh = _model(n).state_dict()
    worst = 0.0
    for k, ws_ in sd_s.items():
        if ws_.dtype != torch.float32 or k.endswith("projection_matrix"):
            continue
        upd_s, upd_w = (ws_ - fresh[k].cpu()).double().flatten(), (sd_w[k] - fresh[k].cpu()).double().flatten()
        if float(upd_s.norm()) == 0.0:
            continue
        rel = float((upd_s - upd_w).norm() / upd_s.norm())
        worst = max(worst, rel)
        assert rel <= 2e-2, f"{k}: updates differ by {rel:.3e} (relative L2)"
    print(f"graph={use_graph}: worst relative L2 between the two runs' weight updates {worst:.3e}")


def test_window_gradients_match_micro_step_gradients(mods):
    """the accumulators after one window of five against five micro-steps: relative L2 per parameter tensor"""
    ops, s1, train = mods
    n, b, G = 517, 3, 5
    rng = np.random.default_rng(13)
    toks = [torch.from_numpy(np.minimum((rng.random((b, n)) < 0.3) * rng.gamma(2.0, 1.2, (b, n)), 5).astype(np.int64)).to(DEV)
            for _ in range(G)]
    flats = {}
    for mode in ("steps", "window"):
        m = _model(n).train()
        tr = train.S1Trainer(m, learning_rate=1e-3, grad_acc=2 * G, window=G if mode == "window" else 1)
        torch.manual_seed(78)
        if mode == "steps":
            for t in toks:
                tr.micro_step(t)
        else:
            tr.window_step(torch.cat(toks))
        flats[mode] = [v.clone() for v in tr.averager.views]
        names = [k for k, p in m.named_parameters() if p.requires_grad]
    worst = ("", 0.0)
    for k, a, w in zip(names, flats["steps"], flats["window"]):
        rel = float((a.double() - w.double()).norm() / a.double().norm())
        if rel > worst[1]:
            worst = (k, rel)
        assert rel <= 1e-4, f"{k}: {rel:.3e}"
    print(f"worst relative L2 between window and micro-step gradients: {worst}")


@pytest.mark.parametrize("cells", [48, 96])
def test_large_windows_cross_the_2gib_line_on_the_fast_kernels(mods, cells):
    """48 and 96 cells of 16,906 genes in one launch chain: the fused q/k/v tensor is 3.1 / 6.2 GB, the FeedForward
    activations 1.35 / 2.7 GB each -- beyond what one 32-bit-addressed window covers.  Logits must carry the bits of
    the same cells run 3 at a time (stabiliser groups of 3), every parameter gradient must equal the sum of the
    3-cell gradients up to f32 summation order, and no product may have left the asynchronous-copy kernels."""
    ops, s1, train = mods
    from scmomer_amd._lib import lib
    n, g = 16906, 3
    m = _model(n, depth=1).train()
    torch.manual_seed(20 + cells)
    tok = torch.randint(0, 6, (cells, n), device=DEV)
    labels = tok.clone()
    labels[torch.rand((cells, n), device=DEV) > 0.15] = 6
    params = [p for p in m.parameters() if p.requires_grad]
    before = lib.scm_gemm_general_launches()
    m.performer.stabiliser_group_cells = g
    logits = m(tok)
    loss = s1.masked_cross_entropy(logits, labels, 6, group_cells=g)
    grads = torch.autograd.grad(loss, params, grad_outputs=torch.ones_like(loss))
    m.performer.stabiliser_group_cells = None
    assert lib.scm_gemm_general_launches() == before, "a product fell back to the general kernel"
    logits = logits.detach()
    sums = [torch.zeros_like(p, dtype=torch.float64) for p in params]
    for lo in range(0, cells, g):
        lg = m(tok[lo:lo + g])
        l1 = s1.masked_cross_entropy(lg, labels[lo:lo + g].contiguous(), 6)
        assert torch.equal(lg.detach(), logits[lo:lo + g]), f"cells {lo}..{lo + g - 1}"
        assert torch.equal(l1.detach(), loss[lo // g].detach())
        for acc, gr in zip(sums, torch.autograd.grad(l1, params)):
            acc += gr.double()
    worst = 0.0
    for (name, _), acc, gr in zip([(k, p) for k, p in m.named_parameters() if p.requires_grad], sums, grads):
        rel = float((gr.double() - acc).norm() / acc.norm())
        worst = max(worst, rel)
        assert rel <= 1e-5, f"{name}: {rel:.3e}"
    print(f"{cells} cells: worst relative L2 of a parameter gradient against the sum over 3-cell runs: {worst:.2e}")
