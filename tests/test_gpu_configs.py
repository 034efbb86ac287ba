"""GPU checks of the two secondary configurations: sharded embedding extraction (the loop of
evaluate_embedding.py:173-178 cut into per-rank shards) and the S3 distillation step
(s3_pretrain_missing.py:286-316: frozen Performer in train mode, trainable tail)."""
import types

import numpy as np
import pytest
import torch

from oracle import performer_oracle as O
from parity_util import margin, rell2

pytestmark = pytest.mark.gpu
DEV = "cuda"
N = 16906


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import scmomer_amd
    return scmomer_amd


def _scmomer(S, p, **kw):
    cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
    m = S.scMomer(cfg, atac_config=None, encoder=S.StudentEncoder(N, 128, dropout=0.0), **kw)
    m.load_state_dict(p, strict=False)
    return m.to(DEV)


def test_sharded_extraction_equals_one_pass(S):
    """Two contiguous shards walked independently and concatenated == one rank walking the whole list, bit for bit,
    when the shard length is a multiple of the batch size (same cells share a batch: same key stabiliser);
    otherwise the batches differ in composition and the results agree within the bf16 tolerance."""
    from scmomer_amd import extract
    p = O.make_scmomer_params(5)
    m = _scmomer(S, p).eval()
    for q in m.parameters():
        q.requires_grad = False
    cells = torch.from_numpy(O.synthetic_cells(9, 50, N))           # host matrix: batches are copied over one by one
    one = extract.extract_embeddings(m, cells, 48, 12, device=DEV)
    parts = [extract.extract_embeddings(m, cells, 48, 12, rank=r, world=2, device=DEV) for r in range(2)]
    assert one.shape == (48, 128) and parts[0].shape == (24, 128)
    assert torch.equal(torch.cat(parts, 0), one)
    assert torch.equal(extract.gather_embeddings(one, 48), one)     # no process group: the local block is the result
    # a cell source given as a function of the index range, shards of 25 cells in batches of 12 (12 + 12 + 1)
    dev_cells = cells.to(DEV)
    uneven = [extract.extract_embeddings(m, lambda lo, hi: dev_cells[lo:hi], 50, 12, rank=r, world=2, device=DEV)
              for r in range(2)]
    assert [u.shape[0] for u in uneven] == [25, 25]
    got = torch.cat(uneven, 0)
    ref = extract.extract_embeddings(m, dev_cells, 50, 12, device=DEV)
    assert margin(got, ref) <= 2e-2 and rell2(got, ref) < 4e-3
    assert torch.equal(got[:24], ref[:24])                           # the first two batches hold the same cells
    # against the CPU oracle: the first batch of the list
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        want = O.scmomer_cell_embedding(cells[:12], p)
    print(f"extraction batch vs oracle: margin {margin(one[:12], want):.3e}, rel L2 {rell2(one[:12], want):.3e}")
    assert margin(one[:12], want) <= 4e-2 and rell2(one[:12], want) < 4e-3
    empty = extract.extract_embeddings(m, cells, 1, 12, rank=1, world=2, device=DEV)
    assert empty.shape == (0, 128)


def test_s3_distillation_step_vs_oracle(S):
    """model.train(); Performer frozen; mode='one', reconstruct=True with linear decoders; BCE + MSE + distillation
    loss and its backward against autograd over the CPU oracle.  The Performer gets no gradient, its redraw counter
    ticks once per forward (the reference ticks it in train mode even when frozen), the optimizer step moves only
    the trainable tail."""
    from scmomer_amd import s3
    n_peaks, bsz = 3000, 3
    p = O.make_scmomer_params(6)
    rng = np.random.default_rng(60)
    t = lambda *s_, sc=0.05: torch.from_numpy((rng.standard_normal(s_) * sc).astype(np.float32))
    dec = {"RNA.output_mean.0.weight": t(N, 128), "RNA.output_mean.0.bias": t(N),
           "ATAC.output_mean.0.weight": t(n_peaks, 128), "ATAC.output_mean.0.bias": t(n_peaks)}
    m = _scmomer(S, {**p, **dec}, rna_decoder=s3.ReconstructNet(N, 128), atac_decoder=s3.ReconstructNet(n_peaks, 128))
    total, trainable = s3.freeze_for_distillation(m)
    assert trainable == sum(v.numel() for k, v in {**p, **dec}.items()
                            if k.startswith(("encoder.", "to_out.", "RNA.", "ATAC.")))
    m.train()
    perf = m.rna_model.model.performer
    perf.fix_projection_matrices_()
    rna = torch.from_numpy(O.synthetic_cells(61, bsz, N))
    atac_raw = torch.from_numpy((rng.random((bsz, n_peaks)) < 0.05).astype(np.float32) * 2.0)
    latent = t(bsz, 128, sc=1.0)
    step = s3.S3Step(m, lr=1e-3, grad_acc=1, pos_weight=2.0, distill_weight=0.5)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    losses = s3.distillation_losses(m, rna.to(DEV), atac_raw.to(DEV), latent.to(DEV), step.pos_weight, 0.5)
    losses["total"].backward()
    assert int(perf.calls_since_last_redraw) == 1

    # ---- oracle: the same step in f32 on the CPU
    names = [k for k in {**p, **dec} if k.startswith(("encoder.", "to_out.", "RNA.", "ATAC."))]
    leaves = {k: {**p, **dec}[k].clone().requires_grad_() for k in names}
    pp = {**p, **dec, **leaves}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    tokens = torch.round(rna).long().clamp(min=0, max=5)
    lm = {k[len("rna_model.model."):]: v for k, v in pp.items() if k.startswith("rna_model.model.")}
    with torch.no_grad():
        per_gene = O.performer_lm_forward(tokens, lm, depth=6, heads=10, head="conv")
        feat = per_gene @ pp["rna_projection.weight"].t()
        feat = feat / feat.norm(dim=-1, keepdim=True)
        feat = O.layer_norm(feat, pp["rna_ln.weight"], pp["rna_ln.bias"])
    h = rna @ pp["encoder.encoder.0.weight"].t() + pp["encoder.encoder.0.bias"]
    h = O.gelu_erf(O.layer_norm(h, pp["encoder.encoder.1.weight"], pp["encoder.encoder.1.bias"]))
    student = h @ pp["encoder.encoder.4.weight"].t() + pp["encoder.encoder.4.bias"]
    out = torch.cat([student, feat], 1) @ pp["to_out.weight"].t() + pp["to_out.bias"]
    r_atac = out @ pp["ATAC.output_mean.0.weight"].t() + pp["ATAC.output_mean.0.bias"]
    r_rna = out @ pp["RNA.output_mean.0.weight"].t() + pp["RNA.output_mean.0.bias"]
    l_atac = torch.nn.functional.binary_cross_entropy_with_logits(r_atac, (atac_raw > 0).float(),
                                                                  pos_weight=torch.tensor([2.0]))
    l_rna = torch.nn.functional.mse_loss(r_rna, rna)
    l_dis = torch.nn.functional.mse_loss(student, latent)
    (l_atac + l_rna + 0.5 * l_dis).backward()
    for k, want in (("atac", l_atac), ("rna", l_rna), ("distill", l_dis)):
        print(f"S3 loss {k}: {float(losses[k].detach()):.6f} vs oracle {float(want.detach()):.6f}")
        assert abs(float(losses[k].detach()) - float(want.detach())) <= 1e-3 * abs(float(want.detach()))
    got = dict(m.named_parameters())
    worst = 0.0
    for k in names:
        e = rell2(got[k].grad, leaves[k].grad)
        worst = max(worst, e)
        assert e < 1e-2, f"{k}: grad rel L2 {e:.3e}"
    print(f"S3: {len(names)} tail gradients, worst rel L2 {worst:.3e}")
    for k, q in got.items():
        if k.startswith(("rna_model.", "rna_projection.", "rna_ln.", "atac_ln.")):
            assert q.grad is None and not q.requires_grad, k

    # ---- the optimizer step of the driver: only the tail moves; the counter keeps ticking
    m.zero_grad(set_to_none=True)
    out_losses = step.micro_step(rna.to(DEV), atac_raw.to(DEV), latent.to(DEV))
    assert abs(float(out_losses["total"]) - float(losses["total"].detach())) <= 1e-6 * abs(float(losses["total"].detach()))
    assert int(perf.calls_since_last_redraw) == 2
    after = dict(m.named_parameters())
    for k, v in before.items():
        moved = not torch.equal(after[k].detach(), v)
        assert moved == k.startswith(("encoder.", "to_out.", "RNA.", "ATAC.")), k


def test_s3_step_graph_replay_equals_eager(S):
    """The recorded S3 micro-step (use_graph=True) gives the losses and the updated weights of the eager one over two
    optimizer steps of two micro-steps each, and ticks the frozen encoder's redraw counter once per micro-step."""
    from scmomer_amd import s3
    n_peaks, bsz = 3000, 3
    p = O.make_scmomer_params(6)
    rng = np.random.default_rng(70)
    t = lambda *s_, sc=0.05: torch.from_numpy((rng.standard_normal(s_) * sc).astype(np.float32))
    dec = {"RNA.output_mean.0.weight": t(N, 128), "RNA.output_mean.0.bias": t(N),
           "ATAC.output_mean.0.weight": t(n_peaks, 128), "ATAC.output_mean.0.bias": t(n_peaks)}
    batches = [(torch.from_numpy(O.synthetic_cells(71 + i, bsz, N)).to(DEV),
                torch.from_numpy((rng.random((bsz, n_peaks)) < 0.05).astype(np.float32)).to(DEV),
                t(bsz, 128, sc=1.0).to(DEV)) for i in range(4)]
    results = []
    for use_graph in (False, True):
        m = _scmomer(S, {**p, **dec}, rna_decoder=s3.ReconstructNet(N, 128), atac_decoder=s3.ReconstructNet(n_peaks, 128))
        s3.freeze_for_distillation(m)
        m.train()
        m.rna_model.model.performer.fix_projection_matrices_()
        step = s3.S3Step(m, lr=1e-3, grad_acc=2, pos_weight=2.0, distill_weight=0.5, use_graph=use_graph)
        losses = [float(step.micro_step(*b)["total"]) for b in batches]
        results.append((losses, {k: v.detach().clone() for k, v in m.named_parameters()},
                        int(m.rna_model.model.performer.calls_since_last_redraw)))
    (l0, w0, c0), (l1, w1, c1) = results
    assert l0 == l1, (l0, l1)
    assert c0 == c1 == len(batches)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k


@pytest.mark.parametrize("cells,genes,normalize,hidden", [(5, 16906, True, 256), (40, 1000, False, 256), (1, 33, True, 256),
                                                         (7, 700, True, 512), (3, 129, True, 896)])
def test_cell_embedding_tail_kernel_equals_the_torch_tail(cells, genes, normalize, hidden):
    """Everything of scMomer.get_cell_embedding after the encoder (scmomer.py:96-113,124-130; student encoder of
    extract_cell_emb.py:27-40) as the two-launch f32 tail kernel against the same layers run by torch in f32: the
    per-gene encoder output and the expression values are random, so only the tail is compared (sums of 16,906 f32
    terms in another order: relative 1e-5 of the row scale)."""
    import types
    import scmomer_amd as S
    from scmomer_amd import ops
    torch.manual_seed(cells)
    m = S.scMomer(types.SimpleNamespace(projection_dim=128, normalize=normalize), atac_config=None,
                  encoder=S.StudentEncoder(genes, 128, hidden_dim=hidden))
    m.rna_projection = torch.nn.Linear(genes, 128, bias=False)
    m = m.to(DEV).eval()
    with torch.no_grad():
        for p in (m.rna_ln.weight, m.rna_ln.bias, m.encoder.encoder[1].weight, m.encoder.encoder[1].bias):
            p.add_(0.3 * torch.randn_like(p))
    gene_scores = torch.relu(torch.randn((cells, genes), device=DEV))
    x = (torch.rand((cells, genes), device=DEV) < 0.2) * torch.rand((cells, genes), device=DEV) * 6
    with torch.no_grad():
        tail = m._tail_operands(x)
        assert tail is not None
        got = ops.cell_tail(gene_scores, x, *tail, normalize=normalize)
        f = m.rna_projection(gene_scores)
        if normalize:
            f = f / f.norm(dim=-1, keepdim=True)
        want = m.to_out(torch.cat([m.encoder(x), m.rna_ln(f)], 1))
    assert tuple(got.shape) == (cells, 128)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale, float((got - want).abs().max()) / scale
    m.train()
    assert m._tail_operands(x) is None                      # dropout is active in train mode: torch's tail
    m.eval()
    with torch.enable_grad():
        assert m._tail_operands(x) is None                  # gradients wanted: torch's tail


@pytest.mark.parametrize("cells,genes,peaks,dropout,per_peak_weight,group_cells",
                         [(3, 16906, 20000, True, False, 0), (5, 1000, 777, False, True, 0), (17, 515, 130, True, False, 0),
                          (32, 333, 64, True, True, 0), (1, 70, 3, False, False, 0), (15, 16906, 2000, True, False, 3),
                          (30, 400, 90, True, True, 3), (8, 300, 50, False, False, 4)])
def test_distillation_tail_kernels_equal_the_torch_tail(cells, genes, peaks, dropout, per_peak_weight, group_cells):
    """The S3 tail (s3_pretrain_missing.py:291-307 after the frozen encoder: student MLP, to_out, the two linear
    decoders, BCE-with-logits with pos_weight + two MSE losses) and its backward as the eight launches of
    ops.DistillCall against the same layers and losses under torch autograd in f32, on random operands, with the
    same dropout multipliers.  Sums of up to 16,906 f32 terms in another order: losses to 2e-6 relative, every
    gradient to 2e-5 of its largest entry; then the accumulating form adds the same gradients to what is there.
    group_cells > 0: the batch is micro-batches of that many cells stacked -- one set of loss means per micro-batch,
    the gradient of their sum (the accumulation window of :296-299 in one call)."""
    from scmomer_amd import ops
    torch.manual_seed(100 + cells)
    P, Hd, Ow = 128, 256, 128
    r = lambda *s, sc=1.0: (torch.randn(s, device=DEV) * sc)
    wp, w1, b1 = r(P, genes, sc=0.05), r(Hd, genes, sc=0.05), r(Hd, sc=0.1)
    ln_p, ln_h = (1 + r(P, sc=0.3), r(P, sc=0.3)), (1 + r(Hd, sc=0.3), r(Hd, sc=0.3))
    w2, b2, wo, bo = r(P, Hd, sc=0.1), r(P, sc=0.1), r(Ow, 2 * P, sc=0.1), r(Ow, sc=0.1)
    wa, ba, wr, br = r(peaks, Ow, sc=0.1), r(peaks, sc=0.1), r(genes, Ow, sc=0.1), r(genes, sc=0.1)
    gene_scores = torch.relu(r(cells, genes))
    x = ((torch.rand((cells, genes), device=DEV) < 0.2) * torch.rand((cells, genes), device=DEV) * 6).contiguous()
    atac_raw = ((torch.rand((cells, peaks), device=DEV) < 0.1) * 3.0).contiguous()
    latent = r(cells, P)
    pos_weight = (1 + torch.rand(peaks, device=DEV) * 3) if per_peak_weight else torch.tensor([2.5], device=DEV)
    drop = torch.nn.functional.dropout(torch.ones((cells, Hd), device=DEV), 0.1, True) if dropout else None
    gloss = torch.tensor([0.7, 1.3, 0.45], device=DEV)

    leaves = [t.clone().requires_grad_() for t in (w1, b1, ln_h[0], ln_h[1], w2, b2, wo, bo, wa, ba, wr, br)]
    tw1, tb1, tlw, tlb, tw2, tb2, two, tbo, twa, tba, twr, tbr = leaves
    f = gene_scores @ wp.t()
    f = f / f.norm(dim=-1, keepdim=True)
    rr = torch.nn.functional.layer_norm(f, (P,), ln_p[0], ln_p[1])
    h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x @ tw1.t() + tb1, (Hd,), tlw, tlb))
    if drop is not None:
        h = h * drop
    a = h @ tw2.t() + tb2
    out = torch.cat([a, rr], 1) @ two.t() + tbo
    gc = group_cells or cells
    za, zr = out @ twa.t() + tba, out @ twr.t() + tbr
    want = torch.stack([torch.stack([
        torch.nn.functional.binary_cross_entropy_with_logits(za[g0:g0 + gc], (atac_raw[g0:g0 + gc] > 0).float(),
                                                             pos_weight=pos_weight),
        torch.nn.functional.mse_loss(zr[g0:g0 + gc], x[g0:g0 + gc]),
        torch.nn.functional.mse_loss(a[g0:g0 + gc], latent[g0:g0 + gc])]) for g0 in range(0, cells, gc)])
    (want * gloss).sum().backward()

    call = ops.DistillCall(gene_scores, x, atac_raw, latent, drop, pos_weight, wp, ln_p, w1, b1, ln_h, w2, b2, wo, bo,
                           wa, ba, wr, br, normalize=True, distill_weight=0.3, group_cells=group_cells)
    got = call.forward()
    want_d, a_d = want.detach(), a.detach()
    assert tuple(got.shape) == (cells // gc, 4)
    assert float(((got[:, :3] - want_d).abs() / want_d.abs()).max()) <= 2e-6, (got, want_d)
    assert float((got[:, 3] - (want_d[:, 0] + want_d[:, 1] + 0.3 * want_d[:, 2])).abs().max()) <= 2e-6 * float(got[:, 3].max())
    assert float((call.student - a_d).abs().max()) <= 2e-5 * float(a_d.abs().max())
    grads = call.backward(gloss)
    worst = 0.0
    for name, g, leaf in zip(ops.DISTILL_GRAD_ORDER, grads, leaves):
        e = float((g.view_as(leaf) - leaf.grad).abs().max()) / float(leaf.grad.abs().max())
        worst = max(worst, e)
        assert e <= 2e-5, f"{name}: {e:.3e}"
    print(f"distillation tail, {cells} cells x {genes} genes x {peaks} peaks: worst gradient error {worst:.2e} of max")
    base = [torch.full_like(g, 0.25) for g in grads]
    call2 = ops.DistillCall(gene_scores, x, atac_raw, latent, drop, pos_weight, wp, ln_p, w1, b1, ln_h, w2, b2, wo, bo,
                            wa, ba, wr, br, normalize=True, group_cells=group_cells)
    call2.forward()
    call2.backward(gloss, grads=base, accumulate=True)
    for name, acc, g in zip(ops.DISTILL_GRAD_ORDER, base, grads):
        assert torch.equal(acc, g + 0.25), name


def test_s3_step_on_the_tail_kernels_follows_the_torch_tail(S, monkeypatch):
    """Two optimizer steps of two micro-steps with the tail on its kernels (forward, backward into the gradient
    accumulators, optimizer kernel) against the same steps with the tail and the optimizer on torch's library
    code, dropout p = 0.1 with the same multipliers in both: same losses to 1e-5, weights within 5 % of the
    distance they moved (Adam divides by the gradient's own magnitude: where a gradient entry is nearly zero, a
    rounding difference moves the weight by a visible fraction of the learning rate)."""
    from scmomer_amd import s3
    n_peaks, bsz = 3000, 3
    p = O.make_scmomer_params(6)
    rng = np.random.default_rng(80)
    t = lambda *s_, sc=0.05: torch.from_numpy((rng.standard_normal(s_) * sc).astype(np.float32))
    dec = {"RNA.output_mean.0.weight": t(N, 128), "RNA.output_mean.0.bias": t(N),
           "ATAC.output_mean.0.weight": t(n_peaks, 128), "ATAC.output_mean.0.bias": t(n_peaks)}
    batches = [(torch.from_numpy(O.synthetic_cells(81 + i, bsz, N)).to(DEV),
                torch.from_numpy((rng.random((bsz, n_peaks)) < 0.05).astype(np.float32)).to(DEV),
                t(bsz, 128, sc=1.0).to(DEV)) for i in range(4)]
    keep = torch.nn.functional.dropout(torch.ones((bsz, 256), device=DEV), 0.1, True)
    monkeypatch.setattr(torch.nn.functional, "dropout", lambda x, p=0.5, training=True, inplace=False: x * keep)
    results = []
    for fused in (False, True):
        cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
        m = S.scMomer(cfg, atac_config=None, encoder=S.StudentEncoder(N, 128, dropout=0.1),
                      rna_decoder=s3.ReconstructNet(N, 128), atac_decoder=s3.ReconstructNet(n_peaks, 128))
        m.load_state_dict({**p, **dec}, strict=False)
        m = m.to(DEV)
        s3.freeze_for_distillation(m)
        m.train()
        m.rna_model.model.performer.fix_projection_matrices_()
        step = s3.S3Step(m, lr=1e-3, grad_acc=2, pos_weight=2.0, distill_weight=0.5, fused=fused)
        assert (step._tail_accs is not None) == fused
        start = {k: v.detach().clone() for k, v in m.named_parameters()}
        losses = [{k: float(v) for k, v in step.micro_step(*b).items()} for b in batches]
        results.append((losses, start, {k: v.detach().clone() for k, v in m.named_parameters()}))
        assert (step._adam_table is not None) == fused        # the second optimizer step ran on the kernel
    (l0, start, w0), (l1, _, w1) = results
    for a, b in zip(l0, l1):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * abs(a[k]), (k, a[k], b[k])
    worst = 0.0
    for k in w0:
        moved = float((w0[k] - start[k]).abs().max())
        if moved == 0:
            assert torch.equal(w0[k], w1[k]), k
            continue
        e = float((w0[k] - w1[k]).abs().max()) / moved
        worst = max(worst, e)
        assert e <= 5e-2, f"{k}: {e:.3e}"
    print(f"S3 on the tail kernels vs torch tail after 2 optimizer steps: worst weight difference {worst:.2e} of the move")


@pytest.mark.parametrize("use_graph", [False, True])
def test_s3_window_of_micro_steps_equals_the_micro_steps(S, monkeypatch, use_graph):
    """window_step over two stacked micro-batches (one launch chain: the frozen encoder once over six cells with a
    key stabiliser per three, the tail kernels once with a set of loss means per micro-batch) against two micro_steps:
    the losses per micro-batch of the first window bit for bit (same weights, same per-cell arithmetic), the later
    ones and the weights after two optimizer steps to rounding (a window's gradients sum over its six cells in one
    go), the redraw counter exactly."""
    from scmomer_amd import s3
    n_peaks, bsz = 1000, 3
    p = O.make_scmomer_params(6)
    rng = np.random.default_rng(90)
    t = lambda *s_, sc=0.05: torch.from_numpy((rng.standard_normal(s_) * sc).astype(np.float32))
    dec = {"RNA.output_mean.0.weight": t(N, 128), "RNA.output_mean.0.bias": t(N),
           "ATAC.output_mean.0.weight": t(n_peaks, 128), "ATAC.output_mean.0.bias": t(n_peaks)}
    batches = [(torch.from_numpy(O.synthetic_cells(91 + i, bsz, N)).to(DEV),
                torch.from_numpy((rng.random((bsz, n_peaks)) < 0.05).astype(np.float32)).to(DEV),
                t(bsz, 128, sc=1.0).to(DEV)) for i in range(4)]
    keep = torch.nn.functional.dropout(torch.ones((bsz, 256), device=DEV), 0.1, True)
    monkeypatch.setattr(torch.nn.functional, "dropout",
                        lambda x, p=0.5, training=True, inplace=False: x * keep.repeat(x.shape[0] // bsz, 1))
    results = []
    for window in (1, 2):
        cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
        m = S.scMomer(cfg, atac_config=None, encoder=S.StudentEncoder(N, 128, dropout=0.1),
                      rna_decoder=s3.ReconstructNet(N, 128), atac_decoder=s3.ReconstructNet(n_peaks, 128))
        m.load_state_dict({**p, **dec}, strict=False)
        m = m.to(DEV)
        s3.freeze_for_distillation(m)
        m.train()
        m.rna_model.model.performer.fix_projection_matrices_()
        step = s3.S3Step(m, lr=1e-3, grad_acc=2, pos_weight=2.0, distill_weight=0.5, use_graph=use_graph)
        start = {k: v.detach().clone() for k, v in m.named_parameters()}
        if window == 1:
            outs = [step.micro_step(*b) for b in batches]
            losses = torch.stack([torch.stack([o[k] for k in s3.LOSS_NAMES]) for o in outs])
        else:
            stacked = [tuple(torch.cat([batches[i][j], batches[i + 1][j]]) for j in range(3)) for i in (0, 2)]
            outs = [step.window_step(*b, 2) for b in stacked]
            losses = torch.cat([torch.stack([o[k] for k in s3.LOSS_NAMES], dim=1) for o in outs])
        weights = {k: v.detach().clone() for k, v in m.named_parameters()}
        if window == 2:
            step.micro_step(*batches[0])
            with pytest.raises(ValueError):
                step.window_step(*stacked[0], 2)            # would straddle the optimizer step
        results.append((losses.clone(), start, weights, int(m.rna_model.model.performer.calls_since_last_redraw)))
    (l0, start, w0, c0), (l1, _, w1, c1) = results
    assert torch.equal(l0[:2], l1[:2]), (l0, l1)
    assert torch.allclose(l0[2:], l1[2:], rtol=1e-5, atol=0), (l0, l1)
    assert c0 == 4 and c1 == 5                      # the refused window never ran; the micro_step before it did
    for k in w0:
        moved = float((w0[k] - start[k]).abs().max())
        assert float((w0[k] - w1[k]).abs().max()) <= 1e-2 * moved, k     # Adam amplifies rounding where a gradient is tiny
