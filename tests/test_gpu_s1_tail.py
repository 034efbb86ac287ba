"""GPU parity of the S1 step tail through the C ABI: the one-kernel masking against the reference's outputs
(golden fixture), against the oracle and against the torch formulation on the same draws (bit-exact), and the
fused cross entropy against torch (f32 tolerance 1e-5 relative on the loss, 1e-5 absolute on the gradient)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def s1():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from scmomer_amd import s1 as m
    return m


def test_mask_matches_reference_fixture(s1, golden_dir):
    g = np.load(os.path.join(golden_dir, "mask_bin.npz"))
    tok = torch.from_numpy(g["s1_bins"]).to(DEV)
    mi, lab = s1.data_mask_device(tok, 0.15, 0.9, 6, 6, rand_pick=torch.from_numpy(g["rand_pick"]).to(DEV),
                                  rand_replace=torch.from_numpy(g["rand_replace"]).to(DEV))
    assert torch.equal(mi.cpu(), torch.from_numpy(g["masked_input"]))
    assert torch.equal(lab.cpu(), torch.from_numpy(g["labels"]))


@pytest.mark.parametrize("seed,b,n", [(0, 3, 16906), (1, 2, 1000), (2, 1, 33), (3, 4, 1025), (4, 3, 7)])
def test_mask_matches_oracle(s1, seed, b, n):
    from oracle import masking_oracle as MO
    rng = np.random.default_rng(seed)
    cells = ((rng.random((b, n)) < 0.15) * rng.gamma(2.0, 1.2, (b, n))).astype(np.float32)
    if b > 1:
        cells[1] = 0.0                                   # a cell with nothing to mask
    if seed == 3:
        cells[0] = 3.0                                   # every position maskable: the cap ceil(prob*N) binds
    tok = MO.bin_s1_tensor(torch.from_numpy(cells))
    rp = rng.random((b, n), dtype=np.float32)
    rr = rng.random((b, n), dtype=np.float32)
    mi, lab = s1.data_mask_device(tok.to(DEV), 0.15, 0.9, 6, 6, rand_pick=torch.from_numpy(rp).to(DEV),
                                  rand_replace=torch.from_numpy(rr).to(DEV))
    omi, olab = MO.data_mask(tok.numpy(), 0.15, 0.9, 6, 6, rp, rr)
    assert np.array_equal(mi.cpu().numpy(), omi) and np.array_equal(lab.cpu().numpy(), olab)


def test_mask_with_tied_draws_keeps_the_count(s1):
    """many equal draws around the k-th value: exactly k positions are picked, none with a smaller draw than an
    unpicked maskable one, lowest positions first among the ties"""
    n = 4096
    tok = torch.full((2, n), 2, dtype=torch.int64)
    tok[:, ::7] = 0
    rp = torch.full((2, n), 0.5)
    rp[0, :100] = 0.9
    rp[1, 1000:1100] = 0.25
    rr = torch.zeros((2, n))
    mi, lab = s1.data_mask_device(tok.to(DEV), 0.15, 0.9, 6, 6, rand_pick=rp.to(DEV), rand_replace=rr.to(DEV))
    lab = lab.cpu()
    for r in range(2):
        maskable = tok[r] != 0
        k = int(torch.ceil(maskable.sum().float() * 0.15))
        picked = lab[r] != 6
        assert int(picked.sum()) == k
        assert bool(maskable[picked].all())
        assert float(rp[r][picked].min()) >= float(rp[r][maskable & ~picked].max())
        ties = (rp[r] == 0.5) & maskable
        idx_p = torch.nonzero(ties & picked).flatten()
        idx_u = torch.nonzero(ties & ~picked).flatten()
        if len(idx_p) and len(idx_u):
            assert int(idx_p.max()) < int(idx_u.min())
    assert bool((mi.cpu()[lab != 6] == 6).all())        # replace draw 0 < 0.9: every picked token replaced


@pytest.mark.parametrize("b,n,c,frac", [(3, 16906, 7, 0.15), (2, 300, 7, 0.5), (1, 17, 2, 1.0), (2, 64, 8, 0.0)])
def test_cross_entropy_matches_torch(s1, b, n, c, frac):
    g = torch.Generator().manual_seed(b * n + c)
    logits = (torch.randn((b, n, c), generator=g) * 3).to(DEV).requires_grad_()
    labels = torch.randint(0, c, (b, n), generator=g)
    ignore = c - 1
    drop = torch.rand((b, n), generator=g) >= frac
    labels[drop] = ignore
    labels = labels.to(DEV)
    loss = s1.masked_cross_entropy(logits, labels, ignore)
    (loss / 5).backward()
    ref_logits = logits.detach().clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_logits.transpose(1, 2), labels, ignore_index=ignore)
    (ref / 5).backward()
    if frac == 0.0 or bool((labels == ignore).all()):
        assert torch.isnan(loss) and torch.isnan(ref)
        return
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert float((logits.grad - ref_logits.grad).abs().max()) <= 1e-5 * float(ref_logits.grad.abs().max()) + 1e-9


@pytest.mark.parametrize("b,n,c", [(3, 16906, 7), (2, 100, 7), (1, 1, 3), (4, 513, 12)])
def test_accuracy_matches_the_loop_definition(b, n, c):
    """prediction = 1 + argmax(softmax(logits)[..., 1:-1]) counted where label != pad (bit-exact counts)"""
    from scmomer_amd import ops
    g = torch.Generator().manual_seed(b * n + c)
    logits = torch.randn((b, n, c), generator=g)
    logits[0, 0, 1:3] = 0.25                                  # a tie: the first maximum wins
    labels = torch.randint(1, c - 1, (b, n), generator=g)
    pad = c - 1
    labels[torch.rand((b, n), generator=g) > 0.2] = pad
    if b > 1:
        labels[1] = pad                                       # a cell with nothing masked: ratio is nan
    ratio, counts = ops.s1_accuracy(logits.to(DEV), labels.to(DEV), pad)
    final = logits[..., 1:-1].argmax(dim=-1) + 1
    keep = labels != pad
    want_n = keep.sum(-1)
    want_c = (keep * (final == labels)).sum(-1)
    assert torch.equal(counts.cpu()[:, 0].long(), want_c) and torch.equal(counts.cpu()[:, 1].long(), want_n)
    torch.testing.assert_close(ratio.cpu(), torch.true_divide(want_c, want_n), equal_nan=True, rtol=1e-6, atol=0)


def test_mask_rejects_cells_longer_than_the_register_budget(s1):
    from scmomer_amd._lib import ScmError
    tok = torch.ones((1, 20481), dtype=torch.int64, device=DEV)
    with pytest.raises(ScmError):
        s1.data_mask_device(tok, 0.15, 0.9, 6, 6)
    mi, lab = s1.data_mask_device(tok[:, :20480], 0.15, 0.9, 6, 6)   # the largest supported length
    assert int((lab != 6).sum()) == int(np.ceil(np.float32(20480) * np.float32(0.15)))
