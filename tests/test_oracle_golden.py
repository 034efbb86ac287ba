"""Pins the CPU oracle against golden vectors produced by the reference itself
(tools/make_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import masking_oracle as MO
from oracle import performer_oracle as O

# float32 oracle vs float32 reference: same math, different op order -> 1e-5 relative
RTOL, ATOL = 2e-4, 1e-6


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max()
    assert err <= atol + rtol * scale, f"max abs err {err:.3e} vs scale {scale:.3e}"


def test_feature_map_and_linear_attention(golden_dir):
    g = load(golden_dir, "feature_map.npz")
    data, v, _, _ = O.attention_inputs(int(g["seed"]), tuple(g["shape"]), float(g["scale"]))
    proj = torch.from_numpy(g["proj"])
    qf = O.softmax_kernel(data, proj, True)
    kf = O.softmax_kernel(data, proj, False)
    np.testing.assert_allclose(qf.numpy(), g["q_features"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(kf.numpy(), g["k_features"], rtol=1e-4, atol=1e-9)
    la = O.linear_attention(qf, kf, v)
    np.testing.assert_allclose(la.numpy(), g["linear_attention"], rtol=1e-3, atol=1e-5)


def test_fast_attention_forward_backward(golden_dir):
    g = load(golden_dir, "fast_attention.npz")
    q, k, v, cot = O.attention_inputs(int(g["seed"]), tuple(g["shape"]), float(g["scale"]))
    proj = torch.from_numpy(g["proj"])
    for dt, rtol in ((torch.float32, 2e-4), (torch.float64, 2e-4)):
        qq, kk, vv = (t.to(dt).clone().detach().requires_grad_() for t in (q, k, v))
        o = O.favor_attention(qq, kk, vv, proj.to(dt))
        (o * cot.to(dt)).sum().backward()
        close(o.detach().numpy(), g["out"], rtol)
        close(qq.grad.numpy(), g["dq"], rtol)
        close(kk.grad.numpy(), g["dk"], rtol)
        close(vv.grad.numpy(), g["dv"], rtol)


@pytest.mark.parametrize("name,n,depth,head", [("lm_linear_n131_d2", 131, 2, "linear"),
                                               ("lm_conv_n300_d1", 300, 1, "conv")])
def test_performer_lm(golden_dir, name, n, depth, head):
    g = load(golden_dir, name + ".npz")
    p = O.make_params(int(g["seed"]), depth=depth, max_seq_len=n, head=head)
    for k_ in list(p):
        if k_.endswith("projection_matrix"):
            np.testing.assert_allclose(p[k_].numpy(), g["param:" + k_], rtol=1e-4, atol=1e-5)
            p[k_] = torch.from_numpy(g["param:" + k_])
    leaves = {}
    for k_, t in p.items():
        if t.dtype == torch.float32 and not k_.endswith("projection_matrix"):
            leaves[k_] = t.clone().requires_grad_()
    pp = dict(p)
    pp.update(leaves)
    tokens = torch.from_numpy(g["tokens"])
    enc = O.performer_lm_forward(tokens, pp, depth=depth, heads=10, head=head, return_encodings=True)
    out = O.performer_lm_forward(tokens, pp, depth=depth, heads=10, head=head)
    close(enc.detach().numpy(), g["encodings"])
    close(out.detach().numpy(), g["out"])
    if head == "linear":
        loss = O.masked_ce_loss(out, torch.from_numpy(g["labels"]), 6)
    else:
        loss = (out * torch.from_numpy(g["cot"])).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    checked = 0
    for k_, leaf in leaves.items():
        grad = leaf.grad.numpy().reshape(-1)
        if f"grad:{k_}:full" in g:
            close(grad, g[f"grad:{k_}:full"], 5e-4)
        elif f"grad:{k_}:sample" in g:
            step = int(g[f"grad:{k_}:stride"])
            close(grad[::step][:2048], g[f"grad:{k_}:sample"], 5e-4)
            l2 = float(np.sqrt((grad.astype(np.float64) ** 2).sum()))
            assert abs(l2 - float(g[f"grad:{k_}:l2"])) <= 5e-4 * float(g[f"grad:{k_}:l2"])
        else:
            continue
        checked += 1
    assert checked >= 10 * depth + 4


def test_mask_and_bins_bit_exact(golden_dir):
    g = load(golden_dir, "mask_bin.npz")
    cells = g["cells"]
    s1 = np.stack([MO.bin_s1(r) for r in cells])
    assert np.array_equal(s1, g["s1_bins"])
    assert np.array_equal(MO.bin_scmomer(cells), g["scmomer_bins"])
    assert list(g["s1_bins"][0, :5]) == [0, 1, 2, 5, 0]
    assert list(g["scmomer_bins"][0, :5]) == [0, 2, 2, 5, 0]
    mi, lab = MO.data_mask(g["s1_bins"], 0.15, 0.9, 6, 6, g["rand_pick"], g["rand_replace"])
    assert np.array_equal(mi, g["masked_input"])
    assert np.array_equal(lab, g["labels"])
    for i in range(cells.shape[0]):
        nnz = int((g["s1_bins"][i] != 0).sum())
        assert int((lab[i] != 6).sum()) == int(np.ceil(np.float32(nnz) * np.float32(0.15)))


def test_projection_matrix_construction(golden_dir):
    g = load(golden_dir, "projection.npz")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    mine = O.gaussian_orthogonal_random_matrix(266, 64, gen).numpy()
    ref = g["proj"]
    # structure: 64-row blocks are orthogonal up to the row scaling
    for m in (mine, ref):
        blk = m[:64] / np.linalg.norm(m[:64], axis=1, keepdims=True)
        np.testing.assert_allclose(blk @ blk.T, np.eye(64), atol=1e-4)
    # same draws under the same default-generator seed (torch.manual_seed) -> same matrix
    torch.manual_seed(int(g["seed"]))
    again = O.gaussian_orthogonal_random_matrix(266, 64, None).numpy()
    np.testing.assert_allclose(again, ref, rtol=1e-3, atol=1e-4)


def test_scmomer_cell_embedding(golden_dir):
    """scMomer.get_cell_embedding of the reference (RNA-only configuration, 16906 genes, depth 6) vs the oracle's
    restatement, same seeded parameters: fp32 against fp32."""
    g = np.load(os.path.join(golden_dir, "scmomer_embed.npz"))
    seed = int(g["seed"])
    p = O.make_scmomer_params(seed)
    for k in g.files:
        if k.startswith("param:"):
            p[k[len("param:"):]] = torch.from_numpy(g[k])
    cells = torch.from_numpy(O.synthetic_cells(seed, 2, 16906))
    torch.set_num_threads(8)
    with torch.no_grad():
        emb = O.scmomer_cell_embedding(cells, p)
    np.testing.assert_allclose(emb.numpy(), g["embedding"], rtol=2e-4, atol=2e-5)


def test_householder_projection_matches_reference(golden_dir):
    """The oracle's own Householder QR (the restatement the device kernel is checked against) reproduces the
    reference's projection matrix from the same Gaussian draws: LAPACK float32 against numpy float64."""
    g = load(golden_dir, "projection.npz")
    torch.manual_seed(int(g["seed"]))
    blocks = torch.stack([torch.randn(64, 64) for _ in range(5)]).numpy()   # the reference's draw order
    mult = torch.randn(266, 64).numpy()
    mine = O.omega_from_draws(blocks, mult, 266)
    np.testing.assert_allclose(mine, g["proj"], rtol=0, atol=1e-4)
    q = O.householder_q(blocks[0])
    np.testing.assert_allclose(q @ q.T, np.eye(64), atol=1e-12)


def test_csr_binning_matches_reference_bins(golden_dir):
    """Sparse rows through the oracle's densify-and-bin restatement give the reference's tokens (both rules)."""
    import scipy.sparse as sp
    g = load(golden_dir, "mask_bin.npz")
    m = sp.csr_matrix(g["cells"])
    rows = list(range(m.shape[0]))
    assert np.array_equal(MO.csr_rows_to_tokens(m.indptr, m.indices, m.data, rows, m.shape[1], 0), g["s1_bins"])
    assert np.array_equal(MO.csr_rows_to_tokens(m.indptr, m.indices, m.data, rows, m.shape[1], 1), g["scmomer_bins"])
