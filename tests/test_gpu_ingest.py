"""GPU parity of the ingest kernels and of the device-side projection redraw, through the C ABI: CSR rows ->
tokens bit-exact against the reference's bins (golden fixture) and the oracle; Householder QR on the device
against the oracle on the same draws (f32 against f64: 2e-4 absolute on entries of magnitude <= 10) and against
the reference's matrix (golden fixture)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ing():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from scmomer_amd import ingest
    return ingest


def test_csr_bins_match_reference_fixture(ing, golden_dir):
    import scipy.sparse as sp
    g = np.load(os.path.join(golden_dir, "mask_bin.npz"))
    csr = ing.DeviceCSR.from_scipy(sp.csr_matrix(g["cells"]), DEV)
    assert np.array_equal(csr.tokens(None, ing.RULE_S1, 5, batch=csr.n_cells).cpu().numpy(), g["s1_bins"])
    assert np.array_equal(csr.tokens(None, ing.RULE_SCMOMER, 5, batch=csr.n_cells).cpu().numpy(), g["scmomer_bins"])
    dense = torch.from_numpy(g["cells"]).to(DEV)
    assert np.array_equal(ing.bin_cells(dense, ing.RULE_S1).cpu().numpy(), g["s1_bins"])
    assert np.array_equal(ing.bin_cells(dense, ing.RULE_SCMOMER).cpu().numpy(), g["scmomer_bins"])
    csr.raise_on_bad_index()


@pytest.mark.parametrize("seed,cells,genes,batch", [(0, 40, 16906, 3), (1, 7, 1000, 7), (2, 5, 33, 9), (3, 3, 36864, 2)])
def test_csr_bins_match_oracle(ing, seed, cells, genes, batch):
    """gathered, repeated and empty rows; duplicate column entries; negative and half-way values"""
    from oracle import masking_oracle as MO
    rng = np.random.default_rng(seed)
    indptr, indices, values = [0], [], []
    for c in range(cells):
        nnz = 0 if c == 1 else int(rng.integers(1, max(2, genes // 6)))
        cols = rng.integers(0, genes, nnz)                 # with repetitions: duplicates must add up
        vals = rng.gamma(2.0, 1.2, nnz).astype(np.float32)
        vals[::5] = np.round(vals[::5]) + 0.5              # ties of the half-to-even rule
        vals[::11] *= -1.0
        indices += cols.tolist()
        values += vals.tolist()
        indptr.append(len(indices))
    csr = ing.DeviceCSR(indptr, indices, values, genes, DEV)
    rows = rng.integers(0, cells, batch).tolist()
    rows[-1] = 1 if cells > 1 else rows[-1]                # the empty cell
    for rule in (ing.RULE_S1, ing.RULE_SCMOMER):
        got = csr.tokens(rows, rule, 5).cpu().numpy()
        want = MO.csr_rows_to_tokens(indptr, np.array(indices), np.array(values, dtype=np.float32), rows, genes, rule, 5)
        # float sums of three or more duplicates depend on the order: compare where at most two entries collide
        assert got.shape == want.shape
        mism = np.argwhere(got != want)
        for b, col in mism:
            lo, hi = indptr[rows[b]], indptr[rows[b] + 1]
            assert (np.array(indices[lo:hi]) == col).sum() >= 3, (b, col, got[b, col], want[b, col])
    csr.raise_on_bad_index()


def test_csr_rejects_bad_rows_and_sizes(ing):
    from scmomer_amd._lib import ScmError
    csr = ing.DeviceCSR([0, 2, 3], [0, 4, 2], [1.0, 7.0, 2.2], 5, DEV)
    out = csr.tokens([1, 5, 0], ing.RULE_S1, 5).cpu().numpy()
    assert out[0].tolist() == [0, 0, 2, 0, 0] and out[1].tolist() == [0] * 5 and out[2].tolist() == [1, 0, 0, 0, 5]
    with pytest.raises(IndexError):
        csr.raise_on_bad_index()
    csr.raise_on_bad_index()                                # the flag was cleared
    with pytest.raises(ValueError):
        ing.DeviceCSR([0, 1], [9], [1.0], 5, DEV)
    big = ing.DeviceCSR([0, 1], [0], [1.0], 40000, DEV)
    with pytest.raises(ScmError):
        big.tokens([0])


def test_scdataset_interface(ing):
    """SCDataset(data, class_limit): dataset[i] is the token row of cell i, len() the number of cells"""
    import scipy.sparse as sp
    from oracle import masking_oracle as MO
    rng = np.random.default_rng(5)
    cells = ((rng.random((6, 300)) < 0.2) * rng.gamma(2.0, 1.5, (6, 300))).astype(np.float32)
    ds = ing.SCDataset(sp.csr_matrix(cells), 7, device=DEV)
    assert len(ds) == 6
    for i in (0, 3, -1):
        assert np.array_equal(ds[i].cpu().numpy(), MO.bin_s1(cells[i], 7))
    assert np.array_equal(ds.batch([4, 4, 2]).cpu().numpy(), np.stack([MO.bin_s1(cells[j], 7) for j in (4, 4, 2)]))
    with pytest.raises(IndexError):
        ds[6]


def test_device_qr_matches_reference_fixture(golden_dir):
    from scmomer_amd import ops
    from oracle import performer_oracle as O
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    torch.manual_seed(int(g["seed"]))
    blocks = torch.stack([torch.randn(64, 64) for _ in range(5)])
    mult = torch.randn(266, 64)
    got = ops.orthogonal_omega(blocks[None].to(DEV), mult[None].to(DEV), 266, 0)[0].cpu().numpy()
    np.testing.assert_allclose(got, g["proj"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(got, O.omega_from_draws(blocks.numpy(), mult.numpy(), 266), rtol=0, atol=2e-4)


@pytest.mark.parametrize("nmat,nb_rows,scaling", [(6, 266, 0), (2, 64, 1), (3, 10, 0), (1, 130, 1)])
def test_device_qr_matches_oracle(nmat, nb_rows, scaling):
    from scmomer_amd import ops
    from oracle import performer_oracle as O
    rng = np.random.default_rng(nmat * 1000 + nb_rows)
    nblk = (nb_rows + 63) // 64
    blocks = rng.standard_normal((nmat, nblk, 64, 64)).astype(np.float32)
    blocks[0, 0, 1:, 0] = 0.0                                # a column with nothing below the diagonal: H = I
    mult = rng.standard_normal((nmat, nb_rows, 64)).astype(np.float32)
    got = ops.orthogonal_omega(torch.from_numpy(blocks).to(DEV), torch.from_numpy(mult).to(DEV) if scaling == 0 else None,
                               nb_rows, scaling).cpu().numpy()
    for m in range(nmat):
        want = O.omega_from_draws(blocks[m], mult[m], nb_rows, scaling)
        np.testing.assert_allclose(got[m], want, rtol=0, atol=2e-4)


def test_redraw_through_the_module_tree():
    """Performer.check_redraw_projections in train() mode after the interval: every layer gets a fresh matrix
    with orthogonal 64-row blocks, the counter restarts, and the model still runs"""
    from scmomer_amd import PerformerLM
    torch.manual_seed(3)
    m = PerformerLM(num_tokens=7, dim=200, depth=2, max_seq_len=128, heads=10, local_attn_heads=0,
                    g2v_position_emb=False, feature_redraw_interval=2).to(DEV)
    m.train()
    fas = [l[0].fn.fast_attention for l in m.performer.net.layers]
    before = [fa.projection_matrix.clone() for fa in fas]
    tok = torch.randint(0, 6, (2, 128), device=DEV)
    for _ in range(2):
        m(tok)
    assert int(m.performer.calls_since_last_redraw) == 2
    assert all(torch.equal(fa.projection_matrix, b) for fa, b in zip(fas, before))
    out = m(tok)                                             # third call: redraw, counter restarts
    assert int(m.performer.calls_since_last_redraw) == 0
    assert torch.isfinite(out).all()
    mats = [fa.projection_matrix.cpu().numpy() for fa in fas]
    assert not np.allclose(mats[0], mats[1])
    for mat, b in zip(mats, before):
        assert not np.allclose(mat, b.cpu().numpy())
        blk = mat[64:128] / np.linalg.norm(mat[64:128], axis=1, keepdims=True)
        np.testing.assert_allclose(blk @ blk.T, np.eye(64), atol=1e-4)
        norms = np.linalg.norm(mat, axis=1)                  # chi(64): mean ~ 7.97
        assert 7.0 < norms.mean() < 9.0
    from scmomer_amd import gaussian_orthogonal_random_matrix
    one = gaussian_orthogonal_random_matrix(266, 64, scaling=1, device=DEV)
    np.testing.assert_allclose(np.linalg.norm(one.cpu().numpy(), axis=1), 8.0, atol=1e-4)
