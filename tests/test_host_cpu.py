"""CPU checks of the host-side mirror: S1 binning/masking bit-exact against the reference's outputs (golden
fixture) and against the oracle, module/state_dict layout of PerformerLM, loud failure without a GPU, the
projection-matrix construction, and the argument routing helpers."""
import os

import numpy as np
import pytest
import torch

import scmomer_amd as S
from oracle import masking_oracle as MO


def test_bins_and_mask_bit_exact_vs_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "mask_bin.npz"))
    cells = torch.from_numpy(g["cells"])
    assert torch.equal(MO.bin_s1_tensor(cells), torch.from_numpy(g["s1_bins"]))
    assert torch.equal(MO.bin_scmomer_tensor(cells), torch.from_numpy(g["scmomer_bins"]))
    tok = torch.from_numpy(g["s1_bins"])
    mi, lab = MO.data_mask_tensor(tok, 0.15, 0.9, 6, 6, rand_pick=torch.from_numpy(g["rand_pick"]),
                           rand_replace=torch.from_numpy(g["rand_replace"]))
    assert torch.equal(mi, torch.from_numpy(g["masked_input"]))
    assert torch.equal(lab, torch.from_numpy(g["labels"]))


def test_bin_edge_values():
    x = torch.tensor([[0.5, 1.5, 2.5, 5.6, -0.2, 4.999, 5.0, 7.3, 0.0]])
    assert MO.bin_scmomer_tensor(x).tolist() == [[0, 2, 2, 5, 0, 5, 5, 5, 0]]       # half-to-even, clamp
    assert MO.bin_s1_tensor(x).tolist() == [[0, 1, 2, 5, 0, 4, 5, 5, 0]]            # cap at 5, truncate


@pytest.mark.parametrize("seed,n", [(0, 64), (1, 257), (2, 2000), (3, 16906)])
def test_oracle_masking_formulations_agree_and_keep_the_mask_invariants(seed, n):
    """test infrastructure checking itself: the oracle's two formulations of data_mask (torch ops in the reference's
    order; numpy loops) agree on random rows, and both keep the invariants of s1_pretrain_rna.py:88-121.  No
    product code runs here: the package's masking is the device kernel (tests/test_gpu_s1_tail.py)."""
    rng = np.random.default_rng(seed)
    cells = ((rng.random((3, n)) < 0.15) * rng.gamma(2.0, 1.2, (3, n))).astype(np.float32)
    if seed == 1:
        cells[1] = 0.0                      # a cell with no expressed gene: nothing to mask
    tok = MO.bin_s1_tensor(torch.from_numpy(cells))
    rp = rng.random((3, n), dtype=np.float32)
    rr = rng.random((3, n), dtype=np.float32)
    mi, lab = MO.data_mask_tensor(tok, 0.15, 0.9, 6, 6, rand_pick=torch.from_numpy(rp), rand_replace=torch.from_numpy(rr))
    omi, olab = MO.data_mask(tok.numpy(), 0.15, 0.9, 6, 6, rp, rr)
    assert np.array_equal(mi.numpy(), omi) and np.array_equal(lab.numpy(), olab)
    # size-independent properties: exactly ceil(0.15 * nnz) positions picked per row, never a 0 token
    picked = lab != 6
    nnz = (tok != 0).sum(-1)
    assert torch.equal(picked.sum(-1), torch.ceil(nnz.float() * 0.15).long())
    assert bool((tok[picked] != 0).all())
    assert torch.equal(lab[picked], tok[picked])
    assert bool(((mi == tok) | (mi == 6)).all())


def _lm(**kw):
    args = dict(num_tokens=7, dim=200, depth=2, max_seq_len=300, heads=10, local_attn_heads=0, g2v_position_emb=False)
    args.update(kw)
    return S.PerformerLM(**args)


def test_state_dict_layout_matches_reference_contract():
    m = _lm()
    sd = m.state_dict()
    expect = {"token_emb.weight": (7, 200), "performer.calls_since_last_redraw": (),
              "norm.weight": (200,), "norm.bias": (200,), "to_out.weight": (7, 200), "to_out.bias": (7,)}
    for i in range(2):
        pre = f"performer.net.layers.{i}."
        expect.update({pre + "0.norm.weight": (200,), pre + "0.norm.bias": (200,),
                       pre + "0.fn.fast_attention.projection_matrix": (266, 64),
                       pre + "0.fn.to_q.weight": (640, 200), pre + "0.fn.to_k.weight": (640, 200),
                       pre + "0.fn.to_v.weight": (640, 200), pre + "0.fn.to_out.weight": (200, 640),
                       pre + "0.fn.to_out.bias": (200,), pre + "1.norm.weight": (200,), pre + "1.norm.bias": (200,),
                       pre + "1.fn.fn.w1.weight": (800, 200), pre + "1.fn.fn.w1.bias": (800,),
                       pre + "1.fn.fn.w2.weight": (200, 800), pre + "1.fn.fn.w2.bias": (200,)})
    got = {k: tuple(v.shape) for k, v in sd.items() if not k.startswith("pos_emb")}
    assert got == expect
    assert sd["performer.calls_since_last_redraw"].dtype == torch.int64
    # attribute paths the reference's scripts touch (s1_pretrain_rna.py:225-228, rna_encoder.py:38)
    assert len(m.performer.net.layers) == 2 and len(m.performer.net.layers[-1]) == 2
    assert m.max_seq_len == 300 and isinstance(m.norm, torch.nn.LayerNorm)
    m.to_out = torch.nn.Identity()          # re-assignable head


def test_gene2vec_table_layout(monkeypatch, tmp_path):
    tab = np.random.default_rng(0).standard_normal((299, 200))
    path = tmp_path / "g2v.npy"
    np.save(path, tab)
    monkeypatch.setenv("SCMOMER_GENE2VEC", str(path))
    m = _lm(g2v_position_emb=True)
    w = m.state_dict()["pos_emb.emb.weight"]
    assert w.dtype == torch.float64 and tuple(w.shape) == (300, 200)       # one zero row appended
    assert torch.equal(w[:299], torch.from_numpy(tab)) and float(w[299].abs().max()) == 0.0
    assert not m.pos_emb.emb.weight.requires_grad


def test_forward_without_gpu_fails_loudly():
    m = _lm()
    with pytest.raises((RuntimeError, ValueError, AssertionError)):
        m(torch.zeros((1, 16), dtype=torch.long))
    fa = S.FastAttention(64)
    q = torch.zeros((1, 2, 8, 64))
    with pytest.raises((RuntimeError, ValueError, AssertionError)):
        fa(q, q, q)


def test_sequence_longer_than_max_seq_len_asserts():
    m = _lm(max_seq_len=8)
    with pytest.raises(AssertionError):
        m(torch.zeros((1, 9), dtype=torch.long))


@pytest.mark.parametrize("kw", [dict(causal=True), dict(local_attn_heads=2), dict(reversible=True), dict(ff_glu=True),
                                dict(use_scalenorm=True), dict(use_rezero=True), dict(generalized_attention=True),
                                dict(no_projection=True), dict(cross_attend=True)])
def test_unsupported_options_raise(kw):
    with pytest.raises(NotImplementedError):
        _lm(**kw)


def test_projection_matrix_construction_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "projection.npz"))
    torch.manual_seed(int(g["seed"]))
    p = S.gaussian_orthogonal_random_matrix(266, 64)
    np.testing.assert_allclose(p.numpy(), g["proj"], rtol=0, atol=0)


def test_redraw_bookkeeping():
    m = _lm(feature_redraw_interval=2)
    before = m.performer.net.layers[0][0].fn.fast_attention.projection_matrix.clone()
    m.train()
    m.performer.check_redraw_projections()
    m.performer.check_redraw_projections()
    assert int(m.performer.calls_since_last_redraw) == 2
    m.performer.check_redraw_projections()                                   # interval reached: redraw, counter reset
    after = m.performer.net.layers[0][0].fn.fast_attention.projection_matrix
    assert int(m.performer.calls_since_last_redraw) == 0 and not torch.equal(before, after)
    m.fix_projection_matrices_()
    fixed = after.clone()
    for _ in range(5):
        m.performer.check_redraw_projections()
    assert torch.equal(fixed, m.performer.net.layers[0][0].fn.fast_attention.projection_matrix)


def test_scmomer_wrapper_layout_and_errors():
    """state_dict keys of the RNA-only scMomer wrapper are the reference's (scmomer.py:21-39); the ATAC branch is
    refused loudly"""
    import types
    from oracle import performer_oracle as O
    cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
    m = S.scMomer(cfg, atac_config=None, encoder=S.StudentEncoder(16906, 128))
    want = O.make_scmomer_params(1)
    got = m.state_dict()
    assert set(got) == set(want)
    for k, v in want.items():
        assert tuple(got[k].shape) == tuple(v.shape) and got[k].dtype == v.dtype, k
    with pytest.raises(NotImplementedError):
        S.scMomer(cfg, atac_config=object())
    with pytest.raises(NotImplementedError):
        m(rna_values=torch.zeros(1, 16906), mode="multimodal")
    with pytest.raises(ValueError):
        m(rna_values=torch.zeros(1, 16906), mode="other")
