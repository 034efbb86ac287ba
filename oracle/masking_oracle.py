"""CPU restatement (numpy, integer/bit-exact) of the S1 binning and masking rules.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows s1_pretrain_rna.py:88-137 and scmomer.py:56,108.  The random draws are
inputs: ``rand_pick`` is the ``torch.rand((batch, seq_len))`` of
``get_mask_subset_with_prob`` (:105) and ``rand_replace`` the ``uniform_(0,1)`` of
``prob_mask_like`` (:89), in that call order (:116 then :118).
"""
from __future__ import annotations

import math

import numpy as np


def bin_s1(row: np.ndarray, class_limit: int = 7) -> np.ndarray:
    """SCDataset.__getitem__, s1_pretrain_rna.py:133-137: clamp values above
    class_limit-2 to it, then truncate toward zero to int64."""
    r = np.array(row, dtype=np.float32, copy=True)
    r[r > (class_limit - 2)] = class_limit - 2
    return np.trunc(r).astype(np.int64)


def bin_scmomer(x: np.ndarray) -> np.ndarray:
    """scMomer.forward / get_cell_embedding, scmomer.py:56,108:
    round half-to-even, to int64, clamp to [0, 5]."""
    return np.clip(np.rint(np.asarray(x, dtype=np.float32)).astype(np.int64), 0, 5)


def csr_rows_to_tokens(indptr, indices, values, rows, n_genes: int, rule: int, limit: int = 5) -> np.ndarray:
    """Sparse rows -> dense float32 rows (duplicate column entries add up, as scipy's ``toarray()`` of
    s1_pretrain_rna.py:134 does) -> tokens.  rule 0: the S1 rule (:135-136) with ``limit = class_limit - 2``;
    rule 1: the scMomer rule (scmomer.py:56,108) with clamp to [0, limit]."""
    indptr = np.asarray(indptr, dtype=np.int64)
    out = np.zeros((len(rows), n_genes), dtype=np.int64)
    for b, r in enumerate(rows):
        dense = np.zeros(n_genes, dtype=np.float32)
        lo, hi = int(indptr[r]), int(indptr[r + 1])
        np.add.at(dense, np.asarray(indices[lo:hi], dtype=np.int64), np.asarray(values[lo:hi], dtype=np.float32))
        if rule == 0:
            out[b] = bin_s1(dense, limit + 2)
        else:
            out[b] = np.clip(np.rint(dense).astype(np.int64), 0, limit)
    return out


def mask_subset_with_prob(maskable: np.ndarray, prob: float, rand_pick: np.ndarray) -> np.ndarray:
    """get_mask_subset_with_prob, s1_pretrain_rna.py:98-110.

    Per row: the ceil(float32(count)*float32(prob)) maskable positions with the largest
    draws (at most ceil(prob*seq_len) of them) are selected.
    """
    b, n = maskable.shape
    max_masked = math.ceil(prob * n)
    out = np.zeros((b, n), dtype=bool)
    for i in range(b):
        cnt = int(maskable[i].sum())
        k = int(np.ceil(np.float32(cnt) * np.float32(prob)))
        k = min(k, max_masked)
        r = np.where(maskable[i], rand_pick[i].astype(np.float32), np.float32(-1e9))
        order = np.argsort(-r, kind="stable")  # descending, first index wins ties
        out[i, order[:k]] = True
    return out


def data_mask(data: np.ndarray, mask_prob: float, replace_prob: float, pad_token_id: int,
              mask_token_id: int, rand_pick: np.ndarray, rand_replace: np.ndarray):
    """data_mask, s1_pretrain_rna.py:113-121 -> (masked_input, labels), int64."""
    data = np.asarray(data, dtype=np.int64)
    no_mask = (data == 0) | (data == pad_token_id)
    mask = mask_subset_with_prob(~no_mask, mask_prob, rand_pick)
    replace = rand_replace.astype(np.float32) < np.float32(replace_prob)
    masked_input = np.where(mask & replace, np.int64(mask_token_id), data)
    labels = np.where(mask, data, np.int64(pad_token_id))
    return masked_input, labels


# ---------------------------------------------------------------------------
# torch-tensor front ends (what CPU tests of the training driver inject as mask_fn)
# ---------------------------------------------------------------------------
def bin_s1_tensor(rows, class_limit: int = 7):
    import torch
    return torch.from_numpy(bin_s1(rows.detach().cpu().numpy(), class_limit)).to(rows.device)


def bin_scmomer_tensor(x):
    import torch
    return torch.from_numpy(bin_scmomer(x.detach().cpu().numpy())).to(x.device)


def data_mask_tensor(data, mask_prob: float, replace_prob: float, pad_token_id: int, mask_token_id: int,
                     rand_pick=None, rand_replace=None):
    """data_mask on torch tensors.  Missing draws are made with torch on data's device in the reference's order
    (s1_pretrain_rna.py:105 then :89,118), so a seeded torch generator reproduces the reference's masks."""
    import torch
    if rand_pick is None:
        rand_pick = torch.rand(data.shape, device=data.device)
    if rand_replace is None:
        rand_replace = torch.zeros_like(data).float().uniform_(0, 1)
    mi, lab = data_mask(data.detach().cpu().numpy(), mask_prob, replace_prob, pad_token_id, mask_token_id,
                        rand_pick.detach().cpu().numpy(), rand_replace.detach().cpu().numpy())
    return torch.from_numpy(mi).to(data.device), torch.from_numpy(lab).to(data.device)
