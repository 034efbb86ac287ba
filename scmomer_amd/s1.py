"""Host-side pieces of the S1 masked-autoencoding step and of cell binning, with the reference's
semantics (s1_pretrain_rna.py:88-137, scmomer.py:56,108).  Index/integer work: results are
bit-identical to the reference given the same random draws.  Runs on whatever device the input
tensors live on (torch ops; this is harness logic around the encoder, not the encoder).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


def bin_s1(rows: torch.Tensor, class_limit: int = 7) -> torch.Tensor:
    """SCDataset.__getitem__ (s1_pretrain_rna.py:133-137): values above class_limit-2 are set to it,
    then truncated toward zero to int64."""
    r = rows.to(torch.float32).clone()
    r[r > (class_limit - 2)] = class_limit - 2
    return r.long()


def bin_scmomer(x: torch.Tensor) -> torch.Tensor:
    """scMomer.forward / get_cell_embedding (scmomer.py:56,108): round half-to-even, clamp to [0,5]."""
    return torch.round(x).long().clamp(min=0, max=5)


def mask_with_tokens(t: torch.Tensor, token_ids) -> torch.Tensor:
    mask = torch.zeros_like(t, dtype=torch.bool)
    for el in token_ids:
        mask |= (t == el)
    return mask


def get_mask_subset_with_prob(mask: torch.Tensor, prob: float, rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """s1_pretrain_rna.py:98-110: per row, the ceil(count*prob) maskable positions with the largest
    uniform draws (at most ceil(prob*seq_len))."""
    batch, seq_len = mask.shape
    device = mask.device
    max_masked = math.ceil(prob * seq_len)
    num_tokens = mask.sum(dim=-1, keepdim=True)
    positions = torch.arange(seq_len, device=device, dtype=torch.float32).unsqueeze(0).expand(batch, seq_len)
    mask_excess = positions >= (num_tokens * prob).ceil()
    mask_excess = mask_excess[:, :max_masked]
    if rand is None:
        rand = torch.rand((batch, seq_len), device=device)
    rand = rand.masked_fill(~mask, -1e9)
    _, sampled_indices = rand.topk(max_masked, dim=-1)
    sampled_indices = (sampled_indices + 1).masked_fill_(mask_excess, 0)
    new_mask = torch.zeros((batch, seq_len + 1), device=device)
    new_mask.scatter_(-1, sampled_indices, 1)
    return new_mask[:, 1:].bool()


def data_mask(data: torch.Tensor, mask_prob: float, replace_prob: float, pad_token_id: int, mask_token_id: int,
              rand_pick: Optional[torch.Tensor] = None,
              rand_replace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """s1_pretrain_rna.py:113-121 -> (masked_input, labels).  Tokens 0 and pad are never masked;
    90 % of the picked positions become the mask token; labels hold the original token at picked
    positions and pad elsewhere.  The two draws happen in the reference's order."""
    no_mask = mask_with_tokens(data, {0, pad_token_id})
    mask = get_mask_subset_with_prob(~no_mask, mask_prob, rand_pick)
    if rand_replace is None:
        rand_replace = torch.zeros_like(data).float().uniform_(0, 1)
    replace = rand_replace < replace_prob
    masked_input = data.clone().detach().masked_fill(mask * replace, mask_token_id)
    labels = data.masked_fill(~mask, pad_token_id)
    return masked_input, labels


def data_mask_device(data: torch.Tensor, mask_prob: float, replace_prob: float, pad_token_id: int, mask_token_id: int,
                     rand_pick: Optional[torch.Tensor] = None,
                     rand_replace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """data_mask as ONE kernel (a radix select per cell instead of a 16,906-wide topk and a dozen elementwise
    passes).  Same draws, same order and same result as data_mask; draws that tie exactly at the k-th value are
    resolved lowest position first (torch.topk leaves that order unspecified)."""
    from . import ops
    if rand_pick is None:
        rand_pick = torch.rand(data.shape, device=data.device)
    if rand_replace is None:
        rand_replace = torch.zeros_like(data).float().uniform_(0, 1)
    return ops.s1_mask(data.contiguous(), rand_pick.contiguous(), rand_replace.contiguous(), mask_prob, replace_prob,
                       pad_token_id, mask_token_id)


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        from . import ops
        lg = logits.contiguous()
        lb = labels.contiguous()
        out = ops.ce_fwd(lg, lb, ignore_index)
        ctx.save_for_backward(lg, lb, out)
        ctx.ignore_index = ignore_index
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        from . import ops
        lg, lb, out = ctx.saved_tensors
        d = ops.ce_bwd(lg, lb, ctx.ignore_index, dloss.to(torch.float32).contiguous(), out)
        return d, None, None


def masked_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int) -> torch.Tensor:
    """CrossEntropyLoss(ignore_index=..., reduction='mean')(logits.transpose(1, 2), labels) of the S1 loop
    (s1_pretrain_rna.py:268,297) on the un-transposed f32 logits [B, N, C<=8]: two small kernels forward, one
    backward."""
    return _MaskedCE.apply(logits, labels, ignore_index)


def synthetic_tokens(seed: int, batch: int, n: int, device=None) -> torch.Tensor:
    """Synthetic S1 cells: 15 % expressed genes with gamma(2, 1.2) log-expression, S1 binning."""
    import numpy as np
    rng = np.random.default_rng(seed)
    nz = rng.random((batch, n)) < 0.15
    val = rng.gamma(2.0, 1.2, (batch, n))
    cells = torch.from_numpy((nz * val).astype(np.float32))
    tok = bin_s1(cells)
    return tok.to(device) if device is not None else tok
