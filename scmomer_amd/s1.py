"""The S1 masked-autoencoding step's tail on the device, with the reference's semantics
(s1_pretrain_rna.py:88-121,268,297): masking as one kernel (bit-identical to the reference given the same random
draws) and the masked cross entropy as fused kernels.  Binning lives in ingest.py (kernels bin_csr / bin_dense).
There is no host formulation of masking or binning in this package.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def data_mask_device(data: torch.Tensor, mask_prob: float, replace_prob: float, pad_token_id: int, mask_token_id: int,
                     rand_pick: Optional[torch.Tensor] = None,
                     rand_replace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """data_mask (s1_pretrain_rna.py:113-121 with :88-110) as ONE kernel: a radix select per cell instead of a
    16,906-wide topk and a dozen elementwise passes.  Tokens 0 and pad are never masked; per cell the
    ceil(count * mask_prob) maskable positions with the largest draws are picked; those with a replace draw below
    replace_prob become the mask token; labels hold the original token at picked positions and pad elsewhere.  The
    two draws happen in the reference's order; draws that tie exactly at the k-th value are resolved lowest
    position first (torch.topk leaves that order unspecified)."""
    from . import ops
    if rand_pick is None:
        rand_pick = torch.rand(data.shape, device=data.device)
    if rand_replace is None:
        # the reference draws into torch.zeros_like(data).float() (s1_pretrain_rna.py:118): uniform_ overwrites every
        # element from the generator alone, so an uninitialised f32 buffer gives the same draws without the fill
        # and the conversion kernels
        rand_replace = torch.empty(data.shape, dtype=torch.float32, device=data.device).uniform_(0, 1)
    return ops.s1_mask(data.contiguous(), rand_pick.contiguous(), rand_replace.contiguous(), mask_prob, replace_prob,
                       pad_token_id, mask_token_id)


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, group_tokens):
        from . import ops
        lg = logits.contiguous()
        lb = labels.contiguous()
        out = ops.ce_fwd(lg, lb, ignore_index, group_tokens)
        ctx.save_for_backward(lg, lb, out)
        ctx.ignore_index, ctx.group_tokens = ignore_index, group_tokens
        return out[:, 0] if group_tokens else out[0, 0]

    @staticmethod
    def backward(ctx, dloss):
        from . import ops
        lg, lb, out = ctx.saved_tensors
        d = ops.ce_bwd(lg, lb, ctx.ignore_index, dloss.to(torch.float32).reshape(-1).contiguous(), out, ctx.group_tokens)
        return d, None, None, None


def masked_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int,
                         group_cells: int = 0) -> torch.Tensor:
    """CrossEntropyLoss(ignore_index=..., reduction='mean')(logits.transpose(1, 2), labels) of the S1 loop
    (s1_pretrain_rna.py:268,297) on the un-transposed f32 logits [B, N, C<=8]: two small kernels forward, one
    backward.  group_cells > 0: the batch holds several micro-batches of group_cells cells each (one accumulation
    window run as one launch chain); the result is the vector of their losses, one mean per micro-batch."""
    return _MaskedCE.apply(logits, labels, ignore_index, group_cells * logits.shape[-2] if group_cells else 0)


def synthetic_tokens(seed: int, batch: int, n: int, device=None) -> torch.Tensor:
    """Synthetic S1 cells: 15 % expressed genes with gamma(2, 1.2) log-expression, binned with the S1 rule
    (s1_pretrain_rna.py:135-136: values above 5 become 5, then truncation)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    nz = rng.random((batch, n)) < 0.15
    val = rng.gamma(2.0, 1.2, (batch, n))
    tok = torch.from_numpy(np.minimum((nz * val).astype(np.float32), np.float32(5)).astype(np.int64))
    return tok.to(device) if device is not None else tok
