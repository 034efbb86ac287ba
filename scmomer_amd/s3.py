"""The S3 distillation step (s3_pretrain_missing.py:233-316): scMomer in mode='one' with reconstruct=True.  The
Performer RNA encoder is frozen but runs in train() mode (its projection redraw counter ticks, :286 with
performer_pytorch.py:520-534); trainable are the student encoder, to_out and the two linear decoders.  The loss is
BCE-with-logits on the binarised ATAC target + MSE on the RNA reconstruction + distill_weight * MSE between the
student's embedding and the precomputed teacher embedding.  The frozen encoder runs on the kernels without a tape;
the small trainable tail (a few [B, 16906] x [16906, 256] and 128-wide products) is torch library code.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
from torch import nn


class ReconstructNet(nn.Module):
    """utils.py:168-179 (Reconstruct_net): one Linear(hidden_layer -> output_dim) under ``output_mean.0``."""

    def __init__(self, output_dim: int, hidden_layer: int):
        super().__init__()
        self.output_mean = nn.Sequential(nn.Linear(hidden_layer, output_dim, bias=True))

    def forward(self, latent_rep):
        return self.output_mean(latent_rep)


def freeze_for_distillation(model) -> Tuple[int, int]:
    """s3_pretrain_missing.py:233-246: everything frozen except encoder, to_out, ATAC and RNA decoders.
    -> (total, trainable) parameter counts."""
    for p in model.parameters():
        p.requires_grad = False
    for part in (model.encoder, model.to_out, model.ATAC, model.RNA):
        for p in part.parameters():
            p.requires_grad = True
    total = sum(p.numel() for p in model.parameters())
    return total, sum(p.numel() for p in model.parameters() if p.requires_grad)


def distillation_losses(model, rna: torch.Tensor, atac_raw: torch.Tensor, latent_real: torch.Tensor,
                        pos_weight: torch.Tensor, distill_weight: float) -> Dict[str, torch.Tensor]:
    """One forward of s3_pretrain_missing.py:291-307 -> {'atac', 'rna', 'distill', 'total'} (total is the sum the
    script backpropagates, before the division by grad_acc)."""
    atac_target = (atac_raw > 0).float()
    r_atac, r_rna, atac_embeds = model(rna_values=rna, reconstruct=True, mode='one')
    loss_atac = nn.functional.binary_cross_entropy_with_logits(r_atac, atac_target, pos_weight=pos_weight)
    loss_rna = nn.functional.mse_loss(r_rna, rna)
    loss_distill = nn.functional.mse_loss(atac_embeds, latent_real)
    return {"atac": loss_atac, "rna": loss_rna, "distill": loss_distill,
            "total": loss_atac + loss_rna + distill_weight * loss_distill}


class S3Step:
    """Optimizer-step bookkeeping of the S3 loop (:296-316): loss / grad_acc, backward on every micro-step; every
    grad_acc-th one averages the gradients over the ranks (one flat all-reduce instead of DDP's hooks), clips to
    1e4 and applies Adam (all parameters are handed to Adam, as in the script; frozen ones never get a gradient)."""

    def __init__(self, model, *, lr: float = 1e-4, grad_acc: int = 1, pos_weight: float = 1.0,
                 distill_weight: float = 1.0, group=None):
        from . import dist as sdist
        self.model = model
        self.grad_acc = int(grad_acc)
        self.distill_weight = distill_weight
        dev = next(model.parameters()).device
        self.pos_weight = torch.tensor([pos_weight], device=dev)
        self.optimizer = torch.optim.Adam(list(model.parameters()), lr=lr)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.averager = sdist.GradientAverager(self.params, group) if self.params else None
        sdist.broadcast_parameters(list(model.parameters()) + list(model.buffers()), src=0, group=group)
        performer = getattr(getattr(getattr(model, "rna_model", None), "model", None), "performer", None)
        if performer is not None and self.averager is not None and self.averager.world > 1:
            performer.redraw_sync = lambda perf: sdist.broadcast_projections(perf, src=0, group=group)
        self.index = 0

    def micro_step(self, rna, atac_raw, latent_real) -> Dict[str, torch.Tensor]:
        self.index += 1
        losses = distillation_losses(self.model, rna, atac_raw, latent_real, self.pos_weight, self.distill_weight)
        (losses["total"] / self.grad_acc).backward()
        if self.index % self.grad_acc == 0:
            if self.averager is not None:
                self.averager.average_()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), int(1e4))
            self.optimizer.step()
            self.optimizer.zero_grad()
        return {k: v.detach() for k, v in losses.items()}
