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
    1e4 and applies Adam (all parameters are handed to Adam, as in the script; frozen ones never get a gradient).
    Gradients are folded into static accumulators (the slices of the all-reduce bucket), which lets the micro-step
    (forward, losses, backward, accumulate: about 150 launches, a third of them library calls of the small tail) be
    recorded once into a HIP graph and replayed (use_graph=True): driven from Python the step is launch-bound, the
    GPU idle for 40 % of it."""

    def __init__(self, model, *, lr: float = 1e-4, grad_acc: int = 1, pos_weight: float = 1.0,
                 distill_weight: float = 1.0, group=None, use_graph: bool = False):
        from . import dist as sdist
        self.model = model
        self.grad_acc = int(grad_acc)
        self.distill_weight = distill_weight
        dev = next(model.parameters()).device
        self.pos_weight = torch.tensor([pos_weight], device=dev)
        self.optimizer = torch.optim.Adam(list(model.parameters()), lr=lr)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.averager = sdist.GradientAverager(self.params, group) if self.params else None
        self.accs = list(self.averager.views) if self.averager is not None else []
        sdist.broadcast_parameters(list(model.parameters()) + list(model.buffers()), src=0, group=group)
        self.performer = getattr(getattr(getattr(model, "rna_model", None), "model", None), "performer", None)
        if self.performer is not None and self.averager is not None and self.averager.world > 1:
            self.performer.redraw_sync = lambda perf: sdist.broadcast_projections(perf, src=0, group=group)
        self.index = 0
        self.use_graph = use_graph
        self._graph = None
        self._seed = None

    def _forward_backward(self, rna, atac_raw, latent_real) -> Dict[str, torch.Tensor]:
        losses = distillation_losses(self.model, rna, atac_raw, latent_real, self.pos_weight, self.distill_weight)
        if self.params:
            if self._seed is None:
                self._seed = torch.full((), 1.0, dtype=losses["total"].dtype, device=losses["total"].device) / self.grad_acc
            grads = torch.autograd.grad(losses["total"], self.params, grad_outputs=self._seed, allow_unused=True)
            left = [(a, g) for a, g in zip(self.accs, grads) if g is not None]
            if left:
                torch._foreach_add_([a for a, _ in left], [g for _, g in left])
        return {k: v.detach() for k, v in losses.items()}

    def _capture(self, rna, atac_raw, latent_real):
        """Record the micro-step for these shapes once; the projection redraw check stays outside the recording."""
        self._static_in = [rna.clone(), atac_raw.clone(), latent_real.clone()]
        saved = [a.clone() for a in self.accs]
        auto = self.performer.auto_check_redraw if self.performer is not None else None
        calls = self.performer.calls_since_last_redraw.clone() if self.performer is not None else None
        if self.performer is not None:
            self.performer.auto_check_redraw = False
        self._static_out: Dict[str, torch.Tensor] = {}

        def body():
            out = self._forward_backward(*self._static_in)
            for k, v in out.items():
                if k not in self._static_out:
                    self._static_out[k] = torch.zeros_like(v)
                self._static_out[k].copy_(v)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body()
        for a, s_ in zip(self.accs, saved):                  # warm-up and recording passes do not count
            a.copy_(s_)
        if calls is not None:
            self.performer.calls_since_last_redraw.copy_(calls)
        self._graph, self._graph_auto = graph, auto
        self._graph_shapes = tuple(tuple(t.shape) for t in self._static_in)

    def micro_step(self, rna, atac_raw, latent_real) -> Dict[str, torch.Tensor]:
        self.index += 1
        shapes = (tuple(rna.shape), tuple(atac_raw.shape), tuple(latent_real.shape))
        if self.use_graph and rna.is_cuda:
            if self._graph is None:
                self._capture(rna, atac_raw, latent_real)
        if self._graph is not None and shapes == self._graph_shapes:
            if self.performer is not None and self._graph_auto:
                self.performer.check_redraw_projections()    # host-side bookkeeping; a redraw needs fresh copies
                from . import stack
                stack.refresh_weights(self.model.rna_model.model)
            for dst, src in zip(self._static_in, (rna, atac_raw, latent_real)):
                dst.copy_(src)
            self._graph.replay()
            losses = {k: v.clone() for k, v in self._static_out.items()}
        else:
            if self._graph is not None and self.performer is not None and self._graph_auto:
                self.performer.check_redraw_projections()    # the model's own check is off while a recording exists
            losses = self._forward_backward(rna, atac_raw, latent_real)
        if self.index % self.grad_acc == 0:
            for p, a in zip(self.params, self.accs):
                p.grad = a
            if self.averager is not None:
                self.averager.average_()
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), int(1e4))
            self.optimizer.step()
            if self.accs:
                torch._foreach_zero_(self.accs)
            for p in self.params:
                p.grad = None
        return losses
