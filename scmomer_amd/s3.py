"""The S3 distillation step (s3_pretrain_missing.py:233-316): scMomer in mode='one' with reconstruct=True.  The
Performer RNA encoder is frozen but runs in train() mode (its projection redraw counter ticks, :286 with
performer_pytorch.py:520-534); trainable are the student encoder, to_out and the two linear decoders.  The loss is
BCE-with-logits on the binarised ATAC target + MSE on the RNA reconstruction + distill_weight * MSE between the
student's embedding and the precomputed teacher embedding.  The frozen encoder runs on the kernels without a tape;
the trainable tail (two [B, 16906] x [16906, 128/256] products, the narrow middle, the two decoders with their
losses, and the backward of all of it) runs on its own eight launches (ops.DistillCall) when the model has the
reference's shape -- student MLP, single-Linear decoders, f32 on the GPU, at most 32 cells -- and as torch library
code otherwise.
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


def _single_linear(dec):
    seq = getattr(dec, "output_mean", None)
    if isinstance(seq, nn.Sequential) and len(seq) == 1 and isinstance(seq[0], nn.Linear) and seq[0].bias is not None:
        return seq[0]
    return None


TAIL_PARAMETERS = ("encoder.encoder.0.weight", "encoder.encoder.0.bias", "encoder.encoder.1.weight",
                   "encoder.encoder.1.bias", "encoder.encoder.4.weight", "encoder.encoder.4.bias", "to_out.weight",
                   "to_out.bias", "ATAC.output_mean.0.weight", "ATAC.output_mean.0.bias", "RNA.output_mean.0.weight",
                   "RNA.output_mean.0.bias")        # in the order of ops.DISTILL_GRAD_ORDER


def tail_parameters(model):
    """The twelve trainable tensors of the tail in the order the kernels take their gradients, or None when the
    model's tail is not the reference's (student MLP of extract_cell_emb.py:27-40, decoders of utils.py:168-179)."""
    from .scmomer import StudentEncoder
    enc = getattr(model.encoder, "encoder", None)
    atac, rna = _single_linear(model.ATAC), _single_linear(model.RNA)
    if not isinstance(model.encoder, StudentEncoder) or enc is None or atac is None or rna is None:
        return None
    return [enc[0].weight, enc[0].bias, enc[1].weight, enc[1].bias, enc[4].weight, enc[4].bias, model.to_out.weight,
            model.to_out.bias, atac.weight, atac.bias, rna.weight, rna.bias]


def encoder_scores(model, rna, group_cells=None):
    """The frozen encoder's per-gene output [cells, genes] (scmomer.py:56,124-126; no tape).  group_cells: `rna` is
    several batches of that many cells stacked; each keeps the key stabiliser of its own forward call."""
    from . import ingest
    performer = getattr(getattr(model.rna_model, "model", None), "performer", None)
    old = getattr(performer, "stabiliser_group_cells", None)
    if performer is not None:
        performer.stabiliser_group_cells = group_cells
    try:
        with torch.no_grad():
            return model.rna_model(ingest.bin_cells(rna, ingest.RULE_SCMOMER, 5)).contiguous()
    finally:
        if performer is not None:
            performer.stabiliser_group_cells = old


def fused_tail_call(model, rna, atac_raw, latent_real, pos_weight, distill_weight: float = 1.0, genes=None,
                    group_cells: int = 0):
    """The kernel call for this batch (the frozen encoder runs here unless its per-gene output `genes` is given), or
    None when the kernels do not apply: another tail than the reference's, a frozen part that wants gradients, tensors that
    are not f32 on the GPU, more cells than one call takes."""
    from . import ingest, ops
    if not (rna.is_cuda and rna.dtype == torch.float32 and rna.dim() == 2 and rna.is_contiguous()):
        return None
    tail = tail_parameters(model)
    if tail is None:
        return None
    frozen = list(model.rna_model.parameters()) + [model.rna_projection.weight, model.rna_ln.weight, model.rna_ln.bias]
    if any(p.requires_grad for p in frozen):
        return None
    w1, b1, lw, lb, w2, b2, wo, bo, wa, ba, wr, br = tail
    p, hd, o = model.rna_projection.weight.shape[0], w1.shape[0], wo.shape[0]
    ops_in = [atac_raw, latent_real, pos_weight, model.rna_projection.weight, model.rna_ln.weight, model.rna_ln.bias, *tail]
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ops_in):
        return None
    if not ops.distill_supported(rna.shape[0], p, hd, o) or model.rna_ln.eps != 1e-5 or model.encoder.encoder[1].eps != 1e-5:
        return None
    if tuple(w2.shape) != (p, hd) or wo.shape[1] != 2 * p or tuple(wr.shape) != (rna.shape[1], o) or wa.shape[1] != o or \
            tuple(atac_raw.shape) != (rna.shape[0], wa.shape[0]) or tuple(latent_real.shape) != (rna.shape[0], p) or \
            pos_weight.numel() not in (1, wa.shape[0]) or w1.shape[1] != rna.shape[1] or \
            model.rna_projection.weight.shape[1] != rna.shape[1]:
        return None
    drop_layer = model.encoder.encoder[3]
    drop = None
    if drop_layer.training and drop_layer.p > 0:      # nn.Dropout's multipliers, drawn as torch draws them
        drop = nn.functional.dropout(torch.ones((rna.shape[0], hd), device=rna.device), drop_layer.p, True)
    if genes is None:
        genes = encoder_scores(model, rna)
    d = lambda t: t.detach()
    return ops.DistillCall(genes, rna, atac_raw, latent_real, drop, pos_weight, d(model.rna_projection.weight),
                           (d(model.rna_ln.weight), d(model.rna_ln.bias)), d(w1), d(b1), (d(lw), d(lb)), d(w2), d(b2),
                           d(wo), d(bo), d(wa), d(ba), d(wr), d(br), normalize=bool(model.config.normalize),
                           distill_weight=distill_weight, group_cells=group_cells)


class _DistillTail(torch.autograd.Function):
    """losses [4] = (BCE, RNA MSE, distillation MSE, their weighted sum) as a function of the twelve tail tensors."""

    @staticmethod
    def forward(ctx, call, *weights):
        ctx.call = call
        return call.forward()[0]

    @staticmethod
    def backward(ctx, g):
        w = torch.tensor([1.0, 1.0, float(ctx.call.args.distill_weight)], device=g.device)
        grads = ctx.call.backward((g[:3] + g[3] * w).float().contiguous())
        return (None, *[gr.view(shape) for gr, shape in zip(grads, (tuple(ctx.call.shapes[n]) for n in _order()))])


def _order():
    from . import ops
    return ops.DISTILL_GRAD_ORDER


def distillation_losses(model, rna: torch.Tensor, atac_raw: torch.Tensor, latent_real: torch.Tensor,
                        pos_weight: torch.Tensor, distill_weight: float, fused: bool = True) -> Dict[str, torch.Tensor]:
    """One forward of s3_pretrain_missing.py:291-307 -> {'atac', 'rna', 'distill', 'total'} (total is the sum the
    script backpropagates, before the division by grad_acc).  fused=False keeps the tail on torch's library code."""
    call = fused_tail_call(model, rna, atac_raw, latent_real, pos_weight, distill_weight) if fused else None
    if call is not None:
        four = _DistillTail.apply(call, *tail_parameters(model))
        return {"atac": four[0], "rna": four[1], "distill": four[2], "total": four[3]}
    else:
        atac_target = (atac_raw > 0).float()
        r_atac, r_rna, atac_embeds = model(rna_values=rna, reconstruct=True, mode='one')
        loss_atac = nn.functional.binary_cross_entropy_with_logits(r_atac, atac_target, pos_weight=pos_weight)
        loss_rna = nn.functional.mse_loss(r_rna, rna)
        loss_distill = nn.functional.mse_loss(atac_embeds, latent_real)
    return {"atac": loss_atac, "rna": loss_rna, "distill": loss_distill,
            "total": loss_atac + loss_rna + distill_weight * loss_distill}


LOSS_NAMES = ("atac", "rna", "distill", "total")


class S3Step:
    """Optimizer-step bookkeeping of the S3 loop (:296-316): loss / grad_acc, backward on every micro-step; every
    grad_acc-th one averages the gradients over the ranks (one flat all-reduce instead of DDP's hooks), clips to
    1e4 and applies Adam (all parameters are handed to Adam, as in the script; frozen ones never get a gradient).
    Gradients are folded into static accumulators (the slices of the all-reduce bucket: the tail kernels add into
    them directly), which lets the micro-step (forward, losses, backward, accumulate) be recorded once into a HIP
    graph and replayed (use_graph=True).  window_step runs several micro-steps of an accumulation window as one
    launch chain: the frozen encoder once over all their cells, the tail per micro-batch.  fused=False keeps the
    tail and the optimizer step on torch's library code."""

    def __init__(self, model, *, lr: float = 1e-4, grad_acc: int = 1, pos_weight: float = 1.0,
                 distill_weight: float = 1.0, group=None, use_graph: bool = False, fused: bool = True):
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
        self._recorded = {}
        self._graph_auto = None
        self._seed = None
        self.fused = fused
        self._gloss = None
        self._adam_table = None
        tail = tail_parameters(model) if fused else None
        acc_of = {id(p): a for p, a in zip(self.params, self.accs)}
        # the kernels' backward adds straight into the accumulators: every tail tensor must be trainable
        self._tail_accs = [acc_of[id(p)] for p in tail] if tail is not None and all(id(p) in acc_of for p in tail) and \
            len(self.params) == len(tail) else None

    def _forward_backward(self, rna, atac_raw, latent_real, groups: int = 1) -> torch.Tensor:
        """-> [groups, 4]: per micro-batch the losses in the order of LOSS_NAMES.  groups > 1 (tail kernels only): the
        inputs are that many micro-batches stacked; the frozen encoder runs once over all of them with one key
        stabiliser per micro-batch, the tail once with one set of loss means per micro-batch."""
        b = rna.shape[0] // groups
        if self._tail_accs is not None:
            if self._gloss is None:
                self._gloss = torch.tensor([1.0, 1.0, float(self.distill_weight)], device=rna.device) / self.grad_acc
            from . import ops
            genes = encoder_scores(self.model, rna, b if groups > 1 else None)
            per_call = max(1, ops.distill_max_cells() // b)          # micro-batches one call of the tail takes
            outs = []
            for g0 in range(0, groups, per_call):
                lo, hi = g0 * b, min(groups, g0 + per_call) * b
                call = fused_tail_call(self.model, rna[lo:hi], atac_raw[lo:hi], latent_real[lo:hi], self.pos_weight,
                                       self.distill_weight, genes=genes[lo:hi], group_cells=b if groups > 1 else 0)
                if call is None:
                    break
                outs.append(call.forward())
                call.backward(self._gloss, grads=self._tail_accs, accumulate=True)
            else:
                return outs[0] if len(outs) == 1 else torch.cat(outs)
            if groups > 1:
                raise RuntimeError("S3 windows need the tail kernels (f32 GPU tensors, at most 32 cells per micro-batch)")   # nothing has been added to the accumulators: the first call already said no
            # the encoder has ticked its redraw counter for this call already: torch's tail on its output
            return self._torch_tail(genes, rna, atac_raw, latent_real).view(1, -1)
        if groups > 1:
            raise RuntimeError("S3 windows need the tail kernels (the reference's tail with all of it trainable)")
        losses = distillation_losses(self.model, rna, atac_raw, latent_real, self.pos_weight, self.distill_weight,
                                     fused=False)
        return self._backward(losses).view(1, -1)

    def _torch_tail(self, genes, rna, atac_raw, latent_real) -> torch.Tensor:
        m = self.model
        feat = m.rna_projection(genes)
        if m.config.normalize:
            feat = feat / feat.norm(dim=-1, keepdim=True)
        student = m.encoder(rna)
        out = m.to_out(torch.cat([student, m.rna_ln(feat)], 1))
        l_atac = nn.functional.binary_cross_entropy_with_logits(m.ATAC(out), (atac_raw > 0).float(), pos_weight=self.pos_weight)
        l_rna = nn.functional.mse_loss(m.RNA(out), rna)
        l_dis = nn.functional.mse_loss(student, latent_real)
        return self._backward({"atac": l_atac, "rna": l_rna, "distill": l_dis,
                               "total": l_atac + l_rna + self.distill_weight * l_dis})

    def _backward(self, losses) -> torch.Tensor:
        if self.params:
            if self._seed is None:
                self._seed = torch.full((), 1.0, dtype=losses["total"].dtype, device=losses["total"].device) / self.grad_acc
            grads = torch.autograd.grad(losses["total"], self.params, grad_outputs=self._seed, allow_unused=True)
            left = [(a, g) for a, g in zip(self.accs, grads) if g is not None]
            if left:
                torch._foreach_add_([a for a, _ in left], [g for _, g in left])
        return torch.stack([losses[k].detach() for k in LOSS_NAMES])

    def _capture(self, rna, atac_raw, latent_real, groups: int):
        """Record the launch chain for these shapes once; the projection redraw check stays outside the recording."""
        static_in = [rna.clone(), atac_raw.clone(), latent_real.clone()]
        static_out = torch.zeros((groups, len(LOSS_NAMES)), device=rna.device)
        saved = [a.clone() for a in self.accs]
        calls = self.performer.calls_since_last_redraw.clone() if self.performer is not None else None
        if self.performer is not None:
            if not self._recorded:
                self._graph_auto = self.performer.auto_check_redraw
            self.performer.auto_check_redraw = False

        def body():
            static_out.copy_(self._forward_backward(*static_in, groups))

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
        rec = (graph, static_in, static_out)
        self._recorded[(tuple(rna.shape), tuple(atac_raw.shape), tuple(latent_real.shape), groups)] = rec
        return rec

    def _run(self, rna, atac_raw, latent_real, groups: int) -> torch.Tensor:
        perf = self.performer
        if self.use_graph and rna.is_cuda:
            key = (tuple(rna.shape), tuple(atac_raw.shape), tuple(latent_real.shape), groups)
            graph, static_in, static_out = self._recorded.get(key) or self._capture(rna, atac_raw, latent_real, groups)
            if perf is not None and self._graph_auto:
                for _ in range(groups):                      # one tick of the redraw counter per micro-step
                    perf.check_redraw_projections()          # host-side bookkeeping; a redraw needs fresh copies
                from . import stack
                stack.refresh_weights(self.model.rna_model.model)
            for dst, src in zip(static_in, (rna, atac_raw, latent_real)):
                dst.copy_(src)
            graph.replay()
            return static_out.clone()
        if perf is not None and self._recorded and self._graph_auto:
            for _ in range(groups):                          # the model's own check is off while a recording exists
                perf.check_redraw_projections()
        elif perf is not None and groups > 1 and perf.auto_check_redraw:
            for _ in range(groups - 1):                      # the encoder's forward ticks once itself
                perf.check_redraw_projections()
        return self._forward_backward(rna, atac_raw, latent_real, groups)

    def _after(self) -> None:
        if self.index % self.grad_acc == 0:
            if self._kernel_optimizer_step():
                return
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

    def micro_step(self, rna, atac_raw, latent_real) -> Dict[str, torch.Tensor]:
        """One pass of s3_pretrain_missing.py:291-316 over a batch -> the four losses as device scalars."""
        self.index += 1
        four = self._run(rna, atac_raw, latent_real, 1)[0]
        self._after()
        return {k: four[i] for i, k in enumerate(LOSS_NAMES)}

    def window_step(self, rna, atac_raw, latent_real, groups: int) -> Dict[str, torch.Tensor]:
        """``groups`` consecutive micro-steps as one launch chain: the inputs are the micro-batches stacked along
        dim 0 in the order the loop would have met them -> the four losses per micro-batch, device vectors [groups].
        Every micro-batch gets what its own micro-step would have given it (own key stabiliser in the frozen encoder,
        own loss means, own dropout draw); the window must not straddle an optimizer step; a projection redraw that
        falls due inside a window happens before the whole window."""
        groups = int(groups)
        if groups < 1 or rna.shape[0] % groups:
            raise ValueError(f"{rna.shape[0]} cells do not split into {groups} micro-batches")
        if self.index % self.grad_acc + groups > self.grad_acc:
            raise ValueError(f"a window of {groups} micro-steps starting at micro-step {self.index} would straddle the "
                             f"optimizer step (grad_acc {self.grad_acc})")
        self.index += groups
        out = self._run(rna, atac_raw, latent_real, groups)
        self._after()
        return {k: out[:, i] for i, k in enumerate(LOSS_NAMES)}

    def _kernel_optimizer_step(self) -> bool:
        """Rank sum, clip to 1e4, Adam and the zeroing of the accumulators in the three launches of the optimizer
        kernel (on torch.optim.Adam's own state; from the second optimizer step on, torch creates the moments in
        its first).  False: not applicable, the caller runs torch's."""
        from . import ops
        from .train import adam_state_for_kernel
        state = adam_state_for_kernel(self.optimizer, self.params) if self.fused else None
        if state is None:
            return False
        ms, vs, steps = state
        if self._adam_table is None or not self._adam_table.matches(self.params, self.accs, ms, vs):
            self._adam_table = ops.AdamTable(self.params, self.accs, ms, vs)
        self.averager.sum_in_place_()
        for st in steps:
            st += 1
        g = self.optimizer.param_groups[0]
        ops.adam_step(self._adam_table, step=int(steps[0]), lr=float(g["lr"]), beta1=g["betas"][0], beta2=g["betas"][1],
                      eps=g["eps"], max_norm=1e4, grad_scale=1.0 / self.averager.world)
        torch.autograd.graph.increment_version(self.params)
        return True
