"""The S1 RNA masked-autoencoding pretraining loop (s1_pretrain_rna.py:146-387) as a reusable driver: same step
semantics (mask -> forward -> CE / grad_acc -> backward; every grad_acc-th micro-step: gradient average over the
ranks, clip to norm 100, Adam, zero), same learning-rate schedule (utils.py:59-145), same accuracy definition,
same checkpoint dictionary (``module.``-prefixed weights, optimizer and scheduler state, epoch, loss), same
early stopping.  What differs is how it runs: masking, loss and encoder are device kernels, gradient accumulation
is one multi-tensor add into static buffers, the data-parallel exchange is one flat all-reduce per optimizer step
(dist.GradientAverager) instead of DDP's bucketed hooks, nothing synchronises with the host inside an epoch (the
reference calls ``.item()`` twice per micro-step, :307,312), and the micro-step can be replayed from a HIP graph.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import LRScheduler

from . import dist as sdist
from . import s1

PREFIX = "module."


# ------------------------------------------------------------------------------------------------
# learning-rate schedule
# ------------------------------------------------------------------------------------------------
class CosineAnnealingWarmupRestarts(LRScheduler):
    """utils.py:59-145.  Cycles of ``cur_cycle_steps`` scheduler steps: linear warm-up from ``min_lr`` to the
    cycle's ``max_lr`` over ``warmup_steps``, then half a cosine back down to ``min_lr``; each new cycle is
    ``cycle_mult`` times longer (warm-up excluded) and its peak ``gamma`` times lower.  Attribute names are the
    reference's, so its ``scheduler_state_dict`` loads here and the other way round.  As in the reference the
    optimizer runs at ``min_lr`` until the first ``step()``."""

    def __init__(self, optimizer: torch.optim.Optimizer, first_cycle_steps: int, cycle_mult: float = 1.,
                 max_lr: float = 0.1, min_lr: float = 0.001, warmup_steps: int = 0, gamma: float = 1.,
                 last_epoch: int = -1):
        assert warmup_steps < first_cycle_steps
        self.first_cycle_steps = first_cycle_steps
        self.cycle_mult = cycle_mult
        self.base_max_lr = max_lr
        self.max_lr = max_lr
        self.min_lr = min_lr
        self.warmup_steps = warmup_steps
        self.gamma = gamma
        self.cur_cycle_steps = first_cycle_steps
        self.cycle = 0
        self.step_in_cycle = last_epoch
        super().__init__(optimizer, last_epoch)
        self.init_lr()

    def init_lr(self):
        self.base_lrs = [self.min_lr for _ in self.optimizer.param_groups]
        for group in self.optimizer.param_groups:
            group["lr"] = self.min_lr

    def _lr_from(self, floor: float) -> float:
        pos = self.step_in_cycle
        if pos == -1:
            return floor
        span = self.max_lr - floor
        if pos < self.warmup_steps:
            return span * pos / self.warmup_steps + floor
        phase = math.pi * (pos - self.warmup_steps) / (self.cur_cycle_steps - self.warmup_steps)
        return floor + span * (1 + math.cos(phase)) / 2

    def get_lr(self):
        return [self._lr_from(floor) for floor in self.base_lrs]

    def _advance_one(self):
        self.step_in_cycle += 1
        if self.step_in_cycle >= self.cur_cycle_steps:          # next cycle: longer by cycle_mult (warm-up excluded)
            self.cycle += 1
            self.step_in_cycle -= self.cur_cycle_steps
            self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps

    def _jump_to(self, epoch):
        first, mult = self.first_cycle_steps, self.cycle_mult
        if epoch < first:
            self.cur_cycle_steps, self.step_in_cycle = first, epoch
        elif mult == 1.:
            self.cycle, self.step_in_cycle = epoch // first, epoch % first
        else:                                                   # geometric series of cycle lengths
            n = int(math.log(epoch / first * (mult - 1) + 1, mult))
            self.cycle = n
            self.step_in_cycle = epoch - int(first * (mult ** n - 1) / (mult - 1))
            self.cur_cycle_steps = first * mult ** n

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
            self._advance_one()
        else:
            self._jump_to(epoch)
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        self.last_epoch = math.floor(epoch)
        lrs = self.get_lr()
        for group, lr in zip(self.optimizer.param_groups, lrs):
            group["lr"] = lr
        self._last_lr = lrs


# ------------------------------------------------------------------------------------------------
# pieces of the loop
# ------------------------------------------------------------------------------------------------
def s1_predictions(logits: torch.Tensor) -> torch.Tensor:
    """s1_pretrain_rna.py:308-309: arg-max over the classes 1..C-2 (softmax is monotonic, so the logits do)."""
    return logits[..., 1:-1].argmax(dim=-1) + 1


def s1_batch_accuracy(logits: torch.Tensor, labels: torch.Tensor, pad_id: int) -> torch.Tensor:
    """s1_pretrain_rna.py:308-312: per cell, correct / predicted over the masked positions; mean over the cells
    (a cell without masked positions yields nan, as in the reference).  Device scalar."""
    if logits.is_cuda and logits.dim() == 3 and logits.dtype == torch.float32:
        from . import ops
        return ops.s1_accuracy(logits.contiguous(), labels.contiguous(), pad_id)[0].mean()   # one kernel per batch
    correct_num, pred_num = _cell_counts(logits, labels, pad_id)
    return torch.true_divide(correct_num, pred_num).mean()


def s1_cell_accuracy(logits: torch.Tensor, labels: torch.Tensor, pad_id: int) -> torch.Tensor:
    """correct / predicted over the masked positions of every cell, f32 [B] (nan for a cell without masked
    positions); s1_batch_accuracy is its mean."""
    if logits.is_cuda and logits.dim() == 3 and logits.dtype == torch.float32:
        from . import ops
        return ops.s1_accuracy(logits.contiguous(), labels.contiguous(), pad_id)[0]
    correct_num, pred_num = _cell_counts(logits, labels, pad_id)
    return torch.true_divide(correct_num, pred_num).to(torch.float32)


def _cell_counts(logits: torch.Tensor, labels: torch.Tensor, pad_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(correct, counted) per cell, int64 [B] each."""
    if logits.is_cuda and logits.dim() == 3 and logits.dtype == torch.float32:
        from . import ops
        counts = ops.s1_accuracy(logits.contiguous(), labels.contiguous(), pad_id)[1].to(torch.int64)
        return counts[:, 0], counts[:, 1]
    final = s1_predictions(logits)
    counted = labels != pad_id
    return (counted * (final == labels)).sum(dim=-1), counted.sum(dim=-1)


def freeze_for_finetune(model) -> Tuple[int, int]:
    """Finetune mode of s1_pretrain_rna.py:223-228: everything frozen except the final norm and the last two
    Performer layers.  -> (total, trainable) parameter counts."""
    for p in model.parameters():
        p.requires_grad = False
    for p in model.norm.parameters():
        p.requires_grad = True
    for p in model.performer.net.layers[-2:].parameters():
        p.requires_grad = True
    total = sum(p.numel() for p in model.parameters())
    return total, sum(p.numel() for p in model.parameters() if p.requires_grad)


def _strip_prefix(sd: Dict[str, torch.Tensor], prefix: str = PREFIX) -> Dict[str, torch.Tensor]:
    if sd and all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return dict(sd)


def checkpoint_dict(epoch: int, model, optimizer, scheduler, val_loss: float) -> dict:
    """The dictionary of s1_pretrain_rna.py:370-376.  The reference saves the DDP wrapper's state_dict, whose keys
    carry a ``module.`` prefix; the same keys are written here so that downstream loaders (the S2 script strips
    the prefix) see the layout they expect."""
    return {"epoch": epoch,
            "model_state_dict": {PREFIX + k: v for k, v in model.state_dict().items()},
            "optimizer_state_dict": optimizer.state_dict(),
            "scheduler_state_dict": scheduler.state_dict(),
            "losses": val_loss}


def save_checkpoint(path: str, epoch: int, model, optimizer, scheduler, val_loss: float) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(checkpoint_dict(epoch, model, optimizer, scheduler, val_loss), path)


def load_checkpoint(path_or_dict, model, optimizer=None, scheduler=None) -> int:
    """Resume (s1_pretrain_rna.py:207-215,258-266): weights (with or without the ``module.`` prefix), then the
    optimizer and scheduler states when present and wanted.  -> the epoch to start with."""
    ckpt = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu",
                                                                         weights_only=True)
    model.load_state_dict(_strip_prefix(ckpt["model_state_dict"]))
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt.get("epoch", 0) + 1


def _device_mask(data, mask_prob, replace_prob, pad_id, mask_id):
    return s1.data_mask_device(data, mask_prob, replace_prob, pad_id, mask_id)


def _device_loss(logits, labels, pad_id):
    return s1.masked_cross_entropy(logits, labels, pad_id)


# ------------------------------------------------------------------------------------------------
# the driver
# ------------------------------------------------------------------------------------------------
class _Recorded:
    """One recorded launch chain: the HIP graph of a micro-step (or of a window of micro-steps) for one batch shape,
    with its static input and output buffers."""
    __slots__ = ("graph", "static_in", "static_out")

    def __init__(self, graph, static_in, static_out):
        self.graph, self.static_in, self.static_out = graph, static_in, static_out


def adam_state_for_kernel(opt, params):
    """(exp_avgs, exp_avg_sqs, steps) of torch.optim.Adam for the given trainable parameters, or None when the step
    cannot go through the optimizer kernel: parameters off the GPU or not f32, an Adam variant the kernel does not
    implement, or moments that do not exist yet (torch creates them in its first step)."""
    if not params or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
        return None
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("capturable") or \
            g.get("differentiable") or isinstance(g["lr"], torch.Tensor):
        return None
    ms, vs, steps = [], [], []
    for p in params:
        st = opt.state.get(p)
        if not st or "exp_avg" not in st or not st["exp_avg"].is_cuda:
            return None
        ms.append(st["exp_avg"]), vs.append(st["exp_avg_sq"]), steps.append(st["step"])
    return ms, vs, steps


class S1Trainer:
    """Owns optimizer, schedule, gradient accumulators and the data-parallel exchange for one model replica.

    ``mask_fn(data, mask_prob, replace_prob, pad_id, mask_id) -> (masked_input, labels)`` and
    ``loss_fn(logits, labels, pad_id) -> scalar`` default to the device kernels (there is no CPU path behind
    them); host tests inject torch formulations to drive the bookkeeping without a GPU.

    ``window``: how many consecutive micro-steps ``window_step`` (and ``train_epoch``) run as ONE launch chain.  The
    micro-steps of an accumulation window share one set of weights (s1_pretrain_rna.py:294-305: no optimizer step
    in between), so ``window`` micro-batches stacked along dim 0 go through the encoder in one forward and one
    backward pass; what the reference computes per micro-batch stays per micro-batch: the key stabiliser of the
    attention (one max per forward call, performer_pytorch.py:89-90), the mean of the loss and the random draws of
    the masking.  ``grad_acc`` must be a multiple of ``window``."""

    def __init__(self, model, *, learning_rate: float = 1e-4, grad_acc: int = 20, mask_prob: float = 0.15,
                 replace_prob: float = 0.9, bin_num: int = 5, max_grad_norm: float = int(1e2),
                 group: Optional[dist.ProcessGroup] = None, use_graph: bool = False,
                 mask_fn: Optional[Callable] = None, loss_fn: Optional[Callable] = None, window: int = 1,
                 fused_optimizer: bool = True):
        self.model = model
        self.grad_acc = int(grad_acc)
        self.window = int(window)
        if self.window < 1 or self.grad_acc % self.window != 0:
            raise ValueError(f"grad_acc ({grad_acc}) must be a multiple of window ({window})")
        self.mask_prob, self.replace_prob = mask_prob, replace_prob
        self.num_classes = bin_num + 2                       # s1_pretrain_rna.py:154-156
        self.pad_id = self.mask_id = self.num_classes - 1
        self.max_grad_norm = max_grad_norm
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.mask_fn = mask_fn or _device_mask
        self.loss_fn = loss_fn or _device_loss
        # the reference hands ALL parameters to Adam (frozen ones never get a gradient and are skipped), which
        # fixes the parameter numbering inside optimizer_state_dict
        self.optimizer = torch.optim.Adam(list(model.parameters()), lr=learning_rate)
        self.scheduler = CosineAnnealingWarmupRestarts(self.optimizer, first_cycle_steps=15, cycle_mult=2,
                                                       max_lr=learning_rate, min_lr=1e-6, warmup_steps=5, gamma=0.9)
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.averager = sdist.GradientAverager(self.params, group) if self.params else None
        # the accumulators ARE the slices of the all-reduce bucket: the exchange needs no gather/scatter copies
        self.accs = list(self.averager.views) if self.averager is not None else []
        sdist.broadcast_parameters(list(model.parameters()) + list(model.buffers()), src=0, group=group)
        # replicas must keep sharing ONE set of projection matrices when the model redraws them (every
        # feature_redraw_interval forward calls in train mode): rank 0's draw wins, as under the reference's DDP
        performer = getattr(model, "performer", None)
        if performer is not None and self.world > 1:
            performer.redraw_sync = lambda perf: sdist.broadcast_projections(perf, src=0, group=group)
        self.index = 0                                       # micro-steps since the start of the epoch
        self._seeds: Dict[Tuple[torch.device, int], torch.Tensor] = {}   # 1/grad_acc per loss: the backward's seed
        self.use_graph = use_graph
        self._recorded: Dict[Tuple[tuple, int], _Recorded] = {}
        self._graph_auto = None                              # the model's own redraw switch while graphs exist
        self.fused_optimizer = fused_optimizer               # clip + Adam + zero as one device kernel where possible
        self._adam_table = None

    # ---- one micro-step, or a window of them -------------------------------------------------
    def _mask_window(self, data: torch.Tensor, groups: int):
        """The masking of `groups` stacked micro-batches, with the random draws made micro-batch by micro-batch in
        the order the reference makes them (pick draws, then replace draws: s1_pretrain_rna.py:114-119), so that a
        window consumes the generator exactly as the same micro-steps run one by one."""
        b = data.shape[0] // groups
        if self.mask_fn is not _device_mask:
            parts = [self.mask_fn(data[g * b:(g + 1) * b], self.mask_prob, self.replace_prob, self.pad_id, self.mask_id)
                     for g in range(groups)]
            return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        pick = torch.empty(data.shape, dtype=torch.float32, device=data.device)
        repl = torch.empty(data.shape, dtype=torch.float32, device=data.device)
        for g in range(groups):
            torch.rand((b,) + tuple(data.shape[1:]), out=pick[g * b:(g + 1) * b])
            repl[g * b:(g + 1) * b].uniform_(0, 1)
        return s1.data_mask_device(data, self.mask_prob, self.replace_prob, self.pad_id, self.mask_id, pick, repl)

    def _forward_backward(self, data: torch.Tensor, groups: int = 1):
        """groups == 1: one micro-step -> (loss / grad_acc, accuracy) as device scalars.  groups > 1: data holds that
        many micro-batches stacked along dim 0 -> the same two values per micro-batch, as vectors [groups]."""
        if groups > 1 and data.shape[0] % groups != 0:
            raise ValueError(f"a window of {groups} micro-batches needs a multiple of {groups} cells, got {data.shape[0]}")
        b = data.shape[0] // groups
        performer = getattr(self.model, "performer", None)
        # The encoder's backward adds its gradients into the accumulators itself (its last launch, the batched
        # reductions, does the add): the forward run inside grad_sink remembers them.  What autograd hands back are
        # the gradients of anything outside the fused stack.
        from . import stack
        with stack.grad_sink(self.params, self.accs):
            if groups == 1:
                masked, labels = self.mask_fn(data, self.mask_prob, self.replace_prob, self.pad_id, self.mask_id)
                logits = self.model(masked)
                raw = self.loss_fn(logits, labels, self.pad_id)
            else:
                masked, labels = self._mask_window(data, groups)
                if performer is not None:
                    before = performer.stabiliser_group_cells
                    performer.stabiliser_group_cells = b        # one key stabiliser per micro-batch
                try:
                    logits = self.model(masked)
                finally:
                    if performer is not None:
                        performer.stabiliser_group_cells = before
                if self.loss_fn is _device_loss:
                    raw = s1.masked_cross_entropy(logits, labels, self.pad_id, group_cells=b)
                else:
                    raw = torch.stack([self.loss_fn(logits[g * b:(g + 1) * b], labels[g * b:(g + 1) * b], self.pad_id)
                                       for g in range(groups)])
            loss = raw / self.grad_acc
            if self.params:
                # autograd.grad hands the gradients back without going through the parameters' AccumulateGrad nodes
                # (which are tied to the stream they were created on: recording this step on a side stream would mix
                # streams).  The backward starts from the unscaled losses with 1/grad_acc as the seed of each: the
                # value the division's own backward would pass on (1.0 * 1/grad_acc), without its two small kernels.
                key = (raw.device, groups)
                if key not in self._seeds:
                    self._seeds[key] = torch.full(raw.shape, 1.0, dtype=raw.dtype, device=raw.device) / self.grad_acc
                grads = torch.autograd.grad(raw, self.params, grad_outputs=self._seeds[key], allow_unused=True)
                left = [(a, g) for a, g in zip(self.accs, grads) if g is not None]
                if left:
                    torch._foreach_add_([a for a, _ in left], [g for _, g in left])
        if groups == 1:
            return loss.detach(), s1_batch_accuracy(logits.detach(), labels, self.pad_id)
        return loss.detach(), s1_cell_accuracy(logits.detach(), labels, self.pad_id).view(groups, b).mean(dim=1)

    def _fused_tail_state(self):
        return adam_state_for_kernel(self.optimizer, self.params) if self.fused_optimizer else None

    def _optimizer_step(self):
        """The sync micro-step's tail (s1_pretrain_rna.py:299-305): average over ranks, clip, Adam, zero.  On the GPU
        (from the second optimizer step on: torch creates Adam's moments in the first) this is one all-reduce, the
        three launches of the optimizer kernel and the one-launch refresh of the bf16 weight copies; Adam's state stays
        torch's own (same state_dict)."""
        fused = self._fused_tail_state()
        if fused is not None:
            from . import ops, stack
            ms, vs, steps = fused
            if self._adam_table is None or not self._adam_table.matches(self.params, self.accs, ms, vs):
                self._adam_table = ops.AdamTable(self.params, self.accs, ms, vs)
            self.averager.sum_in_place_()
            for st in steps:
                st += 1
            g = self.optimizer.param_groups[0]
            ops.adam_step(self._adam_table, step=int(steps[0]), lr=float(g["lr"]), beta1=g["betas"][0], beta2=g["betas"][1],
                          eps=g["eps"], max_norm=float(self.max_grad_norm), grad_scale=1.0 / self.world)
            torch.autograd.graph.increment_version(self.params)      # the kernel wrote the parameters in place
            stack.refresh_weights(self.model)
            return
        for p, a in zip(self.params, self.accs):
            p.grad = a
        if self.averager is not None:
            self.averager.average_()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        if self.accs:
            torch._foreach_zero_(self.accs)
        for p in self.params:
            p.grad = None
        if self._recorded:
            from . import stack
            stack.refresh_weights(self.model)                # the graphs read the bf16 copies in place

    def _capture(self, data: torch.Tensor, groups: int = 1) -> _Recorded:
        """Record the launch chain for this batch shape once; the redraw check stays outside the recording."""
        static_in = data.clone()
        shape = () if groups == 1 else (groups,)
        static_out = [torch.zeros(shape, device=data.device), torch.zeros(shape, device=data.device)]
        saved = [a.clone() for a in self.accs]
        performer = getattr(self.model, "performer", None)
        if performer is not None and not self._recorded:
            self._graph_auto = performer.auto_check_redraw
        if performer is not None:
            performer.auto_check_redraw = False

        def body():
            loss, acc = self._forward_backward(static_in, groups)
            static_out[0].copy_(loss)
            static_out[1].copy_(acc)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: device calls made meanwhile by other threads of the process (the collective
        # library's watchdog polls events) must not invalidate the recording
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body()
        for a, s_ in zip(self.accs, saved):                  # warm-up and recording passes do not count
            a.copy_(s_)
        rec = _Recorded(graph, static_in, static_out)
        self._recorded[(tuple(data.shape), groups)] = rec
        return rec

    def release_graph(self) -> None:
        """Drop the recorded launch chains (the next step records again) and hand the redraw check back to the
        model's own forward, e.g. before the model is used outside this driver."""
        performer = getattr(self.model, "performer", None)
        if self._recorded and performer is not None:
            performer.auto_check_redraw = self._graph_auto
        self._recorded = {}

    def prepare(self, example: torch.Tensor, groups: int = 1) -> None:
        """Record the graph for batches shaped like ``example`` now (otherwise the first step of that shape does it);
        groups > 1: ``example`` is a window of that many stacked micro-batches.  The warm-up passes of the recording
        leave weights and accumulators untouched."""
        if self.use_graph and example.is_cuda and (tuple(example.shape), groups) not in self._recorded:
            self._capture(example, groups)

    def _run(self, data: torch.Tensor, groups: int):
        """`groups` micro-steps over `data` as one launch chain: replayed from its graph when graphs are in use."""
        performer = getattr(self.model, "performer", None)
        if self.use_graph and data.is_cuda:
            rec = self._recorded.get((tuple(data.shape), groups)) or self._capture(data, groups)
            if performer is not None and self._graph_auto:
                for _ in range(groups):                      # one tick of the redraw counter per micro-step
                    performer.check_redraw_projections()     # host-side bookkeeping; a redraw needs fresh copies
                from . import stack
                stack.refresh_weights(self.model)
            rec.static_in.copy_(data)
            rec.graph.replay()
            return rec.static_out[0].clone(), rec.static_out[1].clone()
        if performer is not None and groups > 1 and performer.auto_check_redraw:
            for _ in range(groups - 1):                      # the model's forward ticks once itself
                performer.check_redraw_projections()
        return self._forward_backward(data, groups)

    def micro_step(self, data: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """One pass of s1_pretrain_rna.py:287-305 over a batch of tokens int64 [B, N].  -> (loss / grad_acc,
        batch accuracy) as device scalars; nothing waits for the device."""
        self.index += 1
        loss, acc = self._run(data, 1)
        if self.index % self.grad_acc == 0:
            self._optimizer_step()
        return loss, acc

    def window_step(self, data: torch.Tensor, groups: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """``groups`` (default: ``window``) consecutive micro-steps of s1_pretrain_rna.py:287-305 as one launch chain.
        data: int64 [groups * B, N], the micro-batches stacked in the order the loop would have met them.
        -> (loss / grad_acc, batch accuracy) per micro-batch, device vectors [groups].  Every micro-batch gets what
        its own micro-step would have given it (own key stabiliser, own loss mean, own mask draws); the gradients
        add up in the accumulators in a different order (rounding differences only).  The window must not straddle
        an optimizer step; a projection redraw that falls due inside a window happens before the whole window."""
        groups = self.window if groups is None else int(groups)
        if groups == 1:
            loss, acc = self.micro_step(data)
            return loss.reshape(1), acc.reshape(1)
        if self.index % self.grad_acc + groups > self.grad_acc:
            raise ValueError(f"a window of {groups} micro-steps starting at micro-step {self.index} would straddle the "
                             f"optimizer step (grad_acc {self.grad_acc})")
        self.index += groups
        loss, acc = self._run(data, groups)
        if self.index % self.grad_acc == 0:
            self._optimizer_step()
        return loss, acc

    # ---- epochs -----------------------------------------------------------------------------
    def _mean_over_ranks(self, value: torch.Tensor) -> float:
        """utils.get_reduced (utils.py:148-156): sum over ranks / world size (here every rank gets the result)."""
        t = value.detach().to(torch.float64).reshape(1).clone()
        if self.world > 1:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / self.world

    def _check_tokens(self) -> None:
        """Bad token ids seen by the embedding kernel since the last check raise here, where the host has just
        waited for the device anyway (end of an epoch or of a validation pass: the optimizer steps of that epoch
        have already used the substituted embeddings; the reference's nn.Embedding faults on the step itself).  The
        flag is combined over the ranks first, so all ranks raise together."""
        pop = getattr(self.model, "pop_bad_token_flag", None)
        if pop is None:
            return
        bad = int(pop())
        if self.world > 1:                                   # any rank's bad token stops every rank, at the same point
            t = torch.tensor([bad], dtype=torch.int32)
            if dist.get_backend(self.group) != "gloo":
                t = t.to(next(self.model.parameters()).device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            bad = int(t.item())
        if bad:
            raise IndexError(self.model.bad_token_message() + (f" [seen on at least one of {self.world} ranks]"
                                                               if self.world > 1 else ""))

    def train_epoch(self, batches: Iterable[torch.Tensor]) -> Tuple[float, float]:
        """s1_pretrain_rna.py:281-322 for one epoch.  -> (training loss as the reference prints it, i.e. the mean
        of loss / grad_acc times grad_acc; accuracy in percent), both averaged over the ranks; then one
        scheduler step.  Gradients of an unfinished accumulation window carry over, as in the reference."""
        self.model.train()
        self.index = 0
        running_loss = running_acc = None
        n = 0

        def account(loss, acc, count):
            nonlocal running_loss, running_acc, n
            running_loss = loss.sum() if running_loss is None else running_loss + loss.sum()
            running_acc = acc.sum() if running_acc is None else running_acc + acc.sum()
            n += count

        held: List[torch.Tensor] = []                       # micro-batches waiting for their window to fill up

        def flush(as_window: bool):
            if as_window and self.index % self.grad_acc + len(held) <= self.grad_acc:
                account(*self.window_step(torch.cat(held, dim=0), len(held)), len(held))
            else:
                for h_ in held:
                    account(*self.micro_step(h_), 1)
            held.clear()

        for data in batches:
            if self.window == 1:
                account(*self.micro_step(data), 1)
                continue
            if held and tuple(data.shape) != tuple(held[0].shape):
                flush(False)                                 # a ragged batch (the epoch's last): one by one
            held.append(data)
            if len(held) == self.window:
                flush(True)
        if held:
            flush(False)
        if n == 0:
            raise ValueError("empty epoch")
        epoch_loss = self._mean_over_ranks(running_loss / n)
        epoch_acc = self._mean_over_ranks(100 * running_acc / n)
        self._check_tokens()
        self.scheduler.step()
        return epoch_loss * self.grad_acc, epoch_acc

    @torch.no_grad()
    def validate(self, batches: Iterable[torch.Tensor], n_total_cells: Optional[int] = None) -> Tuple[float, float]:
        """s1_pretrain_rna.py:326-355: mean validation loss over the rank's batches, averaged over ranks, and the
        accuracy over the masked positions of the first ``n_total_cells`` cells of the rank-major concatenation
        (what distributed_concat keeps, utils.py:158-166).  -> (loss, accuracy in percent)."""
        was_training = self.model.training
        self.model.eval()
        running = None
        n = 0
        correct, counted = [], []
        for data in batches:
            masked, labels = self.mask_fn(data, self.mask_prob, self.replace_prob, self.pad_id, self.mask_id)
            logits = self.model(masked)
            loss = self.loss_fn(logits, labels, self.pad_id).detach()
            running = loss if running is None else running + loss
            n += 1
            c_, n_ = _cell_counts(logits, labels, self.pad_id)
            correct.append(c_)
            counted.append(n_)
        if n == 0:
            raise ValueError("empty validation set")
        val_loss = self._mean_over_ranks(running / n)
        self._check_tokens()
        per_cell = torch.stack([torch.cat(correct), torch.cat(counted)], dim=1)      # [cells of this rank, 2]
        if self.world > 1:
            if per_cell.is_cuda and dist.get_backend(self.group) == "gloo":
                per_cell = per_cell.cpu()
            parts = [torch.empty_like(per_cell) for _ in range(self.world)]
            dist.all_gather(parts, per_cell.contiguous(), group=self.group)
            per_cell = torch.cat(parts, dim=0)
        if n_total_cells is not None:
            per_cell = per_cell[:n_total_cells]
        tot = per_cell.sum(dim=0)
        self.model.train(was_training)
        return val_loss, 100.0 * float(tot[0].item()) / float(tot[1].item())

    # ---- the whole run --------------------------------------------------------------------------
    def batches_of(self, dataset, batch_size: int, epoch: int, *, shuffle: bool = True, seed: int = 0):
        """This rank's batches for one epoch: DistributedSampler order (s1_pretrain_rna.py:187-190,280) cut into
        DataLoader batches (the last one may be smaller).  dataset: ingest.SCDataset (binned on the device, one
        launch per batch) or an int64 tensor [cells, genes]."""
        idx = sdist.sampler_indices(len(dataset), self.rank, self.world, shuffle=shuffle, seed=seed, epoch=epoch)
        for lo in range(0, len(idx), batch_size):
            sel = idx[lo:lo + batch_size]
            yield dataset.batch(sel) if hasattr(dataset, "batch") else dataset[sel]

    def fit(self, train_set, val_set, *, epochs: int = 25, batch_size: int = 3, start_epoch: int = 1,
            valid_every: int = 1, patience: int = 3, ckpt_dir: str = "./saved_model/", model_name: str = "pretrain_rna",
            log: Callable[[str], None] = print) -> List[dict]:
        """Epoch loop with validation, checkpoint on improvement and early stopping (s1_pretrain_rna.py:276-387).
        -> one record per epoch."""
        best = float("inf")
        patient = 0
        history = []
        for epoch in range(start_epoch, epochs + 1):
            loss, acc = self.train_epoch(self.batches_of(train_set, batch_size, epoch))
            rec = {"epoch": epoch, "train_loss": loss, "train_acc": acc, "lr": self.optimizer.param_groups[0]["lr"]}
            if self.rank == 0:
                log(f"    ==  Epoch: {epoch} | Training Loss: {loss:.6f} | Accuracy: {acc:6.4f}%  ==")
            if epoch % valid_every == 0:
                # the validation sampler is never re-seeded in the reference: epoch 0 order every time
                vloss, vacc = self.validate(self.batches_of(val_set, batch_size, 0), len(val_set))
                rec.update(val_loss=vloss, val_acc=vacc)
                stop = 0
                if self.rank == 0:
                    log(f"    ==  Epoch: {epoch} | Validation Loss: {vloss:.6f} | Accuracy: {vacc:6.4f}%  ==")
                    if vloss < best:
                        best, patient = vloss, 0
                        save_checkpoint(os.path.join(ckpt_dir, f"{model_name}.pth"), epoch, self.model, self.optimizer,
                                        self.scheduler, vloss)
                        log(f"    == model saved at epoch: {epoch} ==")
                    else:
                        patient += 1
                        stop = int(patient > patience)
                if self.world > 1:
                    flag = torch.tensor([stop], dtype=torch.int32)
                    if dist.get_backend(self.group) != "gloo":
                        flag = flag.to(next(self.model.parameters()).device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
                    stop = int(flag.item())
                history.append(rec)
                if stop:
                    if self.rank == 0:
                        log(f" Early stopping triggered at epoch {epoch}.")
                    break
            else:
                history.append(rec)
        return history
