"""Data-parallel plumbing of the S1 step: cells are sharded across ranks, weights are replicated and the
parameter gradients are averaged once per optimizer step (s1_pretrain_rna.py:187-190,280,294-305, where
DistributedSampler and DistributedDataParallel do this).  Pure torch.distributed: the backend is RCCL
("nccl") on the GPUs and gloo in CPU tests; there is no data-path collective besides the gradient
all-reduce, and none at all for embedding extraction.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def sampler_indices(n_cells: int, rank: int, world: int, *, shuffle: bool = True, seed: int = 0, epoch: int = 0,
                    drop_last: bool = False) -> List[int]:
    """Cell indices of `rank` for one epoch, with torch.utils.data.DistributedSampler's semantics (the sampler
    the reference builds at s1_pretrain_rna.py:187-190 and re-seeds per epoch at :280): a permutation seeded
    with seed+epoch, padded by wrapping around (or truncated when drop_last) to a multiple of `world`, then
    strided: rank r takes positions r, r+world, ..."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n_cells, generator=g).tolist()
    else:
        order = list(range(n_cells))
    if drop_last and n_cells % world:
        per = n_cells // world
    else:
        per = math.ceil(n_cells / world)
    total = per * world
    if total > len(order):
        pad = total - len(order)
        order += (order * math.ceil(pad / max(1, len(order))))[:pad]
    else:
        order = order[:total]
    return order[rank:total:world]


def contiguous_shard(n_cells: int, rank: int, world: int) -> range:
    """Extraction partition: rank r owns one contiguous block of the cell list (sizes differ by at most one);
    the blocks are independent and their outputs are concatenated in rank order."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    base, extra = divmod(n_cells, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


class GradientAverager:
    """One flat f32 bucket for all parameter gradients: a single all-reduce (sum) per optimizer step, then a
    division by the world size, which is what DDP's reducer produces on the sync micro-step; accumulation
    micro-steps in between do not communicate (the reference wraps them in no_sync, s1_pretrain_rna.py:294-298).
    The bucket is allocated once; gradients are gathered into it and scattered back with foreach copies."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=torch.float32, device=ref.device)
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @torch.no_grad()
    def sum_in_place_(self) -> None:
        """The all-reduce alone, on the bucket in place (the caller accumulates into ``views`` and divides by the
        world size itself, e.g. inside its optimizer kernel)."""
        if self.world == 1:
            return
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def average_(self) -> None:
        """p.grad <- mean over ranks of p.grad, for every parameter (a missing grad counts as zero).  A caller that
        accumulates straight into ``views`` (p.grad is the view itself) pays no gather/scatter copies: the
        all-reduce runs on the bucket in place."""
        if self.world == 1:
            return
        in_place = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))
        if not in_place:
            src = [p.grad if p.grad is not None else torch.zeros_like(v) for p, v in zip(self.params, self.views)]
            torch._foreach_copy_(self.views, src)
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # CPU-only backend (rehearsals on a box without RCCL peers): stage the bucket through host memory
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        if not in_place:
            for p, v in zip(self.params, self.views):
                if p.grad is None:
                    p.grad = v.clone()
            torch._foreach_copy_([p.grad for p in self.params], self.views)


def broadcast_parameters(params: Sequence[torch.Tensor], src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Replicate rank `src`'s parameters and buffers (what DDP does at construction)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    gloo = dist.get_backend(group) == "gloo"
    for p in params:
        t = p.data if isinstance(p, torch.nn.Parameter) else p
        if gloo and t.is_cuda:          # CPU-only backend: stage device tensors through host memory
            host = t.cpu()
            dist.broadcast(host, src=src, group=group)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src, group=group)


@torch.no_grad()
def broadcast_projections(performer, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Every replica takes rank `src`'s FAVOR+ projection matrices and redraw counter: one flat broadcast of all
    FastAttention.projection_matrix buffers plus one of performer.calls_since_last_redraw.  The reference gets the
    same effect from DistributedDataParallel's buffer broadcast at every forward (s1_pretrain_rna.py:242 with
    performer_pytorch.py:209,520-534); here it only has to run when the matrices change, i.e. right after a redraw
    (Performer.redraw_sync).  A no-op without a process group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    mats = [m.projection_matrix for m in performer.modules() if hasattr(m, "projection_matrix")]
    if not mats:
        return
    gloo = dist.get_backend(group) == "gloo"
    flat = torch.cat([m.reshape(-1).to(torch.float32) for m in mats])
    counter = performer.calls_since_last_redraw.reshape(1).clone()
    if gloo and flat.is_cuda:           # CPU-only backend: stage through host memory
        flat, counter = flat.cpu(), counter.cpu()
    dist.broadcast(flat, src=src, group=group)
    dist.broadcast(counter, src=src, group=group)
    off = 0
    for m in mats:
        m.copy_(flat[off:off + m.numel()].view_as(m))
        off += m.numel()
    performer.calls_since_last_redraw.copy_(counter.reshape(()))
    if hasattr(performer, "_calls_host"):
        performer._calls_host = None    # the host mirror of the counter is re-read at the next forward


def max_over_ranks(seconds: float, device) -> float:
    """Step time of the job = the slowest rank's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
