"""Sharded cell-embedding extraction: the loop of evaluate_embedding.py:173-178 (``for x in loader:
model.get_cell_embedding(x)``) as a driver.  The cell list is cut into one contiguous shard per rank
(dist.contiguous_shard), every rank walks its shard in batches with the frozen model on its own GPU and keeps its
[cells, 128] block in device memory; there is no collective until the single final gather.
"""
from __future__ import annotations

from typing import Callable, Optional, Union

import torch
import torch.distributed as dist

from . import dist as sdist

CellSource = Union[torch.Tensor, Callable[[int, int], torch.Tensor]]


def _batch_of(cells: CellSource, lo: int, hi: int, device) -> torch.Tensor:
    """Expression rows lo..hi-1 as device f32 [hi-lo, genes]: a tensor is sliced (host tensors are copied per batch,
    as a DataLoader would hand them over), a callable is asked for the range."""
    x = cells(lo, hi) if callable(cells) else cells[lo:hi]
    return x.to(device=device, dtype=torch.float32, non_blocking=True)


@torch.no_grad()
def extract_embeddings(model, cells: CellSource, n_cells: int, batch_size: int = 16, *, rank: int = 0, world: int = 1,
                       device=None, embed_dim: Optional[int] = None) -> torch.Tensor:
    """This rank's block of embeddings, f32 [len(shard), embed_dim] on the device, in cell order.  Batches are the
    consecutive runs of `batch_size` cells of the shard (the last one may be shorter): the key stabiliser of the
    attention is taken over a batch (performer_pytorch.py:90), so results depend on which cells share a batch, and a
    sharded run reproduces a one-rank run exactly when the shard length is a multiple of the batch size."""
    if device is None:
        device = next(model.parameters()).device
    shard = sdist.contiguous_shard(n_cells, rank, world)
    out = None
    for lo in range(shard.start, shard.stop, batch_size):
        hi = min(lo + batch_size, shard.stop)
        emb = model.get_cell_embedding(_batch_of(cells, lo, hi, device))
        if out is None:
            out = torch.empty((len(shard), emb.shape[-1]), dtype=torch.float32, device=device)
        out[lo - shard.start:hi - shard.start] = emb
    if out is None:
        out = torch.empty((0, embed_dim or 128), dtype=torch.float32, device=device)
    return out


def gather_embeddings(local: torch.Tensor, n_cells: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """The one collective of an extraction run: every rank receives the [n_cells, dim] matrix, rank blocks in rank
    order (shard sizes differ by at most one cell: blocks are padded to the longest for the all-gather)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [len(sdist.contiguous_shard(n_cells, r, world)) for r in range(world)]
    longest = max(sizes)
    dim = local.shape[1]
    gloo = dist.get_backend(group) == "gloo"
    buf = torch.zeros((longest, dim), dtype=local.dtype, device="cpu" if gloo else local.device)
    buf[:local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0).to(local.device)
