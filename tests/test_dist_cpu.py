"""CPU (gloo, world_size 2) checks of the data-parallel path: cell sharding with DistributedSampler semantics and
the once-per-optimizer-step flat gradient average, against a single process that sees all cells."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DistributedSampler

from scmomer_amd import dist as sdist


class _Sized:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


@pytest.mark.parametrize("n,world", [(10, 2), (11, 2), (3, 4), (100, 8), (1, 2)])
@pytest.mark.parametrize("shuffle", [True, False])
@pytest.mark.parametrize("drop_last", [True, False])
def test_sampler_indices_match_distributed_sampler(n, world, shuffle, drop_last):
    for r in range(world):
        ref = DistributedSampler(_Sized(n), num_replicas=world, rank=r, shuffle=shuffle, seed=2024, drop_last=drop_last)
        ref.set_epoch(5)
        assert list(iter(ref)) == sdist.sampler_indices(n, r, world, shuffle=shuffle, seed=2024, epoch=5,
                                                        drop_last=drop_last)


def test_contiguous_shards_partition_the_cells():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            parts = [sdist.contiguous_shard(n, r, world) for r in range(world)]
            assert sum(len(p) for p in parts) == n
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            flat = [parts[0].start] + [p.start for p in parts[1:]]
            assert all(parts[i].stop == parts[i + 1].start for i in range(world - 1)) and flat[0] == 0
    with pytest.raises(ValueError):
        sdist.contiguous_shard(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(11)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 7))


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(12, 16, generator=g), torch.randint(0, 7, (12,), generator=g)


def _worker(rank, world, port, grad_acc, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        if rank == 1:                                   # replicas must start identical: rank 0's weights win
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        sdist.broadcast_parameters(list(model.parameters()), src=0)
        x, y = _data()
        idx = sdist.sampler_indices(len(x), rank, world, shuffle=False)
        avg = sdist.GradientAverager(model.parameters())
        loss_fn = torch.nn.CrossEntropyLoss()
        per = len(idx) // grad_acc
        for a in range(grad_acc):                       # accumulation micro-steps: no communication
            sel = idx[a * per:(a + 1) * per]
            (loss_fn(model(x[sel]), y[sel]) / grad_acc).backward()
        avg.average_()                                  # the one collective of the optimizer step
        torch.save([p.grad.clone() for p in model.parameters()], os.path.join(out_dir, f"g{rank}.pt"))
        t = sdist.max_over_ranks(1.0 + rank, "cpu")
        assert t == float(world)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grad_acc", [1, 3])
def test_gradient_average_world2_matches_single_process(tmp_path, grad_acc):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), grad_acc, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                        # every replica holds the same averaged gradient
    # single process over the same micro-batches: mean over ranks of (mean over micro-steps of batch-mean loss)
    model = _model()
    x, y = _data()
    loss_fn = torch.nn.CrossEntropyLoss()
    for r in range(world):
        idx = sdist.sampler_indices(len(x), r, world, shuffle=False)
        per = len(idx) // grad_acc
        for a in range(grad_acc):
            sel = idx[a * per:(a + 1) * per]
            (loss_fn(model(x[sel]), y[sel]) / grad_acc / world).backward()
    for p, g in zip(model.parameters(), g0):
        torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-7)


def test_single_process_is_a_no_op():
    model = _model()
    x, y = _data()
    torch.nn.functional.cross_entropy(model(x), y).backward()
    before = [p.grad.clone() for p in model.parameters()]
    sdist.GradientAverager(model.parameters()).average_()
    for b, p in zip(before, model.parameters()):
        assert torch.equal(b, p.grad)
    assert sdist.max_over_ranks(2.5, "cpu") == 2.5


def _redraw_worker(rank, world, port, out_dir):
    """Each rank owns a replica whose generator is seeded differently; after a redraw every replica must hold
    rank 0's matrices (what DDP's buffer broadcast guarantees in the reference, s1_pretrain_rna.py:242)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scmomer_amd import performer as P
        from scmomer_amd import train

        class Wrap(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.performer = P.Performer(200, 2, 10, 64, feature_redraw_interval=3, qkv_bias=False)

        torch.manual_seed(100 + rank)                   # replicas start with DIFFERENT projections
        model = Wrap()
        trainer = train.S1Trainer(model, grad_acc=1)    # construction replicates rank 0's weights and buffers
        perf = model.performer
        before = [m.projection_matrix.clone() for m in perf.modules() if hasattr(m, "projection_matrix")]
        model.train()
        torch.manual_seed(200 + rank)                   # ... and would draw DIFFERENT new ones
        for _ in range(3):
            perf.check_redraw_projections()             # ticks 0 -> 3
        assert int(perf.calls_since_last_redraw) == 3
        perf.check_redraw_projections()                 # redraw + broadcast from rank 0
        after = [m.projection_matrix.clone() for m in perf.modules() if hasattr(m, "projection_matrix")]
        torch.save({"before": before, "after": after, "calls": int(perf.calls_since_last_redraw)},
                   os.path.join(out_dir, f"omega{rank}.pt"))
        assert trainer.world == world
    finally:
        dist.destroy_process_group()


def test_redrawn_projections_stay_identical_across_ranks(tmp_path):
    world = 2
    mp.spawn(_redraw_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "omega0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "omega1.pt"))
    assert r0["calls"] == 0 and r1["calls"] == 0
    assert len(r0["after"]) == 2
    for a0, a1, b0, b1 in zip(r0["after"], r1["after"], r0["before"], r1["before"]):
        assert torch.equal(b0, b1)                      # replicated at construction
        assert torch.equal(a0, a1)                      # and again after the redraw: rank 0's draw everywhere
        assert not torch.equal(a0, b0)                  # the redraw did happen
    # rank 0's draw is what a single process with rank 0's seed draws
    from scmomer_amd import performer as P
    torch.manual_seed(200)
    want = P.gaussian_orthogonal_random_matrix(266, 64)
    assert torch.equal(r1["after"][0], want)


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scmomer_amd import extract
        n_cells = 7                                     # shards of 4 and 3 cells: the blocks are padded for the gather
        shard = sdist.contiguous_shard(n_cells, rank, world)
        local = torch.arange(shard.start, shard.stop, dtype=torch.float32)[:, None] * torch.ones(1, 5) + 0.5 * rank
        full = extract.gather_embeddings(local, n_cells)
        torch.save(full, os.path.join(out_dir, f"full{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_extraction_gather_keeps_cell_order_with_uneven_shards(tmp_path):
    world = 2
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    f0 = torch.load(os.path.join(tmp_path, "full0.pt"))
    f1 = torch.load(os.path.join(tmp_path, "full1.pt"))
    assert torch.equal(f0, f1) and f0.shape == (7, 5)
    want = torch.arange(7, dtype=torch.float32) + torch.tensor([0, 0, 0, 0, 0.5, 0.5, 0.5])
    assert torch.equal(f0[:, 0], want) and torch.equal(f0[:, 4], want)
