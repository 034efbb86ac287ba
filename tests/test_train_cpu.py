"""The S1 training driver against the reference's own loop pieces (tests/golden/s1_loop.npz, written by
tools/make_golden.py from the reference's data_mask, scheduler and loss driven over a stand-in token model), and
its data-parallel bookkeeping on two gloo ranks against one process.  CPU only: masking and loss are injected as
torch formulations (the driver's defaults are the device kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import performer_oracle as O
from oracle import masking_oracle as MO
from scmomer_amd import train


def _torch_loss(logits, labels, pad_id):
    return torch.nn.functional.cross_entropy(logits.transpose(1, 2), labels, ignore_index=pad_id)


def _trainer(model, g, **kw):
    return train.S1Trainer(model, learning_rate=float(g["lr"]), grad_acc=int(g["grad_acc"]), mask_fn=MO.data_mask_tensor,
                           loss_fn=_torch_loss, **kw)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "s1_loop.npz"))


def test_schedule_matches_reference(g):
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = train.CosineAnnealingWarmupRestarts(opt, first_cycle_steps=15, cycle_mult=2, max_lr=1e-4, min_lr=1e-6,
                                              warmup_steps=5, gamma=0.9)
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(len(g["sched_lrs"]) - 1):
        sch.step()
        lrs.append(opt.param_groups[0]["lr"])
    np.testing.assert_allclose(lrs, g["sched_lrs"], rtol=1e-12, atol=0)
    state = sch.state_dict()
    for k in g["sched_state_keys"]:
        if k in ("_is_initial",):
            continue
        assert float(state[k]) == float(g["sched_state:" + k]), k
    for e, lr, cyc, pos, span in g["sched_jump"]:          # step(epoch) jumps, including the cycle boundaries
        o2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
        s2 = train.CosineAnnealingWarmupRestarts(o2, first_cycle_steps=15, cycle_mult=2, max_lr=1e-4, min_lr=1e-6,
                                                 warmup_steps=5, gamma=0.9)
        s2.step(int(e))
        assert (s2.cycle, s2.step_in_cycle, s2.cur_cycle_steps) == (int(cyc), int(pos), int(span))
        np.testing.assert_allclose(o2.param_groups[0]["lr"], lr, rtol=1e-12)
    with pytest.raises(AssertionError):
        train.CosineAnnealingWarmupRestarts(opt, first_cycle_steps=5, warmup_steps=5)


def test_schedule_state_round_trip():
    def make():
        o = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
        return o, train.CosineAnnealingWarmupRestarts(o, first_cycle_steps=15, cycle_mult=2, max_lr=1e-4, min_lr=1e-6,
                                                      warmup_steps=5, gamma=0.9)
    o1, s1_ = make()
    for _ in range(23):
        s1_.step()
    o2, s2 = make()
    s2.load_state_dict(s1_.state_dict())
    for _ in range(40):
        s1_.step()
        s2.step()
        assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"]


def test_loop_matches_reference_pieces(g, tmp_path):
    genes, bs, epochs = int(g["genes"]), int(g["batch_size"]), int(g["epochs"])
    train_tok, val_tok = O.standin_cells(7, 14, genes), O.standin_cells(8, 5, genes)
    model = O.standin_lm(99)
    torch.manual_seed(123)
    tr = _trainer(model, g)
    # torch's DataLoader draws one int64 (its worker base seed) from the global generator whenever an iterator is
    # created, i.e. once per training and once per validation pass of the reference loop; the fixture's mask draws
    # come after it.  The driver has no DataLoader, so the test consumes the same draw at the same moments.
    plain_batches = tr.batches_of

    def batches_like_dataloader(*a, **k):
        torch.empty((), dtype=torch.int64).random_()
        yield from plain_batches(*a, **k)

    tr.batches_of = batches_like_dataloader
    logs = []
    hist = tr.fit(train_tok, val_tok, epochs=epochs, batch_size=bs, ckpt_dir=str(tmp_path), model_name="m", log=logs.append)
    assert len(hist) == epochs
    np.testing.assert_allclose([h["train_loss"] for h in hist], g["train_loss"], rtol=2e-6)
    np.testing.assert_allclose([h["train_acc"] for h in hist], g["train_acc"], rtol=1e-6)
    np.testing.assert_allclose([h["val_loss"] for h in hist], g["val_loss"], rtol=2e-6)
    np.testing.assert_allclose([h["val_acc"] for h in hist], g["val_acc"], rtol=1e-9)
    np.testing.assert_allclose([h["lr"] for h in hist], g["lr_after_epoch"], rtol=1e-12)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["final:" + k], rtol=1e-5, atol=1e-6)
    st = tr.optimizer.state_dict()["state"]
    for k in st:
        np.testing.assert_allclose(st[k]["exp_avg"].numpy(), g[f"adam_exp_avg:{k}"], rtol=1e-4, atol=1e-7)
    assert any("model saved at epoch: 1" in m for m in logs)

    # checkpoint layout of the reference: module.-prefixed weights, optimizer, scheduler, epoch, loss; resume
    ck = torch.load(os.path.join(tmp_path, "m.pth"), map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "losses"}
    assert all(k.startswith("module.") for k in ck["model_state_dict"])
    assert ck["losses"] == min(h["val_loss"] for h in hist[:ck["epoch"]])
    fresh = O.standin_lm(1)
    tr2 = _trainer(fresh, g)
    start = train.load_checkpoint(os.path.join(tmp_path, "m.pth"), fresh, tr2.optimizer, tr2.scheduler)
    assert start == ck["epoch"] + 1
    for k, v in fresh.state_dict().items():
        assert torch.equal(v, ck["model_state_dict"]["module." + k])
    assert tr2.scheduler.last_epoch == ck["scheduler_state_dict"]["last_epoch"]
    # un-prefixed dictionaries (the layout of utils.save_ckpt, utils.py:9-22) load too
    train.load_checkpoint({"model_state_dict": model.state_dict(), "epoch": 4}, fresh)
    assert torch.equal(fresh[1].weight, model[1].weight)


def test_unfinished_accumulation_window_carries_over(g):
    """5 batches with grad_acc 2: the fifth micro-step's gradient stays in the accumulators (no optimizer step, no
    zeroing at the epoch boundary), exactly as the reference leaves it in .grad"""
    tok = O.standin_cells(7, 14, 64)
    model = O.standin_lm(99)
    torch.manual_seed(0)
    tr = _trainer(model, g)
    tr.train_epoch(tr.batches_of(tok, 3, 1))
    assert tr.index == 5
    assert any(float(a.abs().sum()) > 0 for a in tr.accs)
    assert all(p.grad is None for p in tr.params)


def test_early_stopping(g, tmp_path):
    tok = O.standin_cells(7, 6, 32)
    model = O.standin_lm(99)
    tr = train.S1Trainer(model, learning_rate=0.0, grad_acc=1, mask_fn=MO.data_mask_tensor, loss_fn=_torch_loss)
    seq = iter([1.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0, 2.0])
    tr.validate = lambda batches, n=None: (next(seq), 50.0)
    logs = []
    hist = tr.fit(tok, tok, epochs=8, batch_size=3, patience=2, ckpt_dir=str(tmp_path), log=logs.append)
    assert len(hist) == 4                                   # best at epoch 1, then patience 2 is exceeded at epoch 4
    assert "Early stopping triggered at epoch 4" in logs[-1]


def test_freeze_for_finetune_counts():
    import scmomer_amd as S
    m = S.PerformerLM(num_tokens=7, dim=200, depth=6, max_seq_len=64, heads=10, local_attn_heads=0, g2v_position_emb=True)
    total, trainable = train.freeze_for_finetune(m)
    per_layer = 2 * 200 + 3 * 640 * 200 + 200 * 640 + 200 + 2 * 200 + 800 * 200 + 800 + 200 * 800 + 200
    assert trainable == 2 * per_layer + 2 * 200
    assert not m.token_emb.weight.requires_grad and m.norm.weight.requires_grad
    assert total == sum(p.numel() for p in m.parameters())


# ---------------------------------------------------------------------------------------------- two ranks
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


N_CELLS = 48


def _worker(rank, world, port, out_dir, lr, grad_acc):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)                                # up to 8 ranks share this machine's cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = O.standin_lm(99 + rank)                     # different replicas: rank 0's weights must win
        tr = train.S1Trainer(model, learning_rate=lr, grad_acc=grad_acc, mask_fn=_fixed_mask, loss_fn=_torch_loss)
        tok, val = O.standin_cells(7, N_CELLS, 48), O.standin_cells(8, 5, 48)
        hist = tr.fit(tok, val, epochs=2, batch_size=2, ckpt_dir=os.path.join(out_dir, f"r{rank}"), log=lambda m: None)
        torch.save({"sd": model.state_dict(), "hist": hist}, os.path.join(out_dir, f"out{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _fixed_mask(data, mask_prob, replace_prob, pad_id, mask_id):
    """draws that depend only on the cells, so that one process and several ranks mask identically"""
    gen = torch.Generator().manual_seed(int(data.sum()) + 17 * data.shape[0])
    return MO.data_mask_tensor(data, mask_prob, replace_prob, pad_id, mask_id, rand_pick=torch.rand(data.shape, generator=gen),
                        rand_replace=torch.rand(data.shape, generator=gen))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_match_one_process_with_all_shards(tmp_path, world):
    """world_size 2, 4 and 8 over gloo (the S1 driver as one process per GPU runs it, s1_pretrain_rna.py:75-82,242):
    replicas end identical; the weights equal those of one process that runs every rank's micro-batches itself and
    averages (gradient mean over ranks == DDP); reported accuracies are percentages"""
    from scmomer_amd import dist as sdist
    lr, grad_acc = 0.02, 3
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), lr, grad_acc), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"out{r}.pt"), weights_only=False) for r in range(world)]
    for o in outs[1:]:
        for k in outs[0]["sd"]:
            assert torch.equal(outs[0]["sd"][k], o["sd"][k])
        assert outs[0]["hist"][0]["train_loss"] == o["hist"][0]["train_loss"]
        assert outs[0]["hist"][1]["val_acc"] == o["hist"][1]["val_acc"]
    for h in outs[0]["hist"]:
        assert 0.0 <= h["train_acc"] <= 100.0 and 0.0 <= h["val_acc"] <= 100.0
    assert os.path.exists(os.path.join(tmp_path, "r0", "pretrain_rna.pth"))
    assert not os.path.exists(os.path.join(tmp_path, "r1"))            # only the master writes checkpoints

    # one process: per optimizer step, the mean over all ranks' accumulated gradients
    model = O.standin_lm(99)
    tok = O.standin_cells(7, N_CELLS, 48)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    sch = train.CosineAnnealingWarmupRestarts(opt, first_cycle_steps=15, cycle_mult=2, max_lr=lr, min_lr=1e-6,
                                              warmup_steps=5, gamma=0.9)
    steps = 0
    micro = 0                                               # the driver's micro-step counter restarts every epoch
    for epoch in (1, 2):
        shards = [sdist.sampler_indices(len(tok), r, world, epoch=epoch) for r in range(world)]
        nb = len(shards[0]) // 2
        micro = 0
        for b in range(nb):
            for r in range(world):
                data = tok[shards[r][2 * b:2 * b + 2]]
                mi, lab = _fixed_mask(data, 0.15, 0.9, 6, 6)
                (_torch_loss(model(mi), lab, 6) / grad_acc / world).backward()
            micro += 1
            if micro % grad_acc == 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 100)
                opt.step()
                opt.zero_grad()
                steps += 1
        sch.step()
    assert steps >= 2                                       # every world size takes optimizer steps in this run
    for k, v in model.state_dict().items():
        torch.testing.assert_close(outs[0]["sd"][k], v, rtol=1e-5, atol=1e-6)


def test_window_step_equals_micro_steps_on_the_host_formulation(g):
    """window_step over 3 stacked micro-batches == 3 micro_step calls (stand-in model, torch masking and loss):
    same losses and accuracies per micro-batch, same accumulated gradients, same weights after the optimizer step;
    a window that would straddle the optimizer step is refused"""
    tok = O.standin_cells(7, 18, 64)
    outs = []
    for window in (1, 3):
        model = O.standin_lm(99)
        tr = train.S1Trainer(model, learning_rate=0.01, grad_acc=3, mask_fn=_fixed_mask, loss_fn=_torch_loss, window=window)
        tr.scheduler.step(), tr.scheduler.step()
        losses, accs = [], []
        for w in range(2):
            batch = [tok[6 * w + 2 * i:6 * w + 2 * i + 2] for i in range(3)]
            if window == 1:
                for b in batch:
                    l_, a_ = tr.micro_step(b)
                    losses.append(l_.reshape(1)), accs.append(a_.reshape(1))
            else:
                l_, a_ = tr.window_step(torch.cat(batch))
                assert tuple(l_.shape) == (3,)
                losses.append(l_), accs.append(a_)
        assert tr.index == 6
        outs.append((torch.cat(losses), torch.cat(accs), model.state_dict()))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-6, atol=0)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-6, atol=0)
    for k, v in outs[0][2].items():
        torch.testing.assert_close(outs[1][2][k], v, rtol=1e-5, atol=1e-7)
    tr = train.S1Trainer(O.standin_lm(99), learning_rate=0.01, grad_acc=4, mask_fn=_fixed_mask, loss_fn=_torch_loss, window=2)
    tr.micro_step(tok[:2])
    tr.window_step(torch.cat([tok[2:4], tok[4:6]]))
    with pytest.raises(ValueError):
        tr.window_step(torch.cat([tok[6:8], tok[8:10]]))    # micro-steps 4 and 5: the optimizer step lies between
    with pytest.raises(ValueError):
        train.S1Trainer(O.standin_lm(99), grad_acc=5, mask_fn=_fixed_mask, loss_fn=_torch_loss, window=2)


def test_train_epoch_runs_windows_and_the_ragged_tail_one_by_one(g):
    """train_epoch with window 2 over 7 batches (the last one smaller): three windows, then the tail as micro-steps;
    the epoch's numbers equal those of the same epoch run one micro-step at a time"""
    tok = O.standin_cells(7, 20, 64)
    res = []
    for window in (1, 2):
        model = O.standin_lm(99)
        tr = train.S1Trainer(model, learning_rate=0.01, grad_acc=2, mask_fn=_fixed_mask, loss_fn=_torch_loss, window=window)
        tr.scheduler.step()
        res.append((tr.train_epoch(tr.batches_of(tok, 3, 1)), tr.index, model.state_dict()))
    assert res[0][1] == res[1][1] == 7
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6)
    for k, v in res[0][2].items():
        torch.testing.assert_close(res[1][2][k], v, rtol=1e-5, atol=1e-7)
