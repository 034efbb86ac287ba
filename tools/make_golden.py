#!/usr/bin/env python3
"""Generate golden vectors under tests/golden/ by running the reference itself.

Runs ONLY in the build container (reads /root/reference at run time; never on the
GPU box).  Only inputs and outputs are written - no reference source travels.
Parameters come from ``oracle.performer_oracle.make_params`` (numpy PCG64 draws),
so the fixtures stay small: they hold the seed, the projection matrices
(QR output is LAPACK-dependent, so it is stored, not re-derived), the tokens and
the reference's outputs / gradients.

Import shims: the reference imports three things that are absent here and unused
on this path - the ``local_attention`` package (only instantiated for local heads),
the hard-coded gene2vec ``.npy`` path, and (for the masking script) the data-ingest
module.  They are replaced by inert stand-ins so the reference's own arithmetic runs.

usage: python tools/make_golden.py [outdir]
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import performer_oracle as O  # noqa: E402


def import_reference():
    sys.modules["local_attention"] = types.SimpleNamespace(LocalAttention=object)
    sys.modules["model.atac_encoder"] = types.SimpleNamespace(ViTModel=object, MAEConfig=object)
    sys.modules["model.read_data"] = types.SimpleNamespace(prepare_train_val_data=None)
    sys.path.insert(0, REF)
    state = {"rows": 16906, "dim": 200}
    np_load = np.load

    def _load(path, *a, **k):
        if "gene2vec" in str(path):
            return np.zeros((state["rows"], state["dim"]))
        return np_load(path, *a, **k)

    np.load = _load
    import model.performer.performer_pytorch as pp
    import model.rna_encoder as re_
    import s1_pretrain_rna as s1
    return pp, re_, s1, state


def sample_big(a: np.ndarray, limit=4096):
    """Full array if small, else a strided sample plus reductions."""
    flat = a.reshape(-1)
    if flat.size <= limit:
        return {"full": flat.astype(np.float32)}
    step = flat.size // 2048
    return {"stride": np.int64(step), "sample": flat[::step][:2048].astype(np.float32),
            "l2": np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum())),
            "sum": np.float64(flat.astype(np.float64).sum())}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    torch.set_num_threads(8)
    pp, re_, s1, state = import_reference()

    # ---- (i)/(ii) feature map and linear attention --------------------------------------
    # inputs are re-derived from the seed by the tests (numpy PCG64), only outputs are stored
    gen = torch.Generator().manual_seed(11)
    proj = O.gaussian_orthogonal_random_matrix(266, 64, gen)
    data, vv, _, _ = O.attention_inputs(11, (2, 2, 67, 64), scale=1.5)
    qf = pp.softmax_kernel(data, projection_matrix=proj, is_query=True)
    kf = pp.softmax_kernel(data, projection_matrix=proj, is_query=False)
    la = pp.linear_attention(qf, kf, vv)
    np.savez_compressed(os.path.join(out, "feature_map.npz"), seed=np.int64(11), shape=np.array((2, 2, 67, 64)),
                        scale=np.float64(1.5), proj=proj.numpy(),
                        q_features=qf.numpy(), k_features=kf.numpy(), linear_attention=la.numpy())

    # ---- (iii) FastAttention forward + gradients ---------------------------------------
    fa = pp.FastAttention(64)
    fa.projection_matrix.copy_(proj)
    q, k, v, cot = O.attention_inputs(12, (2, 5, 193, 64), scale=1.0)
    q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
    o = fa(q, k, v)
    (o * cot).sum().backward()
    np.savez_compressed(os.path.join(out, "fast_attention.npz"), seed=np.int64(12), shape=np.array((2, 5, 193, 64)),
                        scale=np.float64(1.0), proj=proj.numpy(), out=o.detach().numpy(),
                        dq=q.grad.numpy(), dk=k.grad.numpy(), dv=v.grad.numpy())

    # ---- (v)/(vi) PerformerLM, both heads, logits + encodings + gradients ----------------
    for name, n, depth, head, bsz in (("lm_linear_n131_d2", 131, 2, "linear", 2),
                                     ("lm_conv_n300_d1", 300, 1, "conv", 3)):
        seed = 100 + n
        p = O.make_params(seed, depth=depth, max_seq_len=n, head=head)
        state["rows"] = n
        if head == "linear":
            m = pp.PerformerLM(num_tokens=7, dim=200, depth=depth, max_seq_len=n, heads=10,
                               local_attn_heads=0, g2v_position_emb=True)
        else:
            m = pp.PerformerLM(num_tokens=7, dim=200, depth=depth, max_seq_len=n, heads=10,
                               local_attn_heads=0, g2v_position_emb=True)
            m.to_out = re_.Identity()
        missing = m.load_state_dict(p, strict=True)
        m.fix_projection_matrices_()
        m.train()
        r2 = np.random.default_rng(seed + 1)
        tokens = torch.from_numpy(r2.integers(0, 7, (bsz, n)).astype(np.int64))
        enc = m(tokens, return_encodings=True)
        res = m(tokens)
        if head == "linear":
            labels = torch.from_numpy(r2.integers(0, 7, (bsz, n)).astype(np.int64))
            labels[r2.random((bsz, n)) < 0.8] = 6
            loss = torch.nn.CrossEntropyLoss(ignore_index=6, reduction="mean")(res.transpose(1, 2), labels)
        else:
            labels = torch.zeros(1, dtype=torch.int64)
            cot = torch.from_numpy(r2.standard_normal(tuple(res.shape)).astype(np.float32))
            loss = (res * cot).sum()
        loss.backward()
        fx = {"seed": np.int64(seed), "tokens": tokens.numpy(), "labels": labels.numpy(),
              "out": res.detach().numpy(), "encodings": enc.detach().numpy(), "loss": np.float64(loss.item())}
        if head == "conv":
            fx["cot"] = cot.numpy()
        for kname, val in p.items():
            if kname.endswith("projection_matrix"):
                fx["param:" + kname] = val.numpy()
        for kname, par in m.named_parameters():
            if par.grad is None:
                continue
            for kk, vv_ in sample_big(par.grad.numpy()).items():
                fx[f"grad:{kname}:{kk}"] = vv_
        np.savez_compressed(os.path.join(out, name + ".npz"), **fx)

    # ---- (vii) masking and binning --------------------------------------------------------
    r3 = np.random.default_rng(7)
    cells = O.synthetic_cells(7, 3, 2000)
    cells[0, :5] = [0.5, 1.5, 2.5, 5.6, -0.2]
    ds = s1.SCDataset(types.SimpleNamespace(), 7)
    s1_bins = []
    for i in range(cells.shape[0]):
        holder = types.SimpleNamespace(toarray=lambda i=i: cells[i:i + 1].copy())
        ds.data = {i: holder}
        s1_bins.append(ds[i].numpy())
    s1_bins = np.stack(s1_bins)
    sc_bins = torch.round(torch.from_numpy(cells)).long().clamp(min=0, max=5).numpy()
    data = torch.from_numpy(s1_bins)
    torch.manual_seed(77)
    masked, labels = s1.data_mask(data, 0.15, 0.9, 6, 6)
    torch.manual_seed(77)
    rand_pick = torch.rand(tuple(data.shape))
    rand_replace = torch.zeros_like(data).float().uniform_(0, 1)
    np.savez_compressed(os.path.join(out, "mask_bin.npz"), cells=cells, s1_bins=s1_bins, scmomer_bins=sc_bins,
                        rand_pick=rand_pick.numpy(), rand_replace=rand_replace.numpy(),
                        masked_input=masked.numpy(), labels=labels.numpy())

    # ---- (viii) projection-matrix construction under a fixed torch seed -------------------
    torch.manual_seed(2024)
    ref_proj = pp.gaussian_orthogonal_random_matrix(266, 64)
    np.savez_compressed(os.path.join(out, "projection.npz"), seed=np.int64(2024), proj=ref_proj.numpy())

    # ---- (x) scMomer.get_cell_embedding, RNA-only configuration, full gene count ---------------
    # (scmomer.py hard-codes 16906 genes and depth 6; the student encoder is the one of extract_cell_emb.py)
    import model.scmomer as sm
    seed = 4242
    state["rows"] = 16906
    p = O.make_scmomer_params(seed)
    enc = torch.nn.Module()
    enc.encoder = torch.nn.Sequential(torch.nn.Linear(16906, 256), torch.nn.LayerNorm(256), torch.nn.GELU(),
                                      torch.nn.Dropout(0.1), torch.nn.Linear(256, 128))
    enc.forward = lambda x: enc.encoder(x)
    cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
    model = sm.scMomer(cfg, atac_config=None, rna_decoder=None, atac_decoder=None, encoder=enc)
    model.load_state_dict(p, strict=True)
    model.eval()
    cells = torch.from_numpy(O.synthetic_cells(seed, 2, 16906))
    with torch.no_grad():
        emb = model.get_cell_embedding(cells)
        per_gene = model.rna_model(torch.round(cells).long().clamp(min=0, max=5))
    fx = {"seed": np.int64(seed), "embedding": emb.numpy(), "per_gene_sample": per_gene.numpy()[:, ::97],
          "per_gene_l2": np.float64(per_gene.double().norm().item())}
    for kname, val in p.items():
        if kname.endswith("projection_matrix"):
            fx["param:" + kname] = val.numpy()
    np.savez_compressed(os.path.join(out, "scmomer_embed.npz"), **fx)

    # ---- (xi) the S1 loop: schedule, accumulation, clipping, Adam, accuracy, checkpoint layout ------------------
    # The loop lives inside main() of the reference script, so it is driven here statement by statement with the
    # reference's own pieces (data_mask, CosineAnnealingWarmupRestarts, the loss, the samplers of torch) on a
    # stand-in token model; single process (DDP over one rank is the identity).
    import utils as ref_utils
    from torch.utils.data import DataLoader, DistributedSampler
    sched_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sched = ref_utils.CosineAnnealingWarmupRestarts(sched_opt, first_cycle_steps=15, cycle_mult=2, max_lr=1e-4,
                                                    min_lr=1e-6, warmup_steps=5, gamma=0.9)
    lrs = [sched_opt.param_groups[0]["lr"]]
    for _ in range(120):
        sched.step()
        lrs.append(sched_opt.param_groups[0]["lr"])
    sched_state = {k: v for k, v in sched.state_dict().items() if isinstance(v, (int, float))}
    jump = []
    for e in (3, 14, 15, 16, 44, 45, 46, 104, 105, 200):
        o2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
        s2 = ref_utils.CosineAnnealingWarmupRestarts(o2, first_cycle_steps=15, cycle_mult=2, max_lr=1e-4, min_lr=1e-6,
                                                     warmup_steps=5, gamma=0.9)
        s2.step(e)
        jump.append([e, o2.param_groups[0]["lr"], s2.cycle, s2.step_in_cycle, s2.cur_cycle_steps])

    grad_acc, batch_size, epochs, genes = 2, 3, 3, 64
    train_tok, val_tok = O.standin_cells(7, 14, genes), O.standin_cells(8, 5, genes)
    model = O.standin_lm(99)
    torch.manual_seed(123)
    train_sampler = DistributedSampler(train_tok, num_replicas=1, rank=0)
    val_sampler = DistributedSampler(val_tok, num_replicas=1, rank=0)
    train_loader = DataLoader(train_tok, batch_size=batch_size, sampler=train_sampler)
    val_loader = DataLoader(val_tok, batch_size=batch_size, sampler=val_sampler)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4 * 300)    # a visible step size for a 3-epoch fixture
    scheduler = ref_utils.CosineAnnealingWarmupRestarts(optimizer, first_cycle_steps=15, cycle_mult=2,
                                                        max_lr=1e-4 * 300, min_lr=1e-6, warmup_steps=5, gamma=0.9)
    loss_fn = torch.nn.CrossEntropyLoss(ignore_index=6, reduction="mean")
    softmax = torch.nn.Softmax(dim=-1)
    rec = {"train_loss": [], "train_acc": [], "val_loss": [], "val_acc": [], "lr_after_epoch": []}
    for i in range(1, epochs + 1):
        train_loader.sampler.set_epoch(i)
        model.train()
        running_loss = cum_acc = 0.0
        for index, data in enumerate(train_loader):
            index += 1
            masked_input, labels = s1.data_mask(data, 0.15, 0.9, 6, 6)
            logits = model(masked_input)
            loss = loss_fn(logits.transpose(1, 2), labels) / grad_acc
            loss.backward()
            if index % grad_acc == 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), int(1e2))
                optimizer.step()
                optimizer.zero_grad()
            running_loss += loss.item()
            final = softmax(logits)[..., 1:-1].argmax(dim=-1) + 1
            pred_num = (labels != 6).sum(dim=-1)
            correct_num = ((labels != 6) * (final == labels)).sum(dim=-1)
            cum_acc += torch.true_divide(correct_num, pred_num).mean().item()
        rec["train_loss"].append(running_loss / len(train_loader) * grad_acc)
        rec["train_acc"].append(100 * cum_acc / len(train_loader))
        scheduler.step()
        rec["lr_after_epoch"].append(optimizer.param_groups[0]["lr"])
        model.eval()
        val_running, preds, truths = 0.0, [], []
        with torch.no_grad():
            for val_index, data in enumerate(val_loader):
                val_index += 1
                masked_input, labels = s1.data_mask(data, 0.15, 0.9, 6, 6)
                logits = model(masked_input)
                val_running += loss_fn(logits.transpose(1, 2), labels).item()
                preds.append(softmax(logits)[..., 1:-1].argmax(dim=-1) + 1)
                truths.append(labels)
        preds, truths = torch.cat(preds), torch.cat(truths)
        rec["val_loss"].append(val_running / val_index)
        rec["val_acc"].append(100 * ((truths != 6) * (preds == truths)).sum().item() / (truths != 6).sum().item())
    fx = {k: np.asarray(v, dtype=np.float64) for k, v in rec.items()}
    fx.update({"sched_lrs": np.asarray(lrs, dtype=np.float64), "sched_jump": np.asarray(jump, dtype=np.float64),
               "sched_state_keys": np.asarray(sorted(sched_state)), "grad_acc": np.int64(grad_acc),
               "batch_size": np.int64(batch_size), "epochs": np.int64(epochs), "genes": np.int64(genes),
               "lr": np.float64(1e-4 * 300)})
    for k, v in sched_state.items():
        fx["sched_state:" + k] = np.float64(v)
    for k, v in model.state_dict().items():
        fx["final:" + k] = v.numpy()
    for k, v in optimizer.state_dict()["state"].items():
        fx[f"adam_exp_avg:{k}"] = v["exp_avg"].numpy()
    np.savez_compressed(os.path.join(out, "s1_loop.npz"), **fx)

    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
