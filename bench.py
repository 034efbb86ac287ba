#!/usr/bin/env python3
"""S1 RNA masked-autoencoding pretrain micro-step benchmark (3 cells x 16,906 genes per GPU per micro-step).

One step = mask 15 % of the expressed genes of a 3-cell batch (s1_pretrain_rna.py:290), run the
Performer RNA encoder forward (16,906 gene tokens, depth 6, dim 200, 10 heads, 266 features),
cross-entropy with ignore_index, backward; every `--grad-acc` steps the gradients are synchronised
(DDP over RCCL when --gpus > 1), clipped and applied with Adam, exactly as the reference's loop
(s1_pretrain_rna.py:294-305).  Cells are sharded across ranks; per-GPU work is fixed (weak scaling).

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (largest share of the step), timed
with HIP events on its launch stream; `cpu_baseline` is the CPU oracle (a port of the
reference path, validated against it) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_GENES = 16906
BATCH = 3
FLOP_FWD_BWD_PER_CELL = 853.50e9   # algorithmic GEMM-term FLOPs, forward+backward, one cell
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA
PEAK_HBM_GBS = 8000.0              # MI355X HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grad-acc", type=int, default=5, help="micro-steps per optimizer step (README uses 5)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--window", type=int, default=5,
                    help="micro-steps run as one launch chain (they share the weights of their accumulation window; "
                         "key stabiliser, loss mean and mask draws stay per 3-cell micro-batch); must divide "
                         "--grad-acc; 1 = one launch chain per micro-step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trace-all", action="store_true", help="time every kernel (per-kernel table to stderr)")
    ap.add_argument("--eager", action="store_true", help="drive every launch from Python instead of a HIP graph")
    ap.add_argument("--cells", type=int, default=3072, help="extract mode: cells per GPU in the synthetic cell list")
    ap.add_argument("--mode", choices=("train", "extract", "finetune", "s3"), default="train",
                    help="train: the S1 micro-step (the headline metric); extract: forward-only cell-embedding "
                         "extraction (scMomer.get_cell_embedding); finetune: the cell-type fine-tune step "
                         "(frozen encoder except its second-to-last layer, norm and head); s3: the distillation step with the "
                         "frozen encoder in train mode; secondary figures")
    return ap.parse_args()


def build_model(device):
    import scmomer_amd as S
    torch.manual_seed(2024)            # identical random-init replica on every rank
    model = S.PerformerLM(num_tokens=7, dim=200, depth=6, max_seq_len=N_GENES, heads=10, local_attn_heads=0,
                          g2v_position_emb=True)
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():              # synthetic gene2vec table (float64, last row zero as in the reference)
        model.pos_emb.emb.weight[:N_GENES] = torch.randn((N_GENES, 200), generator=gen, dtype=torch.float64)
    model.fix_projection_matrices_()   # feature redraw disabled for the measurement (fixed projections)
    return model.to(device)


def _cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    return model, share


def cpu_baseline(mode="train"):
    """The CPU oracle (a port of the reference path, pinned against the reference's own outputs) on this host, fp32,
    with every core this process may use: torch's own thread count, which follows OMP_NUM_THREADS / the affinity
    mask (on a shared GPU box that is this GPU's CPU share, not the machine's core count: oversubscribing a cgroup
    share slows the oracle down).  Timed: config 1 (one cell, forward, eval: 1 warm-up + 3 passes) and the metric's
    own workload (3 cells, forward + CE + backward: 1 warm-up + 2 timed passes, about 24 GB of host memory), about
    30 s in all.
    `value` is the rate of the workload the GPU line reports (train: B=3 forward+backward; extract: B=1 forward)."""
    from oracle import performer_oracle as O
    import numpy as np
    cpu_model, share = _cpu_info()
    threads = torch.get_num_threads()
    p = O.make_params(2024, depth=6, max_seq_len=N_GENES)
    cells = O.synthetic_cells(2024, BATCH, N_GENES)
    tokens = torch.from_numpy(np.minimum(cells, 5).astype(np.int64))
    with torch.no_grad():
        O.performer_lm_forward(tokens[:1, :2048], p, depth=6, heads=10)          # pages the math libraries in
        O.performer_lm_forward(tokens[:1], p, depth=6, heads=10)                 # warm-up at full length
        t0 = time.perf_counter()
        for _ in range(3):
            O.performer_lm_forward(tokens[:1], p, depth=6, heads=10)
        fwd = (time.perf_counter() - t0) / 3
    out = {"unit": "cells/s", "cores": threads, "kind": "port", "cpu_model": cpu_model, "cpus_available": share,
           "cpus_in_machine": os.cpu_count(), "forward_b1_cells_per_s": 1.0 / fwd, "forward_b1_seconds": fwd}
    if mode == "extract":
        out.update(value=1.0 / fwd, sample="config 1: 1 cell (B=1), encoder forward at 16906 tokens, eval, fp32 "
                                           f"torch-CPU oracle, mean of 3 timed passes ({fwd:.2f} s each)")
        return out
    leaves = {k: v.clone().requires_grad_() for k, v in p.items()
              if v.dtype == torch.float32 and not k.endswith("projection_matrix")}
    pp = dict(p)
    pp.update(leaves)
    labels = tokens.clone()
    labels[torch.rand(tokens.shape) > 0.15] = 6
    times = []
    for it in range(3):                                      # 1 warm-up pass + 2 timed passes
        for v in leaves.values():
            v.grad = None
        t0 = time.perf_counter()
        loss = O.masked_ce_loss(O.performer_lm_forward(tokens, pp, depth=6, heads=10), labels, 6)
        loss.backward()
        times.append(time.perf_counter() - t0)
        del loss
    dt = sum(times[1:]) / 2
    out.update(value=BATCH / dt, fwd_bwd_b3_seconds=dt, fwd_bwd_b3_warmup_seconds=times[0],
               sample=f"config 2's workload: 3 cells (B=3), forward + CE + backward at 16906 tokens, fp32 torch-CPU "
                      f"oracle, 1 warm-up + 2 timed passes ({dt:.1f} s each); config 1 (B=1 forward, eval): "
                      f"{fwd:.2f} s per cell")
    return out


# trace id -> kernel symbol as rocprofv3 lists it (for the counter-derived traffic of the dominant kernel)
TRACE_SYMBOL = {"favor_apply_fwd": "favor_apply_kernel<0>", "favor_apply_bwd_q": "favor_apply_kernel<1>",
                "favor_apply_bwd_k": "favor_apply_kernel<2>", "favor_ctx_fwd": "favor_ctx_kernel<0>",
                "favor_ctx_bwd": "favor_ctx_kernel<1>", "wgrad": "wgrad_dma_kernel<false>",
                "wgrad/bias": "wgrad_dma_kernel<true>", "gemm_nt/f32_bias_res": "gemm_nt_dma_kernel<1>",
                "gemm_nt/f32": "gemm_nt_dma_kernel<4>", "gemm_skinny/bf16": "gemm_skinny_kernel<false, 0>",
                "gemm_skinny/gelu_dual": "gemm_skinny_kernel<false, 2>", "gemm_skinny/dgelu": "gemm_skinny_kernel<false, 3>",
                "ffn_chain_fwd": "ffn_chain_kernel", "gemm_ln_bwd": "gemm_nt_dma_kernel<6>",
                "gemm_res_ln": "gemm_nt_dma_kernel<7>"}
HBM_ACHIEVABLE_GBS = 6290.0   # float4 streaming copy on this part (measured ceiling; the datasheet peak is 8000)


def measured_utilisation(kernel, pattern="r*_pmc_mfma_valu_summary.txt"):
    """(matrix-core utilisation, vector-ALU busy share) of `kernel` from the committed rocprofv3 counter passes
    (SQ_VALU_MFMA_BUSY_CYCLES and SQ_ACTIVE_INST_VALU over kernel cycles x SIMDs), or (None, None)."""
    import glob
    sym = TRACE_SYMBOL.get(kernel)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not sym or not files:
        return None, None, None
    for ln in open(files[-1]).read().splitlines():
        if ln.startswith(sym + " ") or ln.startswith(sym[:44] + " "):
            cols = ln[len(sym):].split()
            try:
                return float(cols[2]), float(cols[3]), os.path.basename(files[-1])
            except (IndexError, ValueError):
                break
    return None, None, None


def measured_traffic(kernel, pattern="r*_pmc_fetch_write_summary.txt"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes (profiles/rNN_pmc_fetch_write_
    summary.txt: separate FETCH_SIZE and WRITE_SIZE passes of this same bench, mean per dispatch, in KB).  On gfx950
    FETCH_SIZE counts wide coalesced reads at half their bytes, so it is doubled; WRITE_SIZE is exact.  None when no
    summary covers the kernel."""
    import glob
    import re
    sym = TRACE_SYMBOL.get(kernel)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not sym or not files:
        return None, None
    lines = open(files[-1]).read().splitlines()
    for i, ln in enumerate(lines[:-1]):
        if ln.strip() == sym:
            f = re.search(r"FETCH_SIZE=([0-9.e+]+)", lines[i + 1])
            w = re.search(r"WRITE_SIZE=([0-9.e+]+)", lines[i + 1])
            if f and w:
                fetch_scale = 2.0
                return (fetch_scale * float(f.group(1)) + float(w.group(1))) * 1024.0, os.path.basename(files[-1])
    return None, None


def run_extract(args, world, rank, device):
    """Config 4: batched embedding extraction (evaluate_embedding.py:173-178: scMomer.get_cell_embedding over a cell
    list, scmomer.py:96-113) sharded over the ranks.  The synthetic cell list has --cells cells PER GPU (weak
    scaling; 1M cells over 8 GPUs is 125,000 each); rank r owns one contiguous shard, walks it in batches of
    --batch cells (24 by default: one full round of attention workgroups) and keeps its [cells, 128] block on the
    device; the only collective is the final gather.  The expression rows come from a pool of distinct synthetic
    cells resident in HBM (cell i = pool row i mod pool size): a 1M-cell float matrix is 68 GB and would be
    streamed from host memory or disk in production, which is not what this line measures.  One step = one batch;
    the timed region is the whole job including the gather."""
    import types
    import torch.distributed as dist
    import scmomer_amd as S
    from scmomer_amd import dist as sdist
    from scmomer_amd import extract, ops
    batch = args.batch if args.batch != BATCH else 24
    n_cells = world * args.cells
    torch.manual_seed(2024)
    cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
    model = S.scMomer(cfg, atac_config=None, encoder=S.StudentEncoder(N_GENES, 128))
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        model.rna_model.model.pos_emb.emb.weight[:N_GENES] = torch.randn((N_GENES, 200), generator=gen, dtype=torch.float64)
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad = False
    pool_n = 8 * batch
    gen = torch.Generator().manual_seed(100)
    pool = ((torch.rand((pool_n, N_GENES), generator=gen) < 0.15) *
            torch.distributions.Gamma(2.0, 1 / 1.2).sample((pool_n, N_GENES))).float().to(device)

    def source(lo, hi):
        return pool[torch.arange(lo, hi, device=device) % pool_n]

    shard = sdist.contiguous_shard(n_cells, rank, world)
    steps = (len(shard) + batch - 1) // batch
    extract.extract_embeddings(model, source, min(n_cells, world * batch * max(args.warmup, 1)), batch, rank=rank,
                               world=world, device=device)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local = extract.extract_embeddings(model, source, n_cells, batch, rank=rank, world=world, device=device)
    full = extract.gather_embeddings(local, n_cells)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = sdist.max_over_ranks(time.perf_counter() - t0, device)
    assert tuple(full.shape) == (n_cells, 128)
    ops.trace_collect()
    ops.trace_enable("all")
    with torch.no_grad():
        model.get_cell_embedding(source(0, batch))
    torch.cuda.synchronize()
    ops.trace_enable(None)
    trace = ops.trace_collect()
    if rank == 0:
        value = n_cells / dt
        dominant = max(trace, key=lambda k: trace[k][0])
        ms, fl, cnt, by = trace[dominant]
        gbs = by / (ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dominant, "r*_pmc_fetch_write_extract_summary.txt")
        kernels = {k: {"ms_per_step": round(m_, 4), "launches_per_step": c_, "gbs": round(b_ / (m_ * 1e-3) / 1e9, 1)}
                   for k, (m_, f_, c_, b_) in sorted(trace.items(), key=lambda kv: -kv[1][0])}
        base = cpu_baseline("extract") if world == 1 and not args.no_cpu_baseline else None
        print(json.dumps({
            "metric": "cells/sec, Performer RNA encoder forward (cell-embedding extraction, 16.9k genes)",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "scMomer.get_cell_embedding over a sharded cell list (RNA-only configuration, "
                                   "frozen encoder), final gather of [cells, 128] included",
                       "cells_total": n_cells, "cells_per_gpu": args.cells, "cells_per_batch": batch,
                       "gene_tokens": N_GENES, "depth": 6, "dim": 200, "heads": 10, "nb_features": 266,
                       "parallelism": f"dp{world}"},
            "embedding_norm": float(full.float().norm(dim=-1).mean().item()),
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": ms / cnt, "algorithmic_bytes_per_launch": by / cnt,
                         "whole_step_frac_of_mfma_peak": value / world * 307.52e9 / 1e12 / PEAK_BF16_TFLOPS},
            "cpu_baseline": base, "kernels": kernels}))


def run_s3(args, world, rank, device):
    """Config 5: the S3 distillation step (s3_pretrain_missing.py:286-316): scMomer(mode='one', reconstruct=True)
    in train() mode with the Performer frozen (forward only, no tape), trainable student encoder, to_out and linear
    decoders; BCE + MSE + distillation loss against precomputed teacher embeddings; backward through the tail,
    gradient average over the ranks, clip 1e4, Adam.  One step = one batch of 3 cells (the script's default)."""
    import types
    import torch.distributed as dist
    import scmomer_amd as S
    from scmomer_amd import dist as sdist
    from scmomer_amd import s3
    batch, n_peaks = args.batch, 20000
    torch.manual_seed(2024)
    cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
    model = S.scMomer(cfg, atac_config=None, rna_decoder=s3.ReconstructNet(N_GENES, 128),
                      atac_decoder=s3.ReconstructNet(n_peaks, 128), encoder=S.StudentEncoder(N_GENES, 128))
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        model.rna_model.model.pos_emb.emb.weight[:N_GENES] = torch.randn((N_GENES, 200), generator=gen, dtype=torch.float64)
    s3.freeze_for_distillation(model)
    model.rna_model.model.fix_projection_matrices_()
    model = model.to(device).train()
    window = args.window if args.window else 1
    if args.grad_acc % window:
        raise SystemExit("--window must divide --grad-acc")
    stepper = s3.S3Step(model, lr=1e-4, grad_acc=args.grad_acc, pos_weight=1.0, distill_weight=1.0,
                        use_graph=not args.eager)
    gen = torch.Generator().manual_seed(100 + rank)
    cells = batch * window
    data = []
    for _ in range(2):
        rna = ((torch.rand((cells, N_GENES), generator=gen) < 0.15) *
               torch.distributions.Gamma(2.0, 1 / 1.2).sample((cells, N_GENES))).float().to(device)
        atac = (torch.rand((cells, n_peaks), generator=gen) < 0.05).float().to(device)
        latent = torch.randn((cells, 128), generator=gen).to(device)
        data.append((rna, atac, latent))
    one = (lambda d: stepper.micro_step(*d)) if window == 1 else (lambda d: stepper.window_step(*d, window))
    for i in range(max(args.warmup, 2 * args.grad_acc // window)):    # past torch's first optimizer step
        losses = one(data[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = one(data[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = sdist.max_over_ranks(time.perf_counter() - t0, device)
    if rank == 0:
        value = world * cells * args.steps / dt
        print(json.dumps({
            "metric": "cells/sec, S3 distillation step (frozen Performer forward + trainable tail fwd/bwd, 16.9k genes)",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "scMomer.forward(mode='one', reconstruct=True), train mode, frozen encoder; BCE + "
                                   "MSE + distillation loss; backward; clip + Adam every grad_acc micro-steps",
                       "cells_per_gpu_per_step": cells, "micro_batch_cells": batch,
                       "micro_steps_per_launch_chain": window, "gene_tokens": N_GENES, "atac_peaks": n_peaks, "depth": 6,
                       "dim": 200, "heads": 10, "nb_features": 266, "grad_acc": args.grad_acc,
                       "parallelism": f"dp{world}", "launch": "eager" if args.eager else "graph replay"},
            "final_losses": {k: [float(x) for x in v.reshape(-1)] for k, v in losses.items()},
            "redraw_counter": int(model.rna_model.model.performer.calls_since_last_redraw),
            "roofline": {"whole_step_frac_of_mfma_peak": value / world * 307.52e9 / 1e12 / PEAK_BF16_TFLOPS},
            "cpu_baseline": None}))


def run_finetune(args, world, rank, device):
    """Cell-type fine-tune step of train_celltype.py:126-175: scMomer.forward(mode='one') with a classifier head,
    cross entropy, backward (only the encoder's final norm, its second-to-last layer, the per-gene head,
    rna_projection and the classifier are trainable: the backward stops below that layer), clip 5, Adam.  One step
    = one batch of 6 cells (the script's default)."""
    import types
    import torch.distributed as dist
    import scmomer_amd as S
    from scmomer_amd import dist as sdist
    from scmomer_amd import stack
    batch = args.batch if args.batch != BATCH else 6
    torch.manual_seed(2024)
    cfg = types.SimpleNamespace(projection_dim=128, normalize=True)
    model = S.scMomer(cfg, atac_config=None, encoder=S.StudentEncoder(N_GENES, 128))
    model.sub_task = torch.nn.Sequential(torch.nn.Linear(128, 64), torch.nn.Dropout(0.2), torch.nn.ReLU(),
                                         torch.nn.Linear(64, 20))
    for p in model.parameters():
        p.requires_grad = False
    lm = model.rna_model.model
    for group in (lm.norm, lm.performer.net.layers[-2], lm.to_out, model.rna_projection, model.sub_task):
        for p in group.parameters():
            p.requires_grad = True
    lm.fix_projection_matrices_()
    model = model.to(device).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    averager = sdist.GradientAverager(params)
    gen = torch.Generator().manual_seed(100 + rank)
    cells = [((torch.rand((batch, N_GENES), generator=gen) < 0.15) *
              torch.distributions.Gamma(2.0, 1 / 1.2).sample((batch, N_GENES))).float().to(device) for _ in range(2)]
    labels = torch.randint(0, 20, (batch,), generator=gen).to(device)

    def step(i):
        logits = model(None, cells[i % 2], mode='one', reconstruct=False)
        loss = torch.nn.functional.cross_entropy(logits, labels)
        opt.zero_grad()
        loss.backward()
        averager.average_()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return loss

    for i in range(max(args.warmup, 1)):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = sdist.max_over_ranks(time.perf_counter() - t0, device)
    if rank == 0:
        print(json.dumps({
            "metric": "cells/sec, cell-type fine-tune step (frozen encoder below its second-to-last layer, 16.9k genes)",
            "value": world * batch * args.steps / dt, "unit": "cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "scMomer.forward(mode='one') + classifier + CE + partial backward + clip + Adam",
                       "cells_per_gpu_per_step": batch, "gene_tokens": N_GENES, "depth": 6, "dim": 200, "heads": 10,
                       "nb_features": 266, "parallelism": f"dp{world}", "launch": "eager"},
            "final_loss": float(loss.item()), "roofline": None, "cpu_baseline": None}))


def _say(msg):
    if os.environ.get("SCM_BENCH_VERBOSE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` started directly (no torchrun): this process never touches the GPU runtime (it does
    not even count devices: a rank that finds no device of its own says so and exits non-zero), starts N copies
    of itself as ranks 0..N-1 (one per GPU, env:// rendezvous on 127.0.0.1, the layout torchrun would give them),
    waits for all of them (stopping the others as soon as one fails) and relays rank 0's JSON line.  Returns the exit
    code."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), RANK=str(r),
                   LOCAL_RANK=str(r))
        # RCCL shares device buffers between the ranks of a node through IPC handles; hosts whose driver only offers
        # dmabuf handles need the legacy mode off, or hipIpcGetMemHandle fails with "invalid argument".  An
        # environment that already sets the variable keeps its value.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # Rank 0's output is read by a thread while this one watches all ranks: a rank that fails (no device of its own,
    # a fault) must not leave the others waiting for it in a collective until somebody's time limit.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while True:
        codes = [q.poll() for q in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            time.sleep(2.0)                                  # let the failing rank's message out, then stop the rest
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            for q in procs:
                try:
                    q.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    q.kill()
            codes = [q.wait() for q in procs]
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out0 = chunks[0] if chunks else ""
    if any(codes):
        sys.stdout.write(out0 or "")
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        failed = [c for c in codes if c and c > 0]            # a rank's own failure, not the terminations it caused
        return failed[0] if failed else 1
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None or json.loads(line).get("n_gpus") != n:
        sys.stdout.write(out0 or "")
        print(f"bench.py: rank 0 did not report n_gpus == {n}", file=sys.stderr)
        return 3
    print(line)
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the kernels have no CPU path")
    # rehearsal switches (one-GPU boxes): SCM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, SCM_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device.  Never set by the driver.
    if os.environ.get("SCM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, but only {torch.cuda.device_count()} GPU(s) "
                         f"are visible (--gpus {args.gpus})")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend=os.environ.get("SCM_BENCH_BACKEND", "nccl"), init_method="env://",
                                world_size=world, rank=rank)

    _say("process group up")
    if args.mode in ("extract", "finetune", "s3"):
        {"extract": run_extract, "finetune": run_finetune, "s3": run_s3}[args.mode](args, world, rank, device)
        if world > 1:
            dist.destroy_process_group()
        return
    from scmomer_amd import dist as sdist
    from scmomer_amd import ops, s1, train
    torch.manual_seed(2024 + rank)
    model = build_model(device)
    model.train()
    # The shipped S1 driver runs the step: mask -> encoder forward -> CE / grad_acc -> backward -> accuracy, with
    # gradients folded into static accumulators by one multi-tensor add; every grad_acc-th micro-step it averages
    # the gradients over the ranks (one flat all-reduce), clips, applies Adam and refreshes the bf16 weight copies
    # (s1_pretrain_rna.py:287-312).  The micro-step is launch-bound when driven from Python (about 250 kernel
    # launches), so the driver records it once into a HIP graph and replays it.
    if args.grad_acc % args.window != 0:
        raise SystemExit(f"bench.py: --window {args.window} must divide --grad-acc {args.grad_acc}")
    G = args.window
    trainer = train.S1Trainer(model, learning_rate=1e-4, grad_acc=args.grad_acc, mask_prob=0.15, replace_prob=0.9,
                              use_graph=not args.eager, window=G)

    # each rank owns its own shard of synthetic cells (resident in HBM before timing starts)
    n_batches = 4
    # one step = one launch chain = G micro-batches of --batch cells stacked along dim 0
    data = [s1.synthetic_tokens(1000 * rank + i, G * args.batch, N_GENES, device) for i in range(n_batches)]
    last = {}

    def step(i):
        loss, acc = trainer.window_step(data[i % n_batches], G)
        last["loss"], last["acc"] = loss[-1], acc[-1]

    trainer.prepare(data[0], G)
    # set-up, like the graph recording above: one full accumulation window, so that the first optimizer step (which
    # allocates Adam's moments and loads its kernels: tens of milliseconds, once) is not part of the measurement
    # whatever --warmup is
    for i in range(args.grad_acc // G):
        step(i)
    graph = None if args.eager else True
    _say("graph captured" if graph is not None else "eager mode")
    for i in range(args.warmup):
        step(i)
        _say(f"warm-up step {i} issued")
    torch.cuda.synchronize()
    _say("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    _say("timed region done")
    dt = sdist.max_over_ranks(dt, device)
    final_loss = float(last["loss"].item()) * args.grad_acc
    final_acc = float(last["acc"].item())

    # Per-kernel durations: HIP events on the launch stream around every launch, taken on a few eager
    # micro-steps right after the timed region (event pairs cannot be read back from inside a replayed graph;
    # the kernels, their inputs and their launch configurations are the same).
    ops.trace_collect()
    ops.trace_enable("all")
    traced_steps = 3
    for i in range(traced_steps):
        trainer._forward_backward(data[i % n_batches], G)
    torch.cuda.synchronize()
    ops.trace_enable(None)
    trace = ops.trace_collect()

    if rank == 0:
        cells = world * G * args.batch * args.steps
        value = cells / dt
        # the dominant kernel = the traced kernel with the largest share of the step
        dominant = os.environ.get("SCM_BENCH_TRACE_KERNEL") or max(trace, key=lambda k: trace[k][0])
        ms, fl, cnt, by = trace[dominant]
        ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)          # FLOP per byte where the two roofs meet
        bound = "mfma" if by <= 0 or fl / by >= ridge else "hbm"
        tflops, gbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(dominant)
        mfma_util, valu_busy, util_src = measured_utilisation(dominant)
        step_bytes = sum(v[3] for v in trace.values()) / traced_steps
        # `bound` names the roof the kernel's arithmetic intensity puts it under; `limiter` says what the counter
        # passes show it waiting on (the attention passes sit under the HBM roof but are paced by vector-ALU issue)
        limiter = None
        if valu_busy is not None and mfma_util is not None:
            limiter = (f"vector-ALU issue: vector ALU busy {valu_busy:.0%} of the kernel's cycles, matrix cores "
                       f"{mfma_util:.0%} ({util_src})" if valu_busy >= 0.4 and valu_busy > mfma_util
                       else f"memory streaming: vector ALU busy {valu_busy:.0%}, matrix cores {mfma_util:.0%} ({util_src})")
        roof = {"bound": bound, "limiter": limiter, "kernel": dominant,
                "achieved": tflops if bound == "mfma" else gbs,
                "peak": PEAK_BF16_TFLOPS if bound == "mfma" else PEAK_HBM_GBS,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                "frac": tflops / PEAK_BF16_TFLOPS if bound == "mfma" else gbs / PEAK_HBM_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "hbm_achievable_gbs": HBM_ACHIEVABLE_GBS, "frac_of_achievable_hbm": gbs / HBM_ACHIEVABLE_GBS,
                "mfma_util": mfma_util, "valu_busy": valu_busy, "utilisation_source": util_src,
                "algorithmic_bytes_per_step_all_kernels": step_bytes,
                "algorithmic_bytes_per_step_lower_bound": 365e6 * args.batch * G,
                "avg_launch_ms": ms / cnt, "launches_per_step": cnt / traced_steps,
                "algorithmic_bytes_per_launch": by / cnt, "algorithmic_flops_per_launch": fl / cnt,
                "tflops": tflops, "gbs": gbs, "share_of_kernel_time": ms / sum(v[0] for v in trace.values()),
                "whole_step_tflops": value / world * FLOP_FWD_BWD_PER_CELL / 1e12,
                "whole_step_frac_of_mfma_peak": value / world * FLOP_FWD_BWD_PER_CELL / 1e12 / PEAK_BF16_TFLOPS}
        kernels = {k: {"ms_per_step": round(m_ / traced_steps, 4), "launches_per_step": c_ / traced_steps,
                       "tflops": round(f_ / (m_ * 1e-3) / 1e12, 1), "gbs": round(b_ / (m_ * 1e-3) / 1e9, 1)}
                   for k, (m_, f_, c_, b_) in sorted(trace.items(), key=lambda kv: -kv[1][0])}
        if args.trace_all:
            tot = sum(v[0] for v in trace.values())
            for k, (m_, f_, c_, b_) in sorted(trace.items(), key=lambda kv: -kv[1][0]):
                print(f"  {k:20s} {m_ / traced_steps:7.3f} ms/step {100 * m_ / tot:5.1f}%  x{c_ // traced_steps:3d}  "
                      f"{f_ / (m_ * 1e-3) / 1e12:7.1f} TFLOP/s {b_ / (m_ * 1e-3) / 1e9:7.0f} GB/s", file=sys.stderr)
        base = None
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline()
        out = {"metric": "cells/sec, Performer RNA encoder fwd+bwd (16.9k genes)", "value": value, "unit": "cells/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic",
               "config": {"workload": "S1 RNA masked-autoencoding pretrain (mask_prob 0.15, batch 3): mask + Performer "
                                      "encoder fwd + CE + bwd + accuracy per micro-step; gradient average, clip + Adam "
                                      f"every grad_acc micro-steps (scmomer_amd.train.S1Trainer); {G} micro-step(s) of "
                                      f"{args.batch} cells per launch chain (one step), key stabiliser, loss mean and "
                                      f"mask draws per {args.batch}-cell micro-batch as in the reference's loop",
                          "cells_per_gpu_per_step": G * args.batch, "micro_batch_cells": args.batch,
                          "micro_steps_per_launch_chain": G, "gene_tokens": N_GENES, "depth": 6, "dim": 200,
                          "heads": 10, "nb_features": 266, "grad_acc": args.grad_acc,
                          "parallelism": f"dp{world}", "feature_redraw": "disabled",
                          "launch": "eager" if args.eager else "hip-graph replay of the launch chain"},
               "final_loss": final_loss, "final_batch_accuracy": final_acc, "roofline": roof, "cpu_baseline": base, "kernels": kernels}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
