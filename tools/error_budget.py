#!/usr/bin/env python3
"""Error budget of the bf16 pipeline: which rounding contributes what to the error of the encoder's outputs.

A torch emulation of the encoder forward (the oracle's math, oracle/performer_oracle.py, run in f32 on the GPU for
speed) with bf16 rounding switched on at ONE class of points at a time -- the points where the kernels round:

  w      weight matrices of the six products (bf16 copies of the f32 masters)
  h      LayerNorm outputs (operand of the q/k/v and FeedForward-1 products)
  qkv    q, k, v as stored (bf16 outputs of the fused projection)
  omega  the projection matrix (bf16 image in LDS)
  E      the exponentials exp(S - diag - stab) of queries and keys (operands of the second matrix products)
  ctx    the centred context C' = C - z obar^T (bf16 image)
  o      the attention output (bf16, operand of the output projection)
  act    the FeedForward activation gelu(.) (bf16, operand of FeedForward-2)

and with all of them on.  Reported: relative L2 and max/max error of the logits (Linear-7 head), of the per-gene
scores (1 x 200 conv + ReLU head) and of the 128-wide cell embedding (projection, L2 normalisation, LayerNorm)
against the same emulation with every switch off.  Test infrastructure (it imports the oracle); not product code.

usage: python tools/error_budget.py [--genes 16906] [--cells 1] [--depth 6] [--out FILE]
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import performer_oracle as O  # noqa: E402

POINTS = ("w", "h", "qkv", "omega", "E", "ctx", "o", "act")


def rb(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def attention(q, k, v, omega, sw):
    """FAVOR+ as the kernels evaluate it: pure exponentials as matrix operands, eps terms analytic, centred context"""
    d, m = q.shape[-1], omega.shape[0]
    c, eps = d ** -0.25, O.EPS_KERNEL
    om = rb(omega, "omega" in sw)

    def expo(x, is_query):
        s = torch.matmul(c * x, om.t())
        diag = (x * x).sum(-1, keepdim=True) / 2.0 * c * c
        stab = s.amax(-1, keepdim=True) if is_query else s.max()
        return rb(torch.exp(s - diag - stab), "E" in sw)

    eq, ek = expo(q, True), expo(k, False)
    z = ek.sum(-2) + eps * k.shape[-2]                                   # [B,H,M]
    ctx = torch.matmul(ek.transpose(-1, -2), v) + eps * v.sum(-2, keepdim=True)   # [B,H,M,d]
    obar = ctx.sum(-2, keepdim=True) / z.sum(-1, keepdim=True).unsqueeze(-1)       # [B,H,1,d]
    cc = rb(ctx - z.unsqueeze(-1) * obar, "ctx" in sw)
    den = torch.matmul(eq, z.unsqueeze(-1)) + eps * z.sum(-1, keepdim=True).unsqueeze(-1)
    # (E + eps) C' : the eps row sum of the centred context is zero
    return obar + torch.matmul(eq, cc) / den


def forward(tokens, p, depth, heads, sw):
    x = O.embed(tokens, p)
    b, n, _ = x.shape
    W = lambda k_: rb(p[k_], "w" in sw)
    for i in range(depth):
        pre = f"performer.net.layers.{i}.0."
        h = rb(O.layer_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"]), "h" in sw)
        split = lambda t: rb(t, "qkv" in sw).reshape(b, n, heads, -1).permute(0, 2, 1, 3)
        q, k, v = (split(h @ W(pre + f"fn.to_{c_}.weight").t()) for c_ in "qkv")
        o = attention(q, k, v, p[pre + "fn.fast_attention.projection_matrix"], sw)
        o = rb(o.permute(0, 2, 1, 3).reshape(b, n, -1), "o" in sw)
        x = x + o @ W(pre + "fn.to_out.weight").t() + p[pre + "fn.to_out.bias"]
        pre = f"performer.net.layers.{i}.1."
        h = rb(O.layer_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"]), "h" in sw)
        a = rb(O.gelu_erf(h @ W(pre + "fn.fn.w1.weight").t() + p[pre + "fn.fn.w1.bias"]), "act" in sw)
        x = x + a @ W(pre + "fn.fn.w2.weight").t() + p[pre + "fn.fn.w2.bias"]
    return O.layer_norm(x, p["norm.weight"], p["norm.bias"])


def outputs(enc, p, tail):
    logits = enc @ p["to_out.weight"].t() + p["to_out.bias"]
    genes = torch.relu(enc @ tail["conv_w"] + tail["conv_b"])            # rna_encoder.py:11-16
    f = genes @ tail["proj"].t()                                          # scmomer.py:126-128
    f = f / f.norm(dim=-1, keepdim=True)
    emb = O.layer_norm(f, tail["ln_w"], tail["ln_b"])
    return {"logits": logits, "gene scores": genes, "embedding": emb}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=16906)
    ap.add_argument("--cells", type=int, default=1)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    p = {k: v.to(dev) for k, v in O.make_params(2024, depth=args.depth, max_seq_len=args.genes).items()}
    gen = torch.Generator().manual_seed(5)
    tail = {"conv_w": (torch.randn(200, generator=gen) * 0.05).to(dev), "conv_b": torch.tensor(0.1, device=dev),
            "proj": (torch.randn((128, args.genes), generator=gen) / math.sqrt(args.genes)).to(dev),
            "ln_w": torch.ones(128, device=dev), "ln_b": torch.zeros(128, device=dev)}
    tokens = O.standin_cells(9, args.cells, args.genes).to(dev)
    lines = []
    with torch.no_grad():
        ref = outputs(forward(tokens, p, args.depth, 10, ()), p, tail)
        rows = [(pt, (pt,)) for pt in POINTS] + [("all", POINTS), ("all but E", tuple(x for x in POINTS if x != "E")),
                                                ("all but h", tuple(x for x in POINTS if x != "h")),
                                                ("all but qkv", tuple(x for x in POINTS if x != "qkv")),
                                                ("all but w", tuple(x for x in POINTS if x != "w"))]
        lines.append(f"error budget: {args.cells} cell(s) x {args.genes} genes, depth {args.depth}, f32 emulation on {dev}; "
                     "bf16 rounding at one class of points at a time")
        lines.append(f"{'rounded':12s} " + " ".join(f"{k + ' relL2':>20s} {k + ' max/max':>22s}" for k in ref))
        for name, sw in rows:
            got = outputs(forward(tokens, p, args.depth, 10, sw), p, tail)
            cells = []
            for k in ref:
                dlt = (got[k] - ref[k]).double()
                cells.append(f"{float(dlt.norm() / ref[k].double().norm()):20.3e} "
                             f"{float(dlt.abs().max() / ref[k].double().abs().max()):22.3e}")
            lines.append(f"{name:12s} " + " ".join(cells))
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
