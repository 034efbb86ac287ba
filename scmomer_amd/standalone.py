"""Stand-alone forward of the layer sub-modules (PreLayerNorm, SelfAttention, FeedForward), each
as one autograd node over the same kernels the fused stack uses (stack.attn_fwd/ffn_fwd ...).

These exist so code that calls a sub-module directly (as the reference's SequentialSequence does,
reversible.py:144-145) keeps working; PerformerLM.forward itself takes the fused path.
"""
from __future__ import annotations

import torch

from . import stack

F32 = torch.float32


class _BranchFn(torch.autograd.Function):
    """y = branch(x) without the residual add.  kind: 'attn' | 'ffn'; ln: apply the pre-LayerNorm."""

    @staticmethod
    def forward(ctx, kind, ln, p, w, keys, x, *params):
        if not x.is_cuda:
            raise RuntimeError("scmomer_amd runs on the MI355X kernels only: pass device tensors")
        shape = x.shape
        x3 = x.to(F32).reshape(-1, shape[-2], shape[-1]).contiguous() if x.dim() != 3 else x.to(F32).contiguous()
        fn = stack.attn_fwd if kind == "attn" else stack.ffn_fwd
        y, tape, _ = fn(x3, p, w, ln=ln, resid=None, save=True)
        ctx.kind, ctx.ln, ctx.p, ctx.w, ctx.keys, ctx.tape, ctx.shape = kind, ln, p, w, keys, tape, shape
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        needs = ctx.needs_input_grad
        need = {k: False for k in (stack.ATTN_KEYS if ctx.kind == "attn" else stack.FFN_KEYS)}
        for i, k in enumerate(ctx.keys):
            need[k] = needs[6 + i]
        d = dy.shape[-1]
        dy2 = dy.to(F32).reshape(-1, d).contiguous()
        fn = stack.attn_bwd if ctx.kind == "attn" else stack.ffn_bwd
        dx, g = fn(dy2, ctx.tape, ctx.p, ctx.w, need, ln=ctx.ln, dres=None, need_dx=needs[5])
        out = [g.get(k) if needs[6 + i] else None for i, k in enumerate(ctx.keys)]
        ctx.tape = None
        return (None, None, None, None, None, dx.view(ctx.shape) if (dx is not None and needs[5]) else None, *out)


def _run(kind, owner, p, ln, x):
    cache = stack._cache_of(owner)
    w = cache.attn(0, p) if kind == "attn" else cache.ffn(0, p)
    all_keys = stack.ATTN_KEYS if kind == "attn" else stack.FFN_KEYS
    skip = () if ln else (("g1", "b1") if kind == "attn" else ("g2", "b2"))
    keys = [k for k in all_keys if p.get(k) is not None and k not in skip]
    return _BranchFn.apply(kind, ln, p, w, keys, x, *[p[k] for k in keys])


def pre_layer_norm_forward(mod, x, **kwargs):
    """PreLayerNorm.forward (performer_pytorch.py:294): fn(norm(x))."""
    from .performer import Chunk, FeedForward, SelfAttention
    fn = mod.fn
    if isinstance(fn, SelfAttention):
        return _run("attn", mod, stack.attn_params(mod), True, x)
    if isinstance(fn, Chunk) or isinstance(fn, FeedForward):
        return _run("ffn", mod, stack.ffn_params(mod), True, x)
    raise NotImplementedError("PreLayerNorm wraps SelfAttention or FeedForward on this path")


class _Holder:
    """adapts a bare SelfAttention / FeedForward to the (norm, fn) accessor used for parameters"""

    def __init__(self, fn):
        self.fn = fn
        self.norm = type("N", (), {"weight": None, "bias": None})()


def self_attention_forward(mod, x):
    return _run("attn", mod, stack.attn_params(_Holder(mod)), False, x)


def feed_forward_forward(mod, x):
    return _run("ffn", mod, stack.ffn_params(_Holder(mod)), False, x)
