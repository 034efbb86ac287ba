"""Host-side mirror of the reference's Performer module API
(model/performer/performer_pytorch.py and model/performer/reversible.py), running on the
gfx950 kernels of libscmomer_hip.so.

Same class names, constructor arguments, attribute paths and ``state_dict`` layout as the
reference, so S1 pretraining, embedding extraction and the fine-tune scripts that slice
``performer.net.layers`` work unchanged.  ``PerformerLM.forward`` runs the whole stack
(embedding -> 6 x {LN, QKV, FAVOR+, out-proj, LN, FFN} -> final LN -> head) as ONE autograd
node that sequences C-ABI calls; the sub-modules (``FastAttention`` ...) are usable on their own.

Options the scMomer callers never enable (causal, local heads, reversible, GLU, scalenorm,
rezero, generalized attention, no_projection, cross attention, output_attentions) raise
NotImplementedError.
"""
from __future__ import annotations

import math
import os
import warnings
from functools import partial
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else d


def cast_tuple(val):
    return (val,) if not isinstance(val, tuple) else val


def find_modules(nn_module, type_):
    return [m for m in nn_module.modules() if isinstance(m, type_)]


def get_module_device(module):
    try:
        return next(module.parameters()).device
    except StopIteration:
        return next(module.buffers()).device


class Always(nn.Module):
    def __init__(self, val):
        super().__init__()
        self.val = val

    def forward(self, *args, **kwargs):
        return self.val


def _unsupported(name, value, allowed):
    if value != allowed:
        raise NotImplementedError(f"{name}={value!r} is not part of the scMomer Performer path "
                                  f"(only {allowed!r} is implemented on the MI355X kernels)")


# --------------------------------------------------------------------------------------------
# projection matrix construction: performer_pytorch.py:110-139
# --------------------------------------------------------------------------------------------
def orthogonal_matrix_chunk(cols, device=None):
    unstructured_block = torch.randn((cols, cols), device=device)
    q, _ = torch.linalg.qr(unstructured_block.cpu(), mode="reduced")
    return q.to(device).t()


def _on_gpu(device) -> bool:
    return device is not None and torch.device(device).type == "cuda"


def _projection_draws(nb_rows, nb_columns, scaling, device):
    """The Gaussian draws of one projection matrix, in the reference's order (performer_pytorch.py:119-133): one
    [cols, cols] block per 64 rows (the last one partly used), then the [rows, cols] matrix of the row norms."""
    nblk = (nb_rows + nb_columns - 1) // nb_columns
    blocks = torch.stack([torch.randn((nb_columns, nb_columns), device=device) for _ in range(nblk)])
    mult = torch.randn((nb_rows, nb_columns), device=device) if scaling == 0 else None
    return blocks, mult


def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, scaling=0, device=None):
    if scaling not in (0, 1):
        raise ValueError(f"Invalid scaling {scaling}")
    if _on_gpu(device) and nb_columns == ops.DIM_HEAD:
        # draws and the QR both happen on the device: no host round trip, no stall of the training stream
        blocks, mult = _projection_draws(nb_rows, nb_columns, scaling, device)
        return ops.orthogonal_omega(blocks[None], None if mult is None else mult[None], nb_rows, scaling)[0]
    nb_full_blocks = int(nb_rows / nb_columns)
    block_list = [orthogonal_matrix_chunk(nb_columns, device=device) for _ in range(nb_full_blocks)]
    remaining_rows = nb_rows - nb_full_blocks * nb_columns
    if remaining_rows > 0:
        block_list.append(orthogonal_matrix_chunk(nb_columns, device=device)[:remaining_rows])
    final_matrix = torch.cat(block_list)
    if scaling == 0:
        multiplier = torch.randn((nb_rows, nb_columns), device=device).norm(dim=1)
    else:
        multiplier = math.sqrt(float(nb_columns)) * torch.ones((nb_rows,), device=device)
    return torch.diag(multiplier) @ final_matrix


@torch.no_grad()
def redraw_projection_matrices(fast_attentions, device) -> None:
    """New projection matrices for a list of FastAttention modules with ONE kernel launch (the reference redraws
    them one by one with a host QR each, performer_pytorch.py:527-531); the draws keep the per-module order."""
    fast_attentions = list(fast_attentions)
    if not fast_attentions:
        return
    same = len({(fa.nb_features, fa.dim_heads, fa.ortho_scaling) for fa in fast_attentions}) == 1
    if not (_on_gpu(device) and same and fast_attentions[0].dim_heads == ops.DIM_HEAD):
        for fa in fast_attentions:
            fa.redraw_projection_matrix(device)
        return
    fa0 = fast_attentions[0]
    draws = [_projection_draws(fa0.nb_features, fa0.dim_heads, fa0.ortho_scaling, device) for _ in fast_attentions]
    blocks = torch.stack([d[0] for d in draws])
    mult = torch.stack([d[1] for d in draws]) if fa0.ortho_scaling == 0 else None
    omega = ops.orthogonal_omega(blocks, mult, fa0.nb_features, fa0.ortho_scaling)
    for fa, om in zip(fast_attentions, omega):
        fa.projection_matrix.copy_(om)


# --------------------------------------------------------------------------------------------
# the reference's free functions (performer_pytorch.py:67-92,144-149): materialised feature maps in f32
# --------------------------------------------------------------------------------------------
def _no_grad_path(name, *tensors):
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        raise NotImplementedError(
            f"{name} materialises the feature maps and is forward-only on the MI355X kernels; the differentiable "
            "attention is FastAttention (fused forward and backward, feature maps never written to memory)")


def softmax_kernel(data, *, projection_matrix, is_query, normalize_data=True, eps=1e-4, device=None):
    """performer_pytorch.py:67-92 on the device, exact f32 arithmetic: data [b, h, n, 64], projection_matrix
    [m, 64] -> [b, h, n, m] in data's dtype.  Queries are stabilised by their own row max, keys by ONE max over
    the whole tensor (:89-90).  `device` is accepted for signature compatibility (the result lives on data's
    device, as in the reference)."""
    _no_grad_path("softmax_kernel", data, projection_matrix)
    if not data.is_cuda:
        raise RuntimeError("softmax_kernel runs on the MI355X kernels only: pass device tensors")
    out = ops.feature_map(data.detach().to(F32).contiguous(),
                          projection_matrix.detach().to(device=data.device, dtype=F32).contiguous(),
                          bool(is_query), bool(normalize_data), float(eps))
    return out.to(data.dtype)


def linear_attention(q, k, v):
    """performer_pytorch.py:144-149 on the device, exact f32 arithmetic: feature-mapped q, k [b, h, n, m] and
    v [b, h, n, 64] -> [b, h, n, 64]."""
    _no_grad_path("linear_attention", q, k, v)
    if not q.is_cuda:
        raise RuntimeError("linear_attention runs on the MI355X kernels only: pass device tensors")
    out = ops.linear_attention(q.detach().to(F32).contiguous(), k.detach().to(F32).contiguous(),
                               v.detach().to(F32).contiguous())
    return out.to(v.dtype)


# --------------------------------------------------------------------------------------------
# FAVOR+ attention as a standalone autograd op: FastAttention.forward, performer_pytorch.py:233-250
# --------------------------------------------------------------------------------------------
def _as_bf16_bhnd(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != BF16:
        t = t.to(BF16)
    if t.stride(-1) != 1 or any(s % 8 for s in t.stride()[:-1]):
        t = t.contiguous()
    return t


class _FavorAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, omega):
        if not q.is_cuda:
            raise RuntimeError("FastAttention runs on the MI355X kernels only: pass device tensors")
        in_dtype = q.dtype
        qb, kb, vb = (_as_bf16_bhnd(t) for t in (q, k, v))
        omega_b = ops.prep_omega(omega.to(F32).contiguous())
        out = torch.empty(qb.shape, dtype=BF16, device=q.device)
        outc = torch.empty(qb.shape, dtype=BF16, device=q.device)
        saved = ops.favor_fwd(qb, kb, vb, omega_b, out, outc)
        ctx.saved = (qb, kb, vb, outc, omega_b, saved)
        ctx.in_dtype = in_dtype
        return out.to(in_dtype)

    @staticmethod
    def backward(ctx, dout):
        qb, kb, vb, out, omega_b, saved = ctx.saved
        dob = _as_bf16_bhnd(dout)
        dq, dk, dv = (torch.empty(qb.shape, dtype=BF16, device=qb.device) for _ in range(3))
        ops.favor_bwd(qb, kb, vb, out, dob, omega_b, saved, dq, dk, dv)
        dt = ctx.in_dtype
        return dq.to(dt), dk.to(dt), dv.to(dt), None


class FastAttention(nn.Module):
    """performer_pytorch.py:198-264 (softmax-kernel, non-causal branch)."""

    def __init__(self, dim_heads, nb_features=None, ortho_scaling=0, causal=False, generalized_attention=False,
                 kernel_fn=nn.ReLU(), no_projection=False):
        super().__init__()
        nb_features = default(nb_features, int(dim_heads * math.log(dim_heads)))
        _unsupported("causal", causal, False)
        _unsupported("generalized_attention", generalized_attention, False)
        _unsupported("no_projection", no_projection, False)
        if dim_heads != ops.DIM_HEAD or nb_features != ops.NB_FEATURES:
            raise NotImplementedError("the MI355X kernels are built for dim_heads=64, nb_features=266 "
                                      f"(got {dim_heads}, {nb_features})")
        self.dim_heads = dim_heads
        self.nb_features = nb_features
        self.ortho_scaling = ortho_scaling
        self.create_projection = partial(gaussian_orthogonal_random_matrix, nb_rows=self.nb_features,
                                         nb_columns=dim_heads, scaling=ortho_scaling)
        self.register_buffer("projection_matrix", self.create_projection())
        self.generalized_attention = generalized_attention
        self.kernel_fn = kernel_fn
        self.no_projection = no_projection
        self.causal = causal

    @torch.no_grad()
    def redraw_projection_matrix(self, device):
        projections = self.create_projection(device=device)
        self.projection_matrix.copy_(projections)
        del projections

    def forward(self, q, k, v, output_attentions=False):
        _unsupported("output_attentions", output_attentions, False)
        return _FavorAttentionFn.apply(q, k, v, self.projection_matrix)


# --------------------------------------------------------------------------------------------
# layer pieces (module tree = checkpoint layout)
# --------------------------------------------------------------------------------------------
class PreLayerNorm(nn.Module):
    """performer_pytorch.py:289-295"""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        from .standalone import pre_layer_norm_forward
        return pre_layer_norm_forward(self, x, **kwargs)


class Chunk(nn.Module):
    """performer_pytorch.py:297-308"""

    def __init__(self, chunks, fn, along_dim=-1):
        super().__init__()
        _unsupported("ff_chunks", chunks, 1)
        self.dim = along_dim
        self.chunks = chunks
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, **kwargs)


class FeedForward(nn.Module):
    """performer_pytorch.py:310-331 (glu=False, exact-erf GELU, dropout 0)"""

    def __init__(self, dim, mult=4, dropout=0., activation=None, glu=False):
        super().__init__()
        _unsupported("ff_glu", glu, False)
        _unsupported("ff_dropout", dropout, 0.)
        if activation is not None and activation is not nn.GELU:
            raise NotImplementedError("only nn.GELU is implemented")
        self.glu = glu
        self.w1 = nn.Linear(dim, dim * mult)
        self.act = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.w2 = nn.Linear(dim * mult, dim)

    def forward(self, x, **kwargs):
        from .standalone import feed_forward_forward
        return feed_forward_forward(self, x)


class SelfAttention(nn.Module):
    """performer_pytorch.py:333-406 (global FAVOR+ heads only)"""

    def __init__(self, dim, causal=False, heads=8, dim_head=64, local_heads=0, local_window_size=256,
                 nb_features=None, feature_redraw_interval=1000, generalized_attention=False, kernel_fn=nn.ReLU(),
                 dropout=0., no_projection=False, qkv_bias=False):
        super().__init__()
        assert dim % heads == 0, "dimension must be divisible by number of heads"
        _unsupported("local_heads", local_heads, 0)
        _unsupported("attn_dropout", dropout, 0.)
        dim_head = default(dim_head, dim // heads)
        inner_dim = dim_head * heads
        self.fast_attention = FastAttention(dim_head, nb_features, causal=causal,
                                            generalized_attention=generalized_attention, kernel_fn=kernel_fn,
                                            no_projection=no_projection)
        self.heads = heads
        self.global_heads = heads - local_heads
        self.local_attn = None
        self.to_q = nn.Linear(dim, inner_dim, bias=qkv_bias)
        self.to_k = nn.Linear(dim, inner_dim, bias=qkv_bias)
        self.to_v = nn.Linear(dim, inner_dim, bias=qkv_bias)
        self.to_out = nn.Linear(inner_dim, dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, pos_emb=None, context=None, mask=None, context_mask=None, output_attentions=False,
                **kwargs):
        for name, val in (("pos_emb", pos_emb), ("context", context), ("mask", mask), ("context_mask", context_mask)):
            if exists(val):
                raise NotImplementedError(f"{name} is not used on the scMomer path")
        _unsupported("output_attentions", output_attentions, False)
        from .standalone import self_attention_forward
        return self_attention_forward(self, x)


class SequentialSequence(nn.Module):
    """reversible.py:126-151: x = x + f(x); x = x + g(x) per layer."""

    def __init__(self, layers, args_route={}):
        super().__init__()
        assert all(len(route) == len(layers) for route in args_route.values()), \
            "each argument route map must have the same depth as the number of sequential layers"
        self.layers = layers
        self.args_route = args_route

    def forward(self, x, output_attentions=False, **kwargs):
        _unsupported("output_attentions", output_attentions, False)
        for name, val in kwargs.items():          # the routed keywords (mask, pos_emb) are None on this path
            if exists(val):
                raise NotImplementedError(f"argument {name} is not used on the scMomer path")
        for f, g in self.layers:
            x = x + f(x)
            x = x + g(x)
        return x


class Performer(nn.Module):
    """performer_pytorch.py:450-539"""

    def __init__(self, dim, depth, heads, dim_head, local_attn_heads=0, local_window_size=256, causal=False,
                 ff_mult=4, nb_features=None, feature_redraw_interval=1000, reversible=False, ff_chunks=1,
                 generalized_attention=False, kernel_fn=nn.ReLU(), use_scalenorm=False, use_rezero=False,
                 ff_glu=False, ff_dropout=0., attn_dropout=0., cross_attend=False, no_projection=False,
                 auto_check_redraw=True, qkv_bias=True):
        super().__init__()
        layers = nn.ModuleList([])
        local_attn_heads = cast_tuple(local_attn_heads)
        local_attn_heads = local_attn_heads * depth if len(local_attn_heads) == 1 else local_attn_heads
        assert len(local_attn_heads) == depth, \
            "tuple specifying number of local attention heads per depth must be equal to the total depth"
        _unsupported("reversible", reversible, False)
        _unsupported("use_scalenorm", use_scalenorm, False)
        _unsupported("use_rezero", use_rezero, False)
        _unsupported("cross_attend", cross_attend, False)
        for _, local_heads in zip(range(depth), local_attn_heads):
            layers.append(nn.ModuleList([
                PreLayerNorm(dim, SelfAttention(dim, causal=causal, heads=heads, dim_head=dim_head,
                                                local_heads=local_heads, local_window_size=local_window_size,
                                                nb_features=nb_features, generalized_attention=generalized_attention,
                                                kernel_fn=kernel_fn, dropout=attn_dropout,
                                                no_projection=no_projection, qkv_bias=qkv_bias)),
                PreLayerNorm(dim, Chunk(ff_chunks, FeedForward(dim, mult=ff_mult, dropout=ff_dropout, glu=ff_glu),
                                        along_dim=1)),
            ]))
        route_attn = ((True, False),) * depth
        attn_route_map = {"mask": route_attn, "pos_emb": route_attn}
        self.net = SequentialSequence(layers, args_route={**attn_route_map})
        self.auto_check_redraw = auto_check_redraw
        self.feature_redraw_interval = feature_redraw_interval
        self.register_buffer("calls_since_last_redraw", torch.tensor(0))
        self._calls_host = None
        # called with this module right after a redraw.  Data-parallel drivers put a broadcast of rank 0's matrices
        # here: under the reference's DistributedDataParallel the buffer broadcast at every forward does that
        # (s1_pretrain_rna.py:242), so every replica keeps rank 0's draw.
        self.redraw_sync = None
        # None: the key features of a forward are stabilised by ONE max over the whole batch, as in the reference
        # (performer_pytorch.py:89-90).  An integer g: the batch is g-cell micro-batches stacked along dim 0 (a driver
        # running several of the reference's forward calls as one launch chain); every run of g cells keeps its own
        # max, so each micro-batch gets exactly the result of its own call.
        self.stabiliser_group_cells = None

    def fix_projection_matrices_(self):
        self.feature_redraw_interval = None

    def _calls(self) -> int:
        """Host mirror of the calls_since_last_redraw buffer: the reference compares the device buffer on every
        forward (a synchronisation per call); here the buffer is read once and then tracked on the host."""
        if self._calls_host is None:
            self._calls_host = int(self.calls_since_last_redraw.item())
        return self._calls_host

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._calls_host = None

    def check_redraw_projections(self):
        """performer_pytorch.py:520-534: in train() mode, redraw every FastAttention projection after
        feature_redraw_interval forward calls, else tick the counter.  All matrices are redrawn by one device
        kernel (draws and QR on the GPU), so the training stream does not wait for the host."""
        if not self.training:
            return
        if exists(self.feature_redraw_interval) and self._calls() >= self.feature_redraw_interval:
            device = get_module_device(self)
            redraw_projection_matrices(find_modules(self, FastAttention), device)
            self.calls_since_last_redraw.zero_()
            self._calls_host = 0
            if self.redraw_sync is not None:
                self.redraw_sync(self)
            return
        self.calls_since_last_redraw += 1
        if self._calls_host is not None:
            self._calls_host += 1

    def forward(self, x, output_attentions=False, **kwargs):
        _unsupported("output_attentions", output_attentions, False)
        if self.auto_check_redraw:
            self.check_redraw_projections()
        from .stack import run_layers
        return run_layers(self, x)


class Gene2VecPositionalEmbedding(nn.Module):
    """performer_pytorch.py:436-446: frozen float64 table [max_seq_len + 1, dim] (gene2vec rows plus
    one zero row); forward returns the rows 0..n-1.

    The reference reads a hard-coded ``.npy`` path.  Here the file comes from ``SCMOMER_GENE2VEC``
    (or ``path=``); without it the table starts as zeros and is expected to be filled by
    ``load_state_dict`` (the table is part of every checkpoint)."""

    def __init__(self, dim, max_seq_len, path: Optional[str] = None):
        super().__init__()
        path = path or os.environ.get("SCMOMER_GENE2VEC")
        if path and os.path.exists(path):
            gene2vec_weight = np.load(path)
        else:
            gene2vec_weight = np.zeros((max_seq_len, dim))
        gene2vec_weight = np.concatenate((gene2vec_weight, np.zeros((1, gene2vec_weight.shape[1]))), axis=0)
        self.emb = nn.Embedding.from_pretrained(torch.from_numpy(gene2vec_weight))

    def forward(self, x):
        t = torch.arange(x.shape[1], device=x.device)
        return self.emb(t)


class PerformerLM(nn.Module):
    """performer_pytorch.py:541-636"""

    def __init__(self, *, num_tokens, max_seq_len, dim, depth, heads, dim_head=64, local_attn_heads=0,
                 local_window_size=256, causal=False, ff_mult=4, nb_features=None, feature_redraw_interval=1000,
                 reversible=False, ff_chunks=1, ff_glu=False, emb_dropout=0., ff_dropout=0., attn_dropout=0.,
                 generalized_attention=False, kernel_fn=nn.ReLU(), use_scalenorm=False, use_rezero=False,
                 cross_attend=False, no_projection=False, tie_embed=False, g2v_position_emb=True,
                 auto_check_redraw=True, qkv_bias=False):
        super().__init__()
        local_attn_heads = cast_tuple(local_attn_heads)
        _unsupported("emb_dropout", emb_dropout, 0.)
        _unsupported("tie_embed", tie_embed, False)
        if dim > 200 or dim % 8 != 0:
            raise NotImplementedError("the MI355X kernels need dim <= 200 and dim % 8 == 0")
        self.max_seq_len = max_seq_len
        self.token_emb = nn.Embedding(num_tokens, dim)
        if g2v_position_emb:
            self.pos_emb = Gene2VecPositionalEmbedding(dim, max_seq_len)
        else:
            self.pos_emb = torch.zeros_like
        self.layer_pos_emb = Always(None)
        self.dropout = nn.Dropout(emb_dropout)
        self.performer = Performer(dim, depth, heads, dim_head, local_attn_heads, local_window_size, causal, ff_mult,
                                   nb_features, feature_redraw_interval, reversible, ff_chunks, generalized_attention,
                                   kernel_fn, use_scalenorm, use_rezero, ff_glu, ff_dropout, attn_dropout,
                                   cross_attend, no_projection, auto_check_redraw, qkv_bias)
        self.norm = nn.LayerNorm(dim)
        self.to_out = nn.Linear(dim, num_tokens) if not tie_embed else None

    def check_redraw_projections(self):
        self.performer.check_redraw_projections()

    def fix_projection_matrices_(self):
        self.performer.fix_projection_matrices_()

    def raise_on_bad_token(self):
        """The reference's nn.Embedding faults on a token id outside [0, num_tokens) (performer_pytorch.py:604).
        The embedding kernel cannot raise from the device: it looks such a token up as token 0 and sets a flag.
        This call reads the flag (it synchronises with the device: use it where the host waits anyway, e.g. at the
        end of an epoch), clears it and raises IndexError when any forward since the last check saw a bad id."""
        if self.pop_bad_token_flag():
            raise IndexError(self.bad_token_message())

    def bad_token_message(self) -> str:
        return (f"token id outside [0, {self.token_emb.num_embeddings}) reached the embedding "
                "(negative or NaN expression values, or a wrong class_limit)")

    def pop_bad_token_flag(self) -> bool:
        """Read and clear the bad-token flag without raising (synchronises with the device): data-parallel drivers
        combine it over the ranks first, so that every rank raises together instead of one rank leaving its peers
        blocked in the next collective."""
        bad = False
        for flag in self.__dict__.get("_scm_token_err", {}).values():
            if int(flag.item()):
                flag.zero_()
                bad = True
        return bad

    def forward(self, x, return_encodings=False, output_attentions=False, **kwargs):
        b, n = x.shape
        assert n <= self.max_seq_len, \
            f"sequence length {n} must be less than the max sequence length {self.max_seq_len}"
        _unsupported("output_attentions", output_attentions, False)
        for k_, v_ in kwargs.items():
            if exists(v_):
                raise NotImplementedError(f"argument {k_} is not used on the scMomer path")
        if self.performer.auto_check_redraw:
            self.performer.check_redraw_projections()
        from .stack import run_model
        return run_model(self, x, return_encodings)
