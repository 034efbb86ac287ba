"""Sequencing of the C-ABI kernels for the Performer stack, as autograd nodes.

One ``torch.autograd.Function`` covers embedding -> layers -> final norm/head
(PerformerLM.forward, performer_pytorch.py:599-636; SequentialSequence.forward,
reversible.py:133-151).  Forward saves, per layer, the f32 residual inputs, the bf16 GEMM
operands (LN outputs, fused qkv, attention output, FFN pre-activation and activation) and the
small attention state; backward replays the layers in reverse with the data-gradient,
weight-gradient and LayerNorm-backward kernels and honours ``needs_input_grad`` (frozen
encoders stop early: s1_pretrain_rna.py:223-228, s2_pretrain_multimodal.py:234-241).
"""
from __future__ import annotations

import threading
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops

BF16 = torch.bfloat16
F32 = torch.float32


# --------------------------------------------------------------------------------------------
# per-layer parameter handling
# --------------------------------------------------------------------------------------------
ATTN_KEYS = ("g1", "b1", "wq", "wk", "wv", "bq", "bk", "bv", "wo", "bo")
FFN_KEYS = ("g2", "b2", "w1", "c1", "w2", "c2")


def attn_params(pre_ln) -> Dict[str, Optional[torch.Tensor]]:
    """PreLayerNorm(SelfAttention) -> tensors (biases of q/k/v may be None)."""
    sa = pre_ln.fn
    return {"g1": pre_ln.norm.weight, "b1": pre_ln.norm.bias, "wq": sa.to_q.weight, "wk": sa.to_k.weight,
            "wv": sa.to_v.weight, "bq": sa.to_q.bias, "bk": sa.to_k.bias, "bv": sa.to_v.bias,
            "wo": sa.to_out.weight, "bo": sa.to_out.bias, "omega": sa.fast_attention.projection_matrix,
            "heads": sa.heads}


def ffn_params(pre_ln) -> Dict[str, Optional[torch.Tensor]]:
    """PreLayerNorm(Chunk(FeedForward)) -> tensors."""
    ff = pre_ln.fn.fn if hasattr(pre_ln.fn, "fn") else pre_ln.fn
    return {"g2": pre_ln.norm.weight, "b2": pre_ln.norm.bias, "w1": ff.w1.weight, "c1": ff.w1.bias,
            "w2": ff.w2.weight, "c2": ff.w2.bias}


def _ver(*ts) -> tuple:
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)


class WeightCache:
    """bf16 (and transposed bf16) copies of the f32 master weights, refreshed when a parameter's storage or
    version counter changes (optimizer steps and load_state_dict bump the version).  A refresh rewrites the
    SAME device buffers, so a HIP graph captured over forward/backward keeps reading valid copies: while a
    stream is capturing, existing copies are used as they are (no cast is recorded into the graph), and the
    owner refreshes them eagerly after each weight update (refresh_weights)."""

    def __init__(self):
        self.store: Dict[Any, Tuple[tuple, dict]] = {}

    def _lookup(self, key, ver):
        hit = self.store.get(key)
        if hit is not None and (hit[0] == ver or torch.cuda.is_current_stream_capturing()):
            return hit, True
        if hit is not None and hit[1].get("device") != ver[0]:
            hit = None                      # moved to another device: new buffers
        return hit, False

    def attn(self, key, p) -> dict:
        ver = (p["wq"].device,) + _ver(p["wq"], p["wk"], p["wv"], p["wo"], p["omega"], p["bq"], p["bk"], p["bv"])
        hit, fresh = self._lookup(("a", key), ver)
        if fresh:
            return hit[1]
        old = hit[1] if hit is not None else None
        with torch.no_grad():
            wqkv = torch.cat([p["wq"], p["wk"], p["wv"]], 0).to(F32).contiguous()
            wqkv_b, wqkv_t = ops.cast_weight(wqkv, out=(old["wqkv"], old["wqkv_t"]) if old else None)
            wo_b, wo_t = ops.cast_weight(p["wo"].detach().to(F32).contiguous(),
                                         out=(old["wo"], old["wo_t"]) if old else None)
            omega_b = ops.prep_omega(p["omega"].detach().to(F32).contiguous(), out=old["omega_b"] if old else None)
            bqkv = None
            if p["bq"] is not None:
                bqkv = torch.cat([p["bq"], p["bk"], p["bv"]], 0).to(F32).contiguous()
                if old is not None and old["bqkv"] is not None:
                    bqkv = old["bqkv"].copy_(bqkv)
        out = {"wqkv": wqkv_b, "wqkv_t": wqkv_t, "wo": wo_b, "wo_t": wo_t, "omega_b": omega_b, "bqkv": bqkv,
               "device": ver[0]}
        self.store[("a", key)] = (ver, out)
        return out

    def ffn(self, key, p) -> dict:
        ver = (p["w1"].device,) + _ver(p["w1"], p["w2"])
        hit, fresh = self._lookup(("f", key), ver)
        if fresh:
            return hit[1]
        old = hit[1] if hit is not None else None
        with torch.no_grad():
            w1_b, w1_t = ops.cast_weight(p["w1"].detach().to(F32).contiguous(),
                                         out=(old["w1"], old["w1_t"]) if old else None)
            w2_b, w2_t = ops.cast_weight(p["w2"].detach().to(F32).contiguous(),
                                         out=(old["w2"], old["w2_t"]) if old else None)
            # w2 with its hidden index in the order the one-launch forward reads it (no-tape path)
            w2_c = ops.chain_copy(w2_b)
            if old is not None:
                w2_c = old["w2_c"].copy_(w2_c)
        out = {"w1": w1_b, "w1_t": w1_t, "w2": w2_b, "w2_t": w2_t, "w2_c": w2_c, "device": ver[0]}
        self.store[("f", key)] = (ver, out)
        return out


def _attn_ver(p) -> tuple:
    return (p["wq"].device,) + _ver(p["wq"], p["wk"], p["wv"], p["wo"], p["omega"], p["bq"], p["bk"], p["bv"])


def _ffn_ver(p) -> tuple:
    return (p["w1"].device,) + _ver(p["w1"], p["w2"])


def _cast_table(cache: WeightCache, layers) -> Optional[Tuple[torch.Tensor, int, tuple]]:
    """Device table of scm_cast_item entries covering every weight matrix of the stack (q/k/v weights fill slices
    of the fused operand), or None when some weight is not a contiguous f32 tensor with an existing copy."""
    import struct
    rows = []
    ptrs = []
    for li, (ap, fp) in enumerate(layers):
        ha, hf = cache.store.get(("a", li)), cache.store.get(("f", li))
        if ha is None or hf is None or ap["bq"] is not None:
            return None
        wa, wf = ha[1], hf[1]
        inner, d = ap["wq"].shape
        srcs = [(ap["wq"], wa["wqkv"], 0, d, wa["wqkv_t"], 0, 3 * inner, 0),
                (ap["wk"], wa["wqkv"], inner * d, d, wa["wqkv_t"], inner, 3 * inner, 0),
                (ap["wv"], wa["wqkv"], 2 * inner * d, d, wa["wqkv_t"], 2 * inner, 3 * inner, 0),
                (ap["wo"], wa["wo"], 0, inner, wa["wo_t"], 0, d, 0),
                (fp["w1"], wf["w1"], 0, d, wf["w1_t"], 0, fp["w1"].shape[0], 0),
                (fp["w2"], wf["w2"], 0, fp["w2"].shape[1], wf["w2_t"], 0, d, 0),
                (fp["w2"], wf["w2_c"], 0, wf["w2_c"].shape[1], None, 0, 0, 1)]   # chain-order copy
        for src, dst, dst_off, ldd, dstt, dstt_off, ldt, chain in srcs:
            if src.dtype != F32 or not src.is_contiguous() or not dst.is_contiguous() or \
                    (dstt is not None and not dstt.is_contiguous()):
                return None
            r, c = src.shape
            rows.append(struct.pack("<QiiQqQqii", src.data_ptr(), r, c, dst.data_ptr() + 2 * dst_off, ldd,
                                    dstt.data_ptr() + 2 * dstt_off if dstt is not None else 0, ldt, chain, 0))
            if chain == 0:
                ptrs.append(src.data_ptr())
    if not rows:
        return None
    table = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).to(layers[0][0]["wq"].device)
    return table, len(rows), tuple(ptrs)


def refresh_weights(owner) -> None:
    """Bring the bf16 weight copies of a PerformerLM / Performer up to date now (in place).  Needed only by callers
    that replay a captured graph of the step: call it after every weight update, outside the graph.  Once the
    copies exist, ONE kernel launch recasts every weight matrix of the stack (a device table lists them)."""
    performer = getattr(owner, "performer", owner)
    cache = _cache_of(owner)
    layers = [(attn_params(f), ffn_params(g)) for f, g in performer.net.layers]
    tab = owner.__dict__.get("_scm_cast_table")
    if tab is not None and tab[2] != tuple(p.data_ptr() for ap, fp in layers
                                          for p in (ap["wq"], ap["wk"], ap["wv"], ap["wo"], fp["w1"], fp["w2"])):
        tab = None                               # parameters were re-allocated (moved, re-created): rebuild
    if tab is None:
        for li, (ap, fp) in enumerate(layers):   # creates or refreshes the copies layer by layer
            cache.attn(li, ap)
            cache.ffn(li, fp)
        owner.__dict__["_scm_cast_table"] = _cast_table(cache, layers)
        return
    stale = False
    for li, (ap, fp) in enumerate(layers):
        ha, hf = cache.store[("a", li)], cache.store[("f", li)]
        va, vf = _attn_ver(ap), _ffn_ver(fp)
        if ha[0] != va or hf[0] != vf:
            stale = True
            if ha[0][5] != va[5]:                # the projection matrix changed (redraw): its own small kernel
                ops.prep_omega(ap["omega"].detach().to(F32).contiguous(), out=ha[1]["omega_b"])
            cache.store[("a", li)] = (va, ha[1])
            cache.store[("f", li)] = (vf, hf[1])
    if stale:
        from ._lib import check, lib
        check(lib.scm_cast_weights_batch(tab[0].data_ptr(), tab[1], torch.cuda.current_stream().cuda_stream),
              "scm_cast_weights_batch")


def _cache_of(owner) -> WeightCache:
    c = owner.__dict__.get("_scm_wcache")
    if c is None:
        c = WeightCache()
        owner.__dict__["_scm_wcache"] = c
    return c


# --------------------------------------------------------------------------------------------
# branch forward / backward (shared by the fused stack and the stand-alone modules)
# --------------------------------------------------------------------------------------------
@dataclass
class AttnTape:
    x: Optional[torch.Tensor]      # f32 [T,D] input of the LayerNorm (None when ln=False)
    h: torch.Tensor                # bf16 [T,D] GEMM operand
    mean: Optional[torch.Tensor]
    rstd: Optional[torch.Tensor]
    qkv: torch.Tensor              # bf16 [B,N,3,H,64]
    o: torch.Tensor                # bf16 [B,N,H,64]
    oc: torch.Tensor               # bf16 [B,N,H,64] output minus the per-head mean value vector
    saved: Any
    shape: tuple


def _qkv_views(qkv5):
    return tuple(qkv5[:, :, j].permute(0, 2, 1, 3) for j in range(3))


def _out_product(a: torch.Tensor, w: torch.Tensor, bias, resid, next_ln: Optional[tuple]):
    """y = a @ w^T + bias + resid (f32 [T,D]); with next_ln = (gamma, beta) also the following sublayer's LayerNorm
    (h, mean, rstd) of y -- from the same launch where the sizes allow it.  Returns (y, that tuple or None)."""
    if next_ln is not None and ops.gemm_res_ln_supported(a, w) and (resid is None or resid.is_contiguous()):
        return ops.gemm_res_ln(a, w, bias, resid, next_ln[0], next_ln[1])
    y = torch.empty((a.shape[0], w.shape[0]), dtype=F32, device=a.device)
    ops.gemm_nt(ops.EPI_F32_BIAS_RES, a, w, out0=y, bias=bias, resid=resid)
    return y, (ops.ln_fwd(y, next_ln[0], next_ln[1]) if next_ln is not None else None)


def attn_fwd(x: torch.Tensor, p: dict, w: dict, *, ln: bool, resid: Optional[torch.Tensor], save: bool,
             pre_ln: Optional[tuple] = None, next_ln: Optional[tuple] = None, group_cells: Optional[int] = None):
    """x: f32 [B,N,D].  Returns (y f32 [B,N,D] = resid + to_out(attn(LN(x))) + bias, tape, ln_y).
    pre_ln: (h, mean, rstd) when the producer of x already applied this sublayer's LayerNorm.
    next_ln: (gamma, beta) of the sublayer that reads y: ln_y = its (h, mean, rstd), else None.
    group_cells: cells per key-stabiliser group (None: one stabiliser over the batch, performer_pytorch.py:90)."""
    b, n, d = x.shape
    t = b * n
    heads = p["heads"]
    inner = heads * ops.DIM_HEAD
    x2 = x.reshape(t, d)
    if pre_ln is not None:
        h, mean, rstd = pre_ln
    elif ln:
        h, mean, rstd = ops.ln_fwd(x2, p["g1"], p["b1"])
    else:
        h, mean, rstd = x2.to(BF16), None, None
    qkv = torch.empty((t, 3 * inner), dtype=BF16, device=x.device)
    if w["bqkv"] is not None:
        ops.gemm_nt(ops.EPI_BF16_BIAS, h, w["wqkv"], out0=qkv, bias=w["bqkv"])
    else:
        ops.gemm_nt(ops.EPI_BF16, h, w["wqkv"], out0=qkv)
    qkv5 = qkv.view(b, n, 3, heads, ops.DIM_HEAD)
    q, k, v = _qkv_views(qkv5)
    o = torch.empty((b, n, heads, ops.DIM_HEAD), dtype=BF16, device=x.device)
    oc = torch.empty((b, n, heads, ops.DIM_HEAD), dtype=BF16, device=x.device) if save else None
    saved = ops.favor_fwd(q, k, v, w["omega_b"], o.permute(0, 2, 1, 3), oc.permute(0, 2, 1, 3) if save else None,
                          group_cells)
    y, ln_y = _out_product(o.view(t, inner), w["wo"], p["bo"], resid.reshape(t, d) if resid is not None else None, next_ln)
    tape = AttnTape(x2 if ln else None, h, mean, rstd, qkv5, o, oc, saved, (b, n, d)) if save else None
    return y.view(b, n, d), tape, ln_y


def _dgrad_ln(a: torch.Tensor, w_t: torch.Tensor, tp, gamma, dres, dgamma, dbeta, dres_colsum, ln: bool, want_b: bool,
              defer: Optional[list]):
    """gradient with respect to a sublayer's input: dh = a @ w_t^T, then (ln) dx = dres + LN'(dh).  One launch where
    the sizes allow it (the f32 dh rows stay on the chip), else the product followed by the LayerNorm backward."""
    if not ln:
        dh = torch.empty((a.shape[0], w_t.shape[0]), dtype=F32, device=a.device)
        return ops.gemm_nt(ops.EPI_F32, a, w_t, out0=dh)
    if ops.gemm_ln_bwd_supported(a, w_t) and tp.x.is_contiguous() and (dres is None or dres.is_contiguous()):
        return ops.gemm_ln_bwd(a, w_t, tp.x, tp.mean, tp.rstd, gamma, dres, dgamma, dbeta, dres_colsum, False, want_b, defer)
    dh = torch.empty((a.shape[0], w_t.shape[0]), dtype=F32, device=a.device)
    ops.gemm_nt(ops.EPI_F32, a, w_t, out0=dh)
    return ops.ln_bwd(dh, tp.x, tp.mean, tp.rstd, gamma, dres, dgamma, dbeta, dres_colsum, False, want_b, defer)


def attn_bwd(dy: torch.Tensor, tp: AttnTape, p: dict, w: dict, need: Dict[str, bool], *, ln: bool,
             dres: Optional[torch.Tensor], need_dx: bool, dy_b: Optional[torch.Tensor] = None, want_b: bool = False,
             defer: Optional[list] = None):
    """dy: f32 [T,D] gradient of the branch output (dy_b: the same rounded to bf16, when the producer already
    wrote it; the matrix products read that copy).  Returns (dx f32 [T,D] or None, grads dict), dx being a
    (dx, dx_bf16) pair when want_b and ln.
    With ln: dx = dres + LN'(dh); grads['bo'] is taken from column sums of dy inside the LN kernel."""
    b, n, d = tp.shape
    t = b * n
    heads = p["heads"]
    inner = heads * ops.DIM_HEAD
    dev = dy.device
    g: Dict[str, torch.Tensor] = {}
    need_qkv_w = need["wq"] or need["wk"] or need["wv"] or need["bq"]
    # the two weight gradients of the sublayer (to_out; to_q/k/v) go out as ONE launch once both operand pairs exist
    wo_ops = (tp.o.view(t, inner), dy if dy_b is None else dy_b)
    pair = need["wo"] and need_qkv_w and ops.wgrad_multi_supported(*wo_ops, tp.h)
    if need["wo"]:
        g["wo"] = torch.empty((d, inner), dtype=F32, device=dev)
        if not pair:
            ops.wgrad(*wo_ops, g["wo"], None, True, False, defer)
    need_dh = need_dx or (ln and (need["g1"] or need["b1"]))
    if need_qkv_w or need_dh:
        do = torch.empty((b, n, heads, ops.DIM_HEAD), dtype=BF16, device=dev)
        ops.gemm_nt(ops.EPI_BF16, dy if dy_b is None else dy_b, w["wo_t"], out0=do.view(t, inner))
        dqkv = torch.empty((b, n, 3, heads, ops.DIM_HEAD), dtype=BF16, device=dev)
        q, k, v = _qkv_views(tp.qkv)
        dq, dk, dv = _qkv_views(dqkv)
        ops.favor_bwd(q, k, v, tp.oc.permute(0, 2, 1, 3), do.permute(0, 2, 1, 3), w["omega_b"], tp.saved, dq, dk, dv)
        dqkv2 = dqkv.view(t, 3 * inner)
        if need_qkv_w:
            gw = torch.empty((3 * inner, d), dtype=F32, device=dev)
            gb = torch.empty(3 * inner, dtype=F32, device=dev) if w["bqkv"] is not None else None
            if pair:
                ops.wgrad_multi([(*wo_ops, g["wo"], None, True), (dqkv2, tp.h, gw, gb, False)], False, defer)
            else:
                ops.wgrad(dqkv2, tp.h, gw, gb, False, False, defer)
            g["wq"], g["wk"], g["wv"] = gw[:inner], gw[inner:2 * inner], gw[2 * inner:]
            if gb is not None:
                g["bq"], g["bk"], g["bv"] = gb[:inner], gb[inner:2 * inner], gb[2 * inner:]
    dx = None
    if need_dh:
        if ln:
            g["g1"] = torch.empty(d, dtype=F32, device=dev)
            g["b1"] = torch.empty(d, dtype=F32, device=dev)
            g["bo"] = torch.empty(d, dtype=F32, device=dev) if dres is not None else None
        dx = _dgrad_ln(dqkv2, w["wqkv_t"], tp, p["g1"], dres, g.get("g1"), g.get("b1"), g.get("bo"), ln, want_b, defer)
    if need["bo"] and g.get("bo") is None:
        g["bo"] = dy.sum(0)
    return dx, g


@dataclass
class FfnTape:
    x: Optional[torch.Tensor]
    h: torch.Tensor
    mean: Optional[torch.Tensor]
    rstd: Optional[torch.Tensor]
    pre: torch.Tensor
    act: torch.Tensor
    shape: tuple


def ffn_fwd(x: torch.Tensor, p: dict, w: dict, *, ln: bool, resid: Optional[torch.Tensor], save: bool,
            pre_ln: Optional[tuple] = None, next_ln: Optional[tuple] = None):
    """y = resid + w2(gelu(w1(LN(x)) + c1)) + c2; pre_ln, next_ln and the third result as in attn_fwd"""
    b, n, d = x.shape
    t = b * n
    f = p["w1"].shape[0]
    x2 = x.reshape(t, d)
    if pre_ln is not None:
        h, mean, rstd = pre_ln
    elif ln:
        h, mean, rstd = ops.ln_fwd(x2, p["g2"], p["b2"])
    else:
        h, mean, rstd = x2.to(BF16), None, None
    pre = ops.empty_rows(t, f, BF16, x.device)
    act = ops.empty_rows(t, f, BF16, x.device)
    ops.gemm_nt(ops.EPI_GELU_DUAL, h, w["w1"], out0=pre, out1=act, bias=p["c1"])
    y, ln_y = _out_product(act, w["w2"], p["c2"], resid.reshape(t, d) if resid is not None else None, next_ln)
    tape = FfnTape(x2 if ln else None, h, mean, rstd, pre, act, (b, n, d)) if save else None
    return y.view(b, n, d), tape, ln_y


def ffn_bwd(dy: torch.Tensor, tp: FfnTape, p: dict, w: dict, need: Dict[str, bool], *, ln: bool,
            dres: Optional[torch.Tensor], need_dx: bool, dy_b: Optional[torch.Tensor] = None, want_b: bool = False,
            defer: Optional[list] = None):
    """as attn_bwd, for the FeedForward branch"""
    b, n, d = tp.shape
    t = b * n
    f = p["w1"].shape[0]
    dev = dy.device
    g: Dict[str, torch.Tensor] = {}
    w2_ops = (tp.act, dy if dy_b is None else dy_b)
    pair = need["w2"] and (need["w1"] or need["c1"]) and ops.wgrad_multi_supported(*w2_ops, tp.h)
    if need["w2"]:
        g["w2"] = torch.empty((d, f), dtype=F32, device=dev)
        if not pair:
            ops.wgrad(*w2_ops, g["w2"], None, True, False, defer)
    need_dh = need_dx or (ln and (need["g2"] or need["b2"]))
    if need["w1"] or need["c1"] or need_dh:
        dpre = ops.empty_rows(t, f, BF16, dev)
        ops.gemm_nt(ops.EPI_DGELU, dy if dy_b is None else dy_b, w["w2_t"], out0=dpre, pre=tp.pre)
        if need["w1"] or need["c1"]:
            g["w1"] = torch.empty((f, d), dtype=F32, device=dev)
            g["c1"] = torch.empty(f, dtype=F32, device=dev)
            if pair:    # w2's and w1's weight gradients in one launch
                ops.wgrad_multi([(*w2_ops, g["w2"], None, True), (dpre, tp.h, g["w1"], g["c1"], False)], False, defer)
            else:
                ops.wgrad(dpre, tp.h, g["w1"], g["c1"], False, False, defer)
    dx = None
    if need_dh:
        if ln:
            g["g2"] = torch.empty(d, dtype=F32, device=dev)
            g["b2"] = torch.empty(d, dtype=F32, device=dev)
            g["c2"] = torch.empty(d, dtype=F32, device=dev) if dres is not None else None
        dx = _dgrad_ln(dpre, w["w1_t"], tp, p["g2"], dres, g.get("g2"), g.get("b2"), g.get("c2"), ln, want_b, defer)
    if need["c2"] and g.get("c2") is None:
        g["c2"] = dy.sum(0)
    return dx, g


# --------------------------------------------------------------------------------------------
# the fused stack
# --------------------------------------------------------------------------------------------
@dataclass
class StackSpec:
    layers: List[Tuple[dict, dict]]          # (attn_params, ffn_params) per layer
    cache: WeightCache
    names: List[Tuple[int, str]]             # flat differentiable-parameter order: (layer or -1/-2, key)
    token_emb: Optional[torch.Tensor] = None
    pos_table: Optional[torch.Tensor] = None
    norm_w: Optional[torch.Tensor] = None
    norm_b: Optional[torch.Tensor] = None
    head_w: Optional[torch.Tensor] = None    # [C, D]
    head_b: Optional[torch.Tensor] = None
    head_relu: bool = False
    head_out_shape: Optional[str] = None     # 'linear' | 'conv' | None (encodings only)
    token_err: Optional[torch.Tensor] = None  # device int32 [1]: set by the embedding kernel on a bad token id
    save: bool = True
    group_cells: Optional[int] = None        # cells per key-stabiliser group (None: the whole batch)


def _flat_params(spec: StackSpec) -> List[torch.Tensor]:
    out = []
    for li, key in spec.names:
        if li == -1:
            out.append(spec.token_emb)
        elif li == -2:
            out.append({"norm_w": spec.norm_w, "norm_b": spec.norm_b, "head_w": spec.head_w,
                        "head_b": spec.head_b}[key])
        else:
            a, f = spec.layers[li]
            out.append(a[key] if key in a else f[key])
    return out


# --------------------------------------------------------------------------------------------
# gradient sink: parameter gradients written straight into a caller's accumulators
# --------------------------------------------------------------------------------------------
_TLS = threading.local()     # .sink: id(parameter) -> f32 accumulator of the parameter's shape, while a grad_sink is open


class grad_sink:
    """Context manager for a training driver that accumulates gradients over micro-steps.  A forward pass of the
    fused stack run inside it remembers the table (on its autograd context), and its backward pass -- whichever
    thread autograd runs it on, inside the context or after it -- ADDS every parameter gradient whose final
    reduction it can redirect into accs[i] (the last launch of the backward does the add) and hands None back to
    autograd for that parameter, instead of materialising the gradient for the caller to add.  Gradients it cannot
    redirect are returned as usual.  The table is thread-local where it is looked up (the forward, in the driver's
    own thread): drivers on different threads, or nested ones, do not see each other's accumulators."""

    def __init__(self, params, accs):
        self.table = {id(p): a for p, a in zip(params, accs)}
        self.before = None

    def __enter__(self):
        self.before = getattr(_TLS, "sink", None)
        _TLS.sink = self.table
        return self

    def __exit__(self, *exc):
        _TLS.sink = self.before
        return False


def _redirect_jobs(jobs: list, grads: Dict[Tuple[int, str], torch.Tensor], byname: Dict[Tuple[int, str], Any],
                   sink: Dict[int, torch.Tensor]) -> set:
    """Point the deferred reductions at the sink's accumulators where that is exact: every output of a job must be
    covered by gradient tensors whose accumulators sit at the same relative offsets (the fused q/k/v weight gradient
    is one [3*inner, D] output: its three accumulators must be adjacent).  Returns the names now accumulated in place."""
    sunk: set = set()
    if not sink:
        return sunk
    spans = []                                       # (gradient pointer, bytes, name, accumulator)
    for nm, gt in grads.items():
        acc = sink.get(id(byname.get(nm)))
        spans.append((gt.data_ptr(), gt.numel() * 4, nm, acc if (acc is not None and gt.is_contiguous() and
                                                                 acc.is_contiguous() and acc.dtype == F32 and
                                                                 acc.numel() == gt.numel()) else None))
    for j, _ws in jobs:
        if j.kind == 0:
            outs = [(0, j.P * j.Q * 4)] + ([(1, j.P * 4)] if j.out[1] else [])
        else:
            outs = [(k, (j.hi[k] - j.lo[k]) * 4) for k in range(4) if j.out[k]]
        plan, names, ok = [], [], True
        for k, nbytes in outs:
            o = j.out[k]
            members = [sp for sp in spans if o <= sp[0] < o + nbytes]
            base = next((sp for sp in members if sp[0] == o), None)
            if base is None or base[3] is None or sum(sp[1] for sp in members) != nbytes:
                ok = False
                break
            for sp in members:
                if sp[3] is None or sp[3].data_ptr() - base[3].data_ptr() != sp[0] - o:
                    ok = False
                    break
            if not ok:
                break
            plan.append((k, base[3].data_ptr()))
            names += [sp[2] for sp in members]
        if ok and plan:
            # accumulate applies to the whole job: every output it writes must have been redirected
            assert len(plan) == sum(1 for o in j.out if o), "a reduction job with an output outside the sink"
            for k, ptr in plan:
                j.out[k] = ptr
            j.accumulate = 1
            sunk.update(names)
    return sunk


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: StackSpec, inp: torch.Tensor, *params):
        if not inp.is_cuda:
            raise RuntimeError("scmomer_amd runs on the MI355X kernels only: move the model and inputs to the GPU")
        tapes = []
        if spec.token_emb is not None:
            tokens = inp
            x = ops.embed_fwd(tokens.contiguous(), spec.token_emb.detach(), spec.pos_table, spec.token_err)
        else:
            tokens = None
            x = inp.to(F32).contiguous()
        pre_ln = None
        for li, (ap, fp) in enumerate(spec.layers):
            aw = spec.cache.attn(li, ap)
            fw = spec.cache.ffn(li, fp)
            b_, n_, d_ = x.shape
            nxt = spec.layers[li + 1][0] if li + 1 < len(spec.layers) else None
            chain = not spec.save and fp["c1"] is not None and ops.ffn_chain_supported(b_ * n_, d_, fp["w1"].shape[0])
            # the output product of a sublayer also applies the LayerNorm of the sublayer that follows it
            x1, at, ln2 = attn_fwd(x, ap, aw, ln=True, resid=x, save=spec.save, pre_ln=pre_ln,
                                   next_ln=None if chain else (fp["g2"], fp["b2"]), group_cells=spec.group_cells)
            pre_ln = None
            if chain:
                # no tape: the whole FeedForward sublayer in one launch, which also applies the next layer's LayerNorm
                y, _, pre_ln = ops.ffn_chain_fwd(x1.reshape(b_ * n_, d_), fp["g2"], fp["b2"], fw["w1"], fp["c1"], fw["w2_c"],
                                                 fp["c2"], save=False,
                                                 next_ln=(nxt["g1"], nxt["b1"]) if nxt is not None else None)
                x2, ft = y.view(b_, n_, d_), None
            else:
                x2, ft, pre_ln = ffn_fwd(x1, fp, fw, ln=True, resid=x1, save=spec.save, pre_ln=ln2,
                                         next_ln=(nxt["g1"], nxt["b1"]) if nxt is not None else None)
            tapes.append((at, ft))
            x = x2
        ctx.spec, ctx.tapes, ctx.tokens = spec, tapes, tokens
        ctx.sink = getattr(_TLS, "sink", None)               # the driver's accumulators, if a grad_sink is open
        if spec.norm_w is None:
            ctx.head = None
            return x
        want_out = spec.head_out_shape is not None
        enc, out, mean, rstd = ops.head_fwd(x, spec.norm_w, spec.norm_b, spec.head_w if want_out else None,
                                            spec.head_b if want_out else None, spec.head_relu, not want_out,
                                            want_out)
        ctx.head = (x, mean, rstd, out) if spec.save else None
        if not want_out:
            return enc
        if spec.head_out_shape == "conv":
            return out.view(out.shape[0], -1)
        return out

    @staticmethod
    def backward(ctx, dout):
        spec: StackSpec = ctx.spec
        if not spec.save:
            raise RuntimeError("backward through a stack that was run without saving activations")
        needs = ctx.needs_input_grad
        need_p = {nm: needs[2 + i] for i, nm in enumerate(spec.names)}
        grads: Dict[Tuple[int, str], torch.Tensor] = {}
        nl = len(spec.layers)
        # lowest layer whose parameters (or whose input) still need a gradient
        need_inp = needs[1] or need_p.get((-1, "token_emb"), False)
        first = nl
        for li in range(nl):
            if any(need_p.get((li, k), False) for k in ATTN_KEYS + FFN_KEYS):
                first = li
                break
        if need_inp:
            first = 0
        dout = dout.contiguous()
        # The small reductions that end the gradient kernels (slab sums of the weight gradients, column sums of the
        # LayerNorm / head / embedding gradients: about forty per backward) are collected and run as ONE launch.
        jobs: list = []
        if ctx.head is not None:
            x, mean, rstd, out = ctx.head
            t, d = x.numel() // x.shape[-1], x.shape[-1]
            want_out = spec.head_out_shape is not None
            dev = x.device
            gg = torch.empty(d, dtype=F32, device=dev)
            gb = torch.empty(d, dtype=F32, device=dev)
            gw = torch.empty_like(spec.head_w) if want_out else None
            gbias = torch.empty_like(spec.head_b) if want_out else None
            dx = ops.head_bwd(dout.to(F32).reshape(t, -1) if want_out else None,
                              None if want_out else dout.to(F32).reshape(t, d),
                              out, x, mean, rstd, spec.norm_w, spec.norm_b, spec.head_w if want_out else None,
                              spec.head_relu, gg, gb, gw.view(-1, d) if want_out else None,
                              gbias.view(-1) if want_out else None, False, True, jobs)
            dx, dxb = dx[0].reshape(t, d), dx[1].reshape(t, d)
            grads[(-2, "norm_w")], grads[(-2, "norm_b")] = gg, gb
            if want_out:
                grads[(-2, "head_w")], grads[(-2, "head_b")] = gw, gbias
        else:
            dx = dout.to(F32).reshape(-1, dout.shape[-1])
            dxb = None
        for li in range(nl - 1, first - 1, -1):
            ap, fp = spec.layers[li]
            aw, fw = spec.cache.attn(li, ap), spec.cache.ffn(li, fp)
            at, ft = ctx.tapes[li]
            lowest = li == first and not need_inp
            nf = {k: need_p.get((li, k), False) for k in FFN_KEYS}
            na = {k: need_p.get((li, k), False) for k in ATTN_KEYS}
            attn_needs_any = any(na.values())
            # the LayerNorm-backward kernels also emit their result rounded to bf16: the operand of the next
            # data- and weight-gradient products
            dx1, gf = ffn_bwd(dx, ft, fp, fw, nf, ln=True, dres=dx, need_dx=(not lowest) or attn_needs_any,
                              dy_b=dxb, want_b=True, defer=jobs)
            for k, v in gf.items():
                if v is not None:
                    grads[(li, k)] = v
            if dx1 is None:
                break
            dx1, dx1b = dx1
            dx0, ga = attn_bwd(dx1, at, ap, aw, na, ln=True, dres=dx1, need_dx=not lowest, dy_b=dx1b, want_b=True,
                               defer=jobs)
            for k, v in ga.items():
                if v is not None:
                    grads[(li, k)] = v
            dx, dxb = dx0 if dx0 is not None else (None, None)
        d_inp = None
        if first == 0 and need_inp and dx is not None:
            if ctx.tokens is not None:
                if need_p.get((-1, "token_emb"), False):
                    ge = torch.empty_like(spec.token_emb)
                    ops.embed_bwd(ctx.tokens.contiguous(), dx, spec.token_emb.shape[0], ge, False, jobs)
                    grads[(-1, "token_emb")] = ge
            else:
                d_inp = dx.view(ctx.tapes[0][0].shape) if nl else dx
        sunk = _redirect_jobs(jobs, grads, dict(zip(spec.names, _flat_params(spec))), ctx.sink) if ctx.sink else set()
        ops.reduce_batch(jobs)
        out_grads = []
        for i, nm in enumerate(spec.names):
            gval = grads.get(nm) if (needs[2 + i] and nm not in sunk) else None
            out_grads.append(gval)
        ctx.tapes = None
        return (None, d_inp, *out_grads)


def _build_spec(owner, layer_list, *, save: bool) -> StackSpec:
    layers = [(attn_params(f), ffn_params(g)) for f, g in layer_list]
    names: List[Tuple[int, str]] = []
    for li, (a, f) in enumerate(layers):
        for k in ATTN_KEYS:
            if a[k] is not None:
                names.append((li, k))
        for k in FFN_KEYS:
            names.append((li, k))
    performer = getattr(owner, "performer", owner)
    return StackSpec(layers=layers, cache=_cache_of(owner), names=names, save=save,
                     group_cells=getattr(performer, "stabiliser_group_cells", None))


def _head_kind(to_out, dim: int):
    """'linear' for nn.Linear(dim, C<=8); 'conv' for the rna_encoder.py Identity head (1 x dim conv + ReLU)."""
    if isinstance(to_out, nn.Linear) and to_out.in_features == dim and to_out.out_features <= 8 \
            and to_out.bias is not None:
        return "linear"
    conv = getattr(to_out, "conv1", None)
    if isinstance(conv, nn.Conv2d) and tuple(conv.weight.shape) == (1, 1, 1, dim) and conv.bias is not None \
            and isinstance(getattr(to_out, "act", None), nn.ReLU):
        return "conv"
    return None


def token_error_flag(model, device) -> torch.Tensor:
    """The model's bad-token flag on `device` (created on first use, one per device): every forward ORs into it
    without synchronising; PerformerLM.raise_on_bad_token reads and clears it."""
    flags = model.__dict__.setdefault("_scm_token_err", {})
    key = (device.type, device.index)
    if key not in flags:
        flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return flags[key]


def run_model(model, tokens: torch.Tensor, return_encodings: bool):
    """PerformerLM.forward on the fused stack."""
    if tokens.dtype != torch.int64:
        tokens = tokens.long()
    dim = model.token_emb.weight.shape[1]
    grad_on = torch.is_grad_enabled()
    spec = _build_spec(model, model.performer.net.layers, save=grad_on)
    spec.token_emb = model.token_emb.weight
    spec.token_err = token_error_flag(model, tokens.device)
    spec.names.insert(0, (-1, "token_emb"))
    pos = getattr(model.pos_emb, "emb", None)
    spec.pos_table = pos.weight.detach() if pos is not None else None
    spec.norm_w, spec.norm_b = model.norm.weight, model.norm.bias
    spec.names += [(-2, "norm_w"), (-2, "norm_b")]
    kind = None if (return_encodings or model.to_out is None) else _head_kind(model.to_out, dim)
    post = None
    if kind == "linear":
        spec.head_w, spec.head_b, spec.head_out_shape = model.to_out.weight, model.to_out.bias, "linear"
        spec.names += [(-2, "head_w"), (-2, "head_b")]
    elif kind == "conv":
        spec.head_w, spec.head_b = model.to_out.conv1.weight, model.to_out.conv1.bias
        spec.head_relu, spec.head_out_shape = True, "conv"
        spec.names += [(-2, "head_w"), (-2, "head_b")]
    elif not return_encodings:
        post = model.to_out  # unknown head module: encodings from the kernels, head in torch
    params = _flat_params(spec)
    spec.save = grad_on and (any(p.requires_grad for p in params))
    out = _StackFn.apply(spec, tokens, *params)
    if post is not None:
        return post(out)
    if model.to_out is None and not return_encodings:
        return out @ model.token_emb.weight.t()
    return out


def run_layers(performer, x: torch.Tensor):
    """Performer.forward / SequentialSequence.forward on an already-embedded f32 [B,N,D] input."""
    grad_on = torch.is_grad_enabled()
    spec = _build_spec(performer, performer.net.layers, save=grad_on)
    params = _flat_params(spec)
    spec.save = grad_on and (x.requires_grad or any(p.requires_grad for p in params))
    return _StackFn.apply(spec, x, *params)
